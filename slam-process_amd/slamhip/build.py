"""Builds libslamhip.so (hipcc, gfx950) from slam-process_amd/csrc/*.hip into slam-process_amd/slamhip/."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
OUT = os.environ.get("SLAMHIP_OUT") or os.path.join(HERE, "libslamhip.so")     # tuning builds: another output name ...
OBJ = os.environ.get("SLAMHIP_OBJ") or os.path.join(CSRC, "_obj")              # ... and object directory
# -ffp-contract=off: the f64 mean / blur / scale steps must round like NumPy, one operation at a time
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "slamhip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)

    def one(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc, *FLAGS, *os.environ.get("SLAM_EXTRA_FLAGS", "").split(), "-c", src, "-o", obj]     # tuning builds: -D... overrides
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(one, sources()))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
