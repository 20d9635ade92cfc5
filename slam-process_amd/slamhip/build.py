"""Builds libslamhip.so (hipcc, gfx950) from slam-process_amd/csrc/*.hip into slam-process_amd/slamhip/.

Every compile also asks the compiler for its per-kernel resource report (registers, LDS, scratch, waves per SIMD);
the default build writes it next to the library (kernel_resources.txt) and into profiles/, so the occupancy each
kernel is designed for can be checked without a GPU (tests/test_abi_and_host.py::test_kernel_resources).
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.environ.get("SLAMHIP_OUT") or os.path.join(HERE, "libslamhip.so")     # tuning builds: another output name ...
OBJ = os.environ.get("SLAMHIP_OBJ") or os.path.join(CSRC, "_obj")              # ... and object directory
# -ffp-contract=off: the f64 mean / blur / scale steps must round like NumPy, one operation at a time
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage"]
REPORT = os.path.join(os.path.dirname(OUT), "kernel_resources.txt")
REPORT_COPY = os.path.join(ROOT, "profiles", "r3_kernel_resources.txt")
_FIELDS = (("TotalSGPRs", "sgpr"), ("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("ScratchSize [bytes/lane]", "scratch"),
           ("Occupancy [waves/SIMD]", "waves"), ("SGPRs Spill", "sgpr_spill"), ("VGPRs Spill", "vgpr_spill"),
           ("LDS Size [bytes/block]", "lds"))


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build() -> bool:
    if not os.path.exists(OUT) or not os.path.exists(REPORT):
        return True
    t = min(os.path.getmtime(OUT), os.path.getmtime(REPORT))
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps.append(os.path.join(ROOT, "include", "slamhip.h"))
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def parse_resource_remarks(text: str) -> List[Dict[str, object]]:
    """The compiler's kernel-resource-usage remarks -> one dict per kernel (mangled name and the numbers)."""
    out: List[Dict[str, object]] = []
    cur = None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            out.append(cur)
            continue
        if cur is None:
            continue
        for label, key in _FIELDS:
            m = re.search(r"remark:\s+" + re.escape(label) + r": (\d+)", line)
            if m:
                cur[key] = int(m.group(1))
    return [k for k in out if "vgpr" in k]


def _demangle(names: List[str]) -> List[str]:
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
        except OSError:
            continue
        if r.returncode == 0:
            return r.stdout.splitlines()
    return names


def write_report(per_file: Dict[str, List[Dict[str, object]]], path: str) -> None:
    rows = []
    for src in sorted(per_file):
        kernels = per_file[src]
        plain = _demangle([str(k["name"]) for k in kernels])
        for k, p in zip(kernels, plain):
            if "rocprim::" in p:          # kernels of the library sort (k_sort.hip): not ours to report
                continue
            p = re.sub(r"\(anonymous namespace\)::", "", p)
            p = re.sub(r"^void ", "", p)
            p = p.split("(")[0] if "<" not in p else p[:p.index(">(") + 1] if ">(" in p else p
            rows.append((os.path.basename(src), p, k))
    with open(path, "w") as fh:
        fh.write("# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, written by slamhip/build.py on every build\n")
        fh.write("# file\tkernel\tvgpr\tagpr\tsgpr\tlds_bytes\tscratch_bytes_per_lane\tvgpr_spill\tsgpr_spill\twaves_per_simd\n")
        for f, p, k in rows:
            fh.write("\t".join([f, p] + [str(k.get(x, 0)) for x in
                                         ("vgpr", "agpr", "sgpr", "lds", "scratch", "vgpr_spill", "sgpr_spill", "waves")]) + "\n")


def read_report(path: str = REPORT) -> List[Dict[str, object]]:
    """The report as a list of dicts: file, kernel, vgpr, agpr, sgpr, lds, scratch, vgpr_spill, sgpr_spill, waves."""
    keys = ("file", "kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "vgpr_spill", "sgpr_spill", "waves")
    out = []
    with open(path) as fh:
        for line in fh:
            if line.startswith("#") or not line.strip():
                continue
            parts = line.rstrip("\n").split("\t")
            out.append({k: (v if i < 2 else int(v)) for i, (k, v) in enumerate(zip(keys, parts))})
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    remarks: Dict[str, List[Dict[str, object]]] = {}

    def one(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc, *FLAGS, *os.environ.get("SLAM_EXTRA_FLAGS", "").split(), "-c", src, "-o", obj]     # tuning builds: -D... overrides
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        remarks[src] = parse_resource_remarks(r.stderr)
        if verbose:
            rest = [ln for ln in r.stderr.splitlines()
                    if "kernel-resource-usage" not in ln and not re.match(r"^\s*(\d+ \|.*|\|\s*\^.*)?$", ln)]
            if any("warning:" in ln or "error:" in ln for ln in rest):      # (include chains of the remarks alone: nothing to say)
                sys.stderr.write("\n".join(rest) + "\n")
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(one, sources()))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    write_report(remarks, REPORT)
    if not os.environ.get("SLAMHIP_OUT") and not os.environ.get("SLAM_EXTRA_FLAGS") and os.path.isdir(os.path.dirname(REPORT_COPY)):
        write_report(remarks, REPORT_COPY)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
