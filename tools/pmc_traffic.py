"""HBM traffic per kernel from two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE) over
`bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-kernel-timing --frames-per-step 800 --distinct-frames 800`.

usage: pmc_traffic.py <fetch_csv> <write_csv> <frames_per_launch> <beams> <samples> <out.json> [table up-scaling factor]

Counters are KB per dispatch; the median over a kernel's dispatches is used.  On gfx950 FETCH_SIZE counts half
of the bytes of wide coalesced reads (a 4 GiB read-only stream of 16-byte loads reports 2,097,162 KB; the
calibration run is profiles/r1_pmc/probe_fetch_size_counter_collection.csv), so fetch bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE * 1024 is exact.
"""
import csv
import json
import re
import statistics
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            k = re.sub(r"^void |\(.*$", "", k)
            suffix = ""
            m = re.match(r"k_rows_finish<.*?, (\d+), \d+>", k)       # lean / mid / full build of the correction kernel
            if m:
                suffix = {"0": "", "1": "_mid", "2": "_rest"}[m.group(1)]
            k = re.sub(r"<.*$", "", k).split("::")[-1] + suffix
            if k.startswith("k_"):
                acc[k[2:]].append(float(r["Counter_Value"]))
    return {k: statistics.median(v) for k, v in acc.items()}


def main():
    fetch_csv, write_csv, fpl, beams, samples, out = sys.argv[1:7]
    fpl = int(fpl)
    upscale = int(sys.argv[7]) if len(sys.argv) > 7 else 1
    fe, wr = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        f, w = fe.get(k, 0.0), wr.get(k, 0.0)
        kernels[k] = dict(fetch_bytes_per_frame=int(2 * f * 1024 / fpl), write_bytes_per_frame=int(w * 1024 / fpl),
                          raw_FETCH_SIZE_KB_per_launch=f, raw_WRITE_SIZE_KB_per_launch=w)
    not_pipeline = ("synth", "copy16", "read16", "read_tiles", "mix16")        # the generator and the bandwidth probes
    doc = dict(method=__doc__.strip(), frames_per_launch=fpl, shape=[int(beams), int(samples)], upscale=upscale, kernels=kernels,
               total_bytes_per_frame=sum(v["fetch_bytes_per_frame"] + v["write_bytes_per_frame"] for v in kernels.values()),
               pipeline_bytes_per_frame=sum(v["fetch_bytes_per_frame"] + v["write_bytes_per_frame"]
                                            for k, v in kernels.items() if k not in not_pipeline))
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: (v["fetch_bytes_per_frame"], v["write_bytes_per_frame"]) for k, v in kernels.items()}))
    print("bytes/frame: all kernels", doc["total_bytes_per_frame"], "pipeline (without the generator)", doc["pipeline_bytes_per_frame"])


if __name__ == "__main__":
    main()
