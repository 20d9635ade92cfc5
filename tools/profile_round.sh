#!/bin/bash
# Kernel trace + HBM traffic passes of the default bench shape; results under gpurun_out/prof (copy what is to be
# kept into profiles/).  Run from the repo root on the GPU box.
set -e
root=$(pwd)
mkdir -p "$root/gpurun_out/prof"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 "$root/bench.py" --cpu-frames 0 --steps 30 --warmup 2 --no-other-configs --no-pcie-fed > "$root/gpurun_out/prof/kt_bench.json" 2> "$root/gpurun_out/prof/kt.log"
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) "$root/gpurun_out/prof/kernel_stats.csv"
cd "$root"
tools/pmc_pass.sh fetch "FETCH_SIZE"
tools/pmc_pass.sh write "WRITE_SIZE"
python3 tools/pmc_traffic.py gpurun_out/pmc/fetch_counter_collection.csv gpurun_out/pmc/write_counter_collection.csv 800 256 4096 gpurun_out/prof/pmc_traffic.json
