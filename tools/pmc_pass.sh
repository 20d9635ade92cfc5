#!/bin/bash
# One rocprofv3 counter pass over a short bench run.  usage: tools/pmc_pass.sh <name> "<COUNTER ...>" [bench args]
# Writes gpurun_out/pmc/<name>_counter_collection.csv (each pass needs a run of its own).
set -e
name=$1; shift
ctrs=$1; shift
root=$(pwd)
mkdir -p "$root/gpurun_out/pmc"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$name
timeout -s KILL 240 rocprofv3 --pmc $ctrs -d /tmp/pmc_$name -o $name --output-format csv -- python3 "$root/bench.py" --steps 2 --warmup 1 --cpu-frames 0 --no-kernel-timing --frames-per-step 800 --distinct-frames 800 --no-other-configs --no-pcie-fed "$@" > "$root/gpurun_out/pmc/$name.log" 2>&1
f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
cp "$f" "$root/gpurun_out/pmc/${name}_counter_collection.csv"
echo "pass $name ok"
