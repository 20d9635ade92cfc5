#!/bin/bash
# Every profile of the bench, run from the repo root on the GPU box; results under gpurun_out/prof2 (copy what is kept
# into profiles/).  One rocprofv3 run per pass, each under its own time limit; any failure ends the script.
#   1. kernel trace + stats of the default bench command (the line it printed is kept beside the trace)
#   2. FETCH_SIZE and WRITE_SIZE passes of the headline shape -> per-kernel HBM traffic
#   3. one pass of eight SQ counters for each of the three shapes
#   4. kernel trace + stats, FETCH_SIZE and WRITE_SIZE passes of the two other shapes
# usage: tools/profile_all.sh [regex]      (only the passes whose name matches, e.g. '^(fetch|write)_')
# BENCH_EXTRA="--fused 1" TAG=_fused tools/profile_all.sh ...   the same passes with extra bench arguments, kept apart
set -e
ONLY=$1
root=$(pwd)
out="$root/gpurun_out/prof2$TAG"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp

pass() {   # pass <name> <rocprofv3 options...> -- <bench args...>
    name=$1; shift
    if [ -n "$ONLY" ] && [[ ! $name =~ $ONLY ]]; then return 0; fi
    opts=()
    while [ "$1" != "--" ]; do opts+=("$1"); shift; done
    shift
    rm -rf /tmp/p2_$name
    timeout -s KILL 280 rocprofv3 "${opts[@]}" -d /tmp/p2_$name -o $name --output-format csv -- python3 "$root/bench.py" "$@" $BENCH_EXTRA \
        > "$out/$name.json" 2> "$out/$name.log"
    for f in $(find /tmp/p2_$name -name "*.csv"); do cp "$f" "$out/${name}_$(basename $f | sed "s/^${name}_//")"; done
    echo "pass $name ok"
}

SHORT="--steps 2 --warmup 1 --cpu-frames 0 --no-kernel-timing --no-other-configs --no-pcie-fed"
pass default --kernel-trace --stats -- --no-other-configs --no-pcie-fed
pass fetch --pmc FETCH_SIZE -- $SHORT --frames-per-step 800 --distinct-frames 800
pass write --pmc WRITE_SIZE -- $SHORT --frames-per-step 800 --distinct-frames 800
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"
pass sq_256x4096 --pmc $SQ -- $SHORT --frames-per-step 800 --distinct-frames 800
pass sq_256x16384 --pmc $SQ -- $SHORT --beams 256 --samples 16384 --frames-per-step 200 --distinct-frames 200
pass sq_1024x8192_up4 --pmc $SQ -- $SHORT --beams 1024 --samples 8192 --frames-per-step 100 --distinct-frames 100 --upscale 4
pass kt_256x16384 --kernel-trace --stats -- --cpu-frames 0 --no-other-configs --no-pcie-fed --steps 5 --warmup 1 --beams 256 --samples 16384 --frames-per-step 200 --distinct-frames 200
pass kt_1024x8192_up4 --kernel-trace --stats -- --cpu-frames 0 --no-other-configs --no-pcie-fed --steps 5 --warmup 1 --beams 1024 --samples 8192 --frames-per-step 100 --distinct-frames 100 --upscale 4
LONG="--beams 256 --samples 16384 --frames-per-step 200 --distinct-frames 200"
WIDE="--beams 1024 --samples 8192 --frames-per-step 100 --distinct-frames 100 --upscale 4"
pass fetch_256x16384 --pmc FETCH_SIZE -- $SHORT $LONG
pass write_256x16384 --pmc WRITE_SIZE -- $SHORT $LONG
pass fetch_1024x8192_up4 --pmc FETCH_SIZE -- $SHORT $WIDE
pass write_1024x8192_up4 --pmc WRITE_SIZE -- $SHORT $WIDE
PLAIN="--beams 1024 --samples 8192 --frames-per-step 100 --distinct-frames 100"
pass fetch_1024x8192 --pmc FETCH_SIZE -- $SHORT $PLAIN
pass write_1024x8192 --pmc WRITE_SIZE -- $SHORT $PLAIN
cd "$root"
[ -f "$out/fetch_counter_collection.csv" ] && [ -f "$out/write_counter_collection.csv" ] && \
    python3 tools/pmc_traffic.py "$out/fetch_counter_collection.csv" "$out/write_counter_collection.csv" 800 256 4096 "$out/pmc_traffic_256x4096.json"
[ -f "$out/fetch_256x16384_counter_collection.csv" ] && [ -f "$out/write_256x16384_counter_collection.csv" ] && \
    python3 tools/pmc_traffic.py "$out/fetch_256x16384_counter_collection.csv" "$out/write_256x16384_counter_collection.csv" 200 256 16384 "$out/pmc_traffic_256x16384.json"
[ -f "$out/fetch_1024x8192_up4_counter_collection.csv" ] && [ -f "$out/write_1024x8192_up4_counter_collection.csv" ] && \
    python3 tools/pmc_traffic.py "$out/fetch_1024x8192_up4_counter_collection.csv" "$out/write_1024x8192_up4_counter_collection.csv" 100 1024 8192 "$out/pmc_traffic_1024x8192_up4.json" 4
[ -f "$out/fetch_1024x8192_counter_collection.csv" ] && [ -f "$out/write_1024x8192_counter_collection.csv" ] && \
    python3 tools/pmc_traffic.py "$out/fetch_1024x8192_counter_collection.csv" "$out/write_1024x8192_counter_collection.csv" 100 1024 8192 "$out/pmc_traffic_1024x8192.json"
for s in 256x4096 256x16384 1024x8192_up4; do
    [ -f "$out/sq_${s}_counter_collection.csv" ] && python3 tools/pmc_summary.py "$out/sq_${s}_counter_collection.csv" > "$out/sq_counters_$s.csv"
done
ls -la "$out"
