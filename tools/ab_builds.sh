#!/bin/bash
# A/B of two library builds on the three benchmark shapes, on one box: the default library, then the build named by
# $1 (default slam-process_amd/slamhip/libslamhip_new.so; build it with SLAMHIP_OUT=... SLAMHIP_OBJ=... and, for a
# variant of the sources, SLAM_EXTRA_FLAGS=-D...).  Prints rate, step time and the three largest kernels per run.
# The first run on a fresh box is slower than the rest: run the script twice in one call and read the second pass.
set -e
other=${1:-$PWD/slam-process_amd/slamhip/libslamhip_new.so}
for lib in "" "$other"; do
  for shape in "--beams 1024 --samples 8192 --frames-per-step 100 --distinct-frames 100 --upscale 4" "--beams 256 --samples 16384 --frames-per-step 200 --distinct-frames 200" "--frames-per-step 800"; do
    SLAMHIP_LIB=$lib timeout -k 10 200 python bench.py $shape --cpu-frames 0 --no-other-configs --no-pcie-fed --steps 10 --warmup 2 | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('${lib:+other}' or 'default', j['config']['beams'], j['config']['samples'], round(j['value'],1), j['ms_per_step'], 'sort', k['bl_sort_small']['avg_ms'], 'finish', k['rows_finish']['avg_ms'], 'frame', k['frame_onepass']['avg_ms'])
"
  done
done
