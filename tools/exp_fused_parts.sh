#!/bin/bash
# What each part of the fused kernel's second phase costs: the default library (both modes), then tuning builds of
# k_fused.hip with parts left out (SLAM_EXTRA_FLAGS=-DFU_EXP=n, built as slamhip/libslamhip_exp<n>.so; their
# results are wrong, only the times count).  Prints rate, step time and the fused kernel's time per run.
set -e
run() {
  echo "start $3" >&2
  SLAMHIP_LIB=$1 timeout -k 10 ${EXP_TIMEOUT:-100} python bench.py --fused $2 --cpu-frames 0 --no-other-configs --no-pcie-fed --steps 10 --warmup 2 --frames-per-step 800 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['kernels']
print('$3', round(j['value'],1), j['ms_per_step'], {n:round(v['avg_ms'],3) for n,v in k.items() if v['avg_ms']>0.1}, j['config']['decode_correct'])
"
}
run "" 0 split
run "" 1 fused
for e in "$@"; do run $PWD/slam-process_amd/slamhip/libslamhip_exp$e.so 1 exp$e; done
