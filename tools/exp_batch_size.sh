#!/bin/bash
# Rate against frames per step (every frame resident is processed once per step): launch tails and the small per-batch
# kernels are fixed costs per step.  usage: tools/exp_batch_size.sh 400 800 1200 1600
for F in "$@"; do
timeout -k 10 250 python bench.py --steps 10 --warmup 2 --cpu-frames 0 --no-other-configs --no-pcie-fed --frames-per-step $F --distinct-frames $((2*F)) 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($F, j['config']['frames_per_step'], j['value'], j['ms_per_step'], j['step_ms']['median'])
"
done
