#!/usr/bin/env python3
"""Experiment: the correction kernel on rows that are still in the Infinity Cache against rows that come from HBM.

Stage 2 (correction) of a small batch is run over and over on the same decoded rows: with F frames of 256 x 4096 the
rows it reads (10 MB per frame) and the kept rows it writes (6 MB per frame) stay within the 256 MiB cache for small F.
Prints the correction kernel's time per frame for several F; the HBM figure is the F = 400 line.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "slam-process_amd"))

import torch  # noqa: E402
from slamhip import synth  # noqa: E402
from slamhip.pipeline import FramePipeline, STAGE_CORRECT, STAGE_DECODE, synth_frames  # noqa: E402

B, N = 256, 4096
tl = synth.text_len(B, N)
stride = ((tl + 255) // 256) * 256
out = {}
for F in (4, 6, 8, 10, 12, 16, 24, 400):
    text = synth_frames(F, B, N)
    pipe = FramePipeline(F, tl, tile_w=64, tile_h=64)
    pipe.set_uniform_layout(F, stride, tl)
    pipe.run(text, stages=STAGE_DECODE | STAGE_CORRECT)
    torch.cuda.synchronize()
    for _ in range(3):
        pipe.run(text, stages=STAGE_CORRECT)
    torch.cuda.synchronize()
    pipe.ctx.timing_reset()
    pipe.ctx.timing(True)
    reps = 20 if F < 100 else 5
    for _ in range(reps):
        pipe.run(text, stages=STAGE_CORRECT)
    pipe.ctx.timing(False)
    t = pipe.ctx.timing_read()
    us = {k: round(v[1] / v[0] / F * 1e3, 3) for k, v in t.items()}
    out[F] = dict(rows_finish_us_per_frame=us.get("rows_finish"), tile_bits_us_per_frame=us.get("tile_bits"),
                  working_set_MB=round(F * 16.9, 1))
    del pipe, text
    torch.cuda.empty_cache()
print(json.dumps(out))
