#!/usr/bin/env python3
"""Experiment: what the framing kernel costs without its stores.

Run once per library build (SLAMHIP_LIB names the build): the default one, one built with -DSLAM_EXP_SKIP=1 (row
columns not written) and one with -DSLAM_EXP_SKIP=3 (row bits not written either).  Only the decode stage runs, so
the wrong rows of the measurement builds are never read.  Prints the kernel's time per frame of 256 x 4096.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "slam-process_amd"))

import torch  # noqa: E402
from slamhip import synth  # noqa: E402
from slamhip.pipeline import FramePipeline, STAGE_DECODE, synth_frames  # noqa: E402

B, N, F = 256, 4096, 400
tl = synth.text_len(B, N)
stride = ((tl + 255) // 256) * 256
text = synth_frames(F, B, N)
pipe = FramePipeline(F, tl, tile_w=64, tile_h=64)
pipe.set_uniform_layout(F, stride, tl)
for _ in range(2):
    pipe.run(text, stages=STAGE_DECODE)
torch.cuda.synchronize()
pipe.ctx.timing_reset()
pipe.ctx.timing(True)
for _ in range(8):
    pipe.run(text, stages=STAGE_DECODE)
pipe.ctx.timing(False)
t = pipe.ctx.timing_read()
print(json.dumps({"lib": os.environ.get("SLAMHIP_LIB", "default"), "frames": F,
                  "us_per_frame": {k: round(v[1] / v[0] / F * 1e3, 3) for k, v in t.items()}}))
