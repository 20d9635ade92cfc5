#!/usr/bin/env python3
"""Experiment: the step time of the default batch, step by step, from a cold start -- with and without a few seconds
of work on all host cores right before (what bench.py's CPU baseline does).  Prints the mean of every 5 steps."""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "slam-process_amd"))


def burn(sec):
    t = time.time()
    x = 0
    while time.time() - t < sec:
        x += sum(i * i for i in range(10000))
    return x


if __name__ == "__main__":
    cpu_first = len(sys.argv) > 1 and sys.argv[1] == "cpu"
    idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    if cpu_first:
        with mp.get_context("spawn").Pool(16) as pool:
            pool.map(burn, [3.0] * 16)
    import torch
    from slamhip import synth
    from slamhip.pipeline import FramePipeline, synth_frames
    B, N, F = 256, 4096, 800
    tl = synth.text_len(B, N)
    stride = ((tl + 255) // 256) * 256
    text = synth_frames(1600, B, N)
    pipe = FramePipeline(F, tl)
    pipe.set_uniform_layout(F, stride, tl)
    torch.cuda.synchronize()
    if idle:
        time.sleep(idle)
    n = 80
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        pipe.run(text, text_base_offset=(i % 2) * F * stride)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print(json.dumps({"cpu_first": cpu_first, "idle_s": idle, "first_steps_ms": [round(x, 2) for x in ms[:6]],
                      "mean_of_5": [round(sum(ms[i:i + 5]) / 5, 3) for i in range(0, n, 5)]}))
