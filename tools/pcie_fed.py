"""Host-fed rate of the batched path: text in pinned host memory, copied to the GPU batch by batch while the previous
batch is processed (two device buffers, copy stream + compute stream).  Not the benchmark's `value` (that one has the
text resident in HBM); this is the number the PCIe link allows.   python tools/pcie_fed.py [frames_per_batch] [batches]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-process_amd"))
import torch  # noqa: E402
from slamhip import synth  # noqa: E402
from slamhip.pipeline import FramePipeline, synth_frames  # noqa: E402


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    B, N = 256, 4096
    tl = synth.text_len(B, N)
    stride = ((tl + 255) // 256) * 256
    dev_text = synth_frames(F, B, N)
    host = torch.empty(dev_text.numel(), dtype=torch.uint8, pin_memory=True)
    host.copy_(dev_text)
    bufs = [torch.empty_like(dev_text) for _ in range(2)]
    pipe = FramePipeline(F, tl)
    pipe.set_uniform_layout(F, stride, tl)
    copy_s, comp_s = torch.cuda.Stream(), torch.cuda.Stream()
    ready = [torch.cuda.Event() for _ in range(2)]
    freed = [torch.cuda.Event() for _ in range(2)]

    def run(n):
        for i in range(n):
            j = i & 1
            with torch.cuda.stream(copy_s):
                if i >= 2:
                    copy_s.wait_event(freed[j])
                bufs[j].copy_(host, non_blocking=True)
                ready[j].record(copy_s)
            with torch.cuda.stream(comp_s):
                comp_s.wait_event(ready[j])
                pipe.run(bufs[j])
                freed[j].record(comp_s)
        torch.cuda.synchronize()

    run(3)
    t0 = time.perf_counter()
    run(K)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(5):
        bufs[0].copy_(host, non_blocking=True)
    torch.cuda.synchronize()
    link = 5 * host.numel() / (time.perf_counter() - t1) / 1e9
    print(json.dumps(dict(frames_per_s_host_fed=round(F * K / dt, 1), h2d_GBps=round(link, 1), frames_per_batch=F, batches=K,
                          text_bytes_per_frame=tl)))


if __name__ == "__main__":
    main()
