#!/usr/bin/env python3
"""Throughput of the capture-log configuration (every debugDoc log as one ragged batch) on one GPU.

    python tools/captures_bench.py [--repeat 20] [--tile 1024] [--out profiles/r2_captures.json]

The 19 capture logs of the reference travel as data in tests/golden/captures_all.npz (rebuilt to their text here).
Timed: `slamhip.batch.process_texts` end to end from host bytes (pack + copy to the device + one batched call + the
small read-backs, rows not fetched, tiles left on the device), and the batched device call alone with the packed
text already resident.  Counters of every log are checked against the reference's once, before the clock.
"""
import argparse
import json
import lzma
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "slam-process_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def load():
    z = np.load(os.path.join(ROOT, "tests", "golden", "captures_all.npz"))
    index = json.loads(z["index"].tobytes().decode("utf-8"))
    texts = []
    for i, e in enumerate(index):
        data = lzma.decompress(z[f"data_{i}"].tobytes())
        texts.append("« ".encode("utf-8") + "".join(f"{b:02X} " for b in data).encode("ascii"))
        assert len(texts[-1]) == e["n_text"]
    return index, texts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=20)
    ap.add_argument("--tile", type=int, default=1024)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from slamhip import _lib, batch
    from slamhip.pipeline import FramePipeline
    index, texts = load()
    nbytes = sum(len(t) for t in texts)
    res = batch.process_texts(texts, tile=a.tile, rows=False, tiles="device")
    for e, r in zip(index, res):
        st = r["stats"]
        assert (int(st[_lib.STAT_VALID]), int(st[_lib.STAT_DISCARDED]), int(st[_lib.STAT_KEPT])) == (e["valid"], e["discarded"], e["kept"]), e["name"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.repeat):
        batch.process_texts(texts, tile=a.tile, rows=False, tiles="device")
    torch.cuda.synchronize()
    end_to_end = (time.perf_counter() - t0) / a.repeat
    # the device call alone on resident text
    arrs = [np.frombuffer(t, np.uint8) for t in texts]
    buf, offs, lens = batch._pack(arrs, torch.device("cuda", 0))
    pipe = FramePipeline(len(arrs), max(lens), tile_w=a.tile, tile_h=a.tile)
    pipe.set_layout(offs, lens)
    pipe.run(buf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.repeat):
        pipe.run(buf)
    torch.cuda.synchronize()
    device_only = (time.perf_counter() - t0) / a.repeat
    out = dict(what="the 19 capture logs of the reference (debugDoc/*.txt) as one ragged batch on one MI355X; counters "
                    "checked against the reference's before the clock",
               logs=len(texts), text_MB=round(nbytes / 1e6, 3), tile=a.tile, repeat=a.repeat,
               end_to_end=dict(seconds_per_batch=round(end_to_end, 5), logs_per_s=round(len(texts) / end_to_end, 1),
                               text_MB_per_s=round(nbytes / 1e6 / end_to_end, 1),
                               includes="host packing into pinned memory, host-to-device copy, the batched call on a pipeline kept from the first call, "
                                        "read-back of stats and matrices; rows not fetched, tiles left on the device"),
               device_call=dict(seconds_per_batch=round(device_only, 6), logs_per_s=round(len(texts) / device_only, 1),
                                text_GB_per_s=round(nbytes / 1e9 / device_only, 2),
                                includes="slam_process_frames_dev on resident text, 19 frames of 0.6 - 3.3 MB"))
    line = json.dumps(out)
    print(line)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
