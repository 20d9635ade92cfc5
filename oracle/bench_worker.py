"""CPU-baseline worker of bench.py (test infrastructure, like everything under oracle/; never on a product path).

    python oracle/bench_worker.py B N first_frame n_frames tile [dump.npz]

Regenerates n_frames synthetic frames with the NumPy generator (byte-identical to the device generator), then
times the CPU oracle over them — hex unpack, sequential record scan, brute-force baseline search, pivot, blur,
raster — on this one process / one core, and prints {"n": n_frames, "dt": seconds} as JSON.  Given a dump path it
also leaves its first frame's results there (after the clock), for the bench to hold the device's against.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "slam-process_amd"))


def main():
    B, N, first, n, tile = (int(x) for x in sys.argv[1:6])
    import numpy as np
    from oracle import slam_oracle as O
    from slamhip import synth
    with open(os.path.join(ROOT, "slam-process_amd", "slamhip", "beam_angle.json")) as fh:
        mapping = {int(k): float(v) for k, v in json.load(fh).items()}
    lut = np.load(os.path.join(ROOT, "slam-process_amd", "slamhip", "viridis_lut.npy"))
    texts = [synth.make_frame_text(B, N, first + i).tobytes() for i in range(n)]
    first_out = None
    t0 = time.perf_counter()
    for t in texts:
        by = O.read_bytes(t)
        rows, v, d = O.parse_groups(by)
        kept, ng, nb = O.filter_corrected(rows["flag"], rows["ue"], rows["bs"], rows["rss"], rows["clk"])
        mat, aod, aoa, s, c = O.pivot_mean(kept["ue"], kept["bs"], kept["rss"], mapping)
        r = O.render(mat, aod, aoa, lut, use_log=False, blur_sigma=1.0, W=tile, H=tile)
        if first_out is None:
            first_out = (rows, v, d, kept, ng, nb, mat, r)
    dt = time.perf_counter() - t0
    if len(sys.argv) > 6 and first_out is not None:
        rows, v, d, kept, ng, nb, mat, r = first_out
        np.savez_compressed(sys.argv[6], frame=first, counters=np.array([v, d, ng, nb, len(kept["ue"])], np.int64),
                            rows=np.stack([rows[k].astype(np.int64) for k in ("flag", "ue", "bs", "rss", "clk")], 1),
                            kept=np.stack([kept[k].astype(np.int64) for k in ("ue", "bs", "rss", "clk")], 1),
                            matrix=mat, blurred=r["blurred"], vmin=r["vmin"], vmax=r["vmax"], tile=r["tile"])
    print(json.dumps(dict(n=n, dt=dt)))


if __name__ == "__main__":
    main()
