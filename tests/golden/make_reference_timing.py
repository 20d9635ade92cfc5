"""Times the reference's own functions on a small synthetic frame, in the build container.

Run:  python tests/golden/make_reference_timing.py
The reference's Python cannot travel to the GPU box, so its speed is measured here, once, on frames from the same
generator the benchmark uses (slamhip/synth.py), through the in-memory functions of its three scripts:
_read_bytes + _parse_groups, the group loop of process_excel_filtered, and map_angles .. generate_heatmap.
Writes profiles/reference_cpu_timing.json (per-record costs and the time they imply for one 256 x 4096 frame).
"""
import json
import logging
import os
import platform
import sys
import tempfile
import time
import warnings
from pathlib import Path

import matplotlib

matplotlib.use("Agg")
warnings.filterwarnings("ignore")
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "slam-process_amd"))
sys.path.insert(0, "/root/reference")
from slamhip import synth  # noqa: E402
import serial_hex_to_excel_v3 as S  # noqa: E402
import bs_beam_correction as B  # noqa: E402
import excel_heatmap_v3 as E  # noqa: E402


def main():
    rows_b, n_rec = 16, 4096                     # 16 sweeps of 4096 records: groups as long as the benchmark's
    lg = logging.getLogger("timing")
    lg.addHandler(logging.NullHandler())
    lg.propagate = False
    logging.getLogger().setLevel(logging.ERROR)
    text = synth.make_frame_text(rows_b, n_rec, 0).tobytes()
    with tempfile.TemporaryDirectory() as d:
        p = Path(d) / "f.txt"
        p.write_bytes(text)
        t0 = time.perf_counter()
        by = S._read_bytes(p, lg)
        t1 = time.perf_counter()
        rows, valid, disc = S._parse_groups(by, lg)
        t2 = time.perf_counter()
    df = pd.DataFrame(rows, columns=[B.FLAG_COL, B.UE_BEAM_COL, B.BS_BEAM_COL, B.RSS_COL, B.CLK_COL])
    t3 = time.perf_counter()
    df = B._coerce_types(df)
    df["_gid"] = B._detect_groups(df)
    pieces = []
    for g, sub in df.groupby("_gid", sort=True):
        f = B._filter_group_corrected(sub, B._identify_baselines(sub))
        if not f.empty:
            pieces.append(f)
    res = pd.concat(pieces, ignore_index=True)
    t4 = time.perf_counter()
    mapping = {int(k): float(v) for k, v in json.load(open(os.path.join(HERE, "beam_angle.json"))).items()}
    main_df = pd.DataFrame({"ue_beam": res[B.OUT_UE], "bs_beam": res[B.OUT_BS], "rss": res[B.OUT_RSS].astype(float)})
    t5 = time.perf_counter()
    dfm = E.map_angles(main_df, mapping)
    mat, aod, aoa = E.compute_rssi_matrix(dfm)
    t6 = time.perf_counter()
    with tempfile.TemporaryDirectory() as d:
        E.generate_heatmap(mat, aod, aoa, os.path.join(d, "h.png"), "t", use_log=False, blur_sigma=1.0)
    t7 = time.perf_counter()
    n = rows_b * n_rec
    per = dict(read_bytes_us_per_record=(t1 - t0) / n * 1e6, parse_groups_us_per_record=(t2 - t1) / n * 1e6,
               correction_us_per_row=(t4 - t3) / valid * 1e6, map_and_pivot_us_per_kept_row=(t6 - t5) / len(res) * 1e6,
               blur_and_figure_s_per_frame=t7 - t6)
    R = 256 * 4096
    kappa = len(res) / valid
    frame_s = (per["read_bytes_us_per_record"] + per["parse_groups_us_per_record"] + per["correction_us_per_row"]
               + kappa * per["map_and_pivot_us_per_kept_row"]) * R / 1e6 + per["blur_and_figure_s_per_frame"]
    doc = dict(what="the reference's own functions, one thread, measured in the build container (not on the GPU box)",
               sample=f"{rows_b} x {n_rec} synthetic records ({valid} valid, {len(res)} kept)", cpu=platform.processor() or platform.machine(),
               python=platform.python_version(), pandas=pd.__version__, per_unit=per,
               implied_seconds_per_256x4096_frame=frame_s, implied_frames_per_second_one_thread=1.0 / frame_s)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(doc, open(os.path.join(ROOT, "profiles", "reference_cpu_timing.json"), "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
