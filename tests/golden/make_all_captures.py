"""Every capture log of the reference's debugDoc/ (19 data files, 36 MB of text) as one fixture, with what the
reference's own functions give for each.

Run in the build container:  python tests/golden/make_all_captures.py     (about five minutes: the reference's loops)
Writes captures_all.npz:
  data_<i>      the capture's data bytes, LZMA-compressed (the text is "« " + "XX " per byte; rebuilding it gives the
                file byte for byte, checked here, digest in the index)
  blurred_<i>   gaussian_blur_nan_aware(matrix, 1.0) of the reference, float64
  index         JSON: per capture its name, text digest, valid / discarded of _parse_groups, digest of its rows, groups,
                baselines and kept rows of the process_excel_filtered loop with a digest of those rows, the axes and a
                digest of compute_rssi_matrix's matrix."""
import glob
import hashlib
import json
import logging
import lzma
import os
import sys
from pathlib import Path

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import matplotlib  # noqa: E402
matplotlib.use("Agg")
import serial_hex_to_excel_v3 as S  # noqa: E402
import bs_beam_correction as B  # noqa: E402
import excel_heatmap_v3 as E  # noqa: E402


def text_of(data: bytes) -> bytes:
    return "« ".encode("utf-8") + "".join(f"{b:02X} " for b in data).encode("ascii")


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    lg = logging.getLogger("all")
    lg.addHandler(logging.NullHandler())
    lg.propagate = False
    logging.disable(logging.CRITICAL)
    mapping = {int(k): float(v) for k, v in json.load(open(os.path.join(HERE, "beam_angle.json"))).items()}
    out, index = {}, []
    for i, p in enumerate(sorted(glob.glob("/root/reference/debugDoc/*.txt"))):
        raw = open(p, "rb").read()
        by = S._read_bytes(Path(p), lg)
        assert text_of(bytes(by)) == raw, p
        rows, valid, disc = S._parse_groups(by, lg)
        arr = np.array(rows, dtype=np.int64)
        df = B._coerce_types(pd.DataFrame(rows, columns=[B.FLAG_COL, B.UE_BEAM_COL, B.BS_BEAM_COL, B.RSS_COL, B.CLK_COL]))
        df["_gid"] = B._detect_groups(df)
        pieces, n_bl = [], 0
        for g, sub in df.groupby("_gid", sort=True):
            bl = B._identify_baselines(sub)
            n_bl += len(bl)
            f = B._filter_group_corrected(sub, bl)
            if not f.empty:
                pieces.append(f)
        kept_df = pd.concat(pieces, ignore_index=True)
        kept = kept_df.to_numpy().astype(np.int64)
        main_df = pd.DataFrame({"ue_beam": kept[:, 0], "bs_beam": kept[:, 1], "rss": kept[:, 2].astype(np.float64)})
        mat, aod, aoa = E.compute_rssi_matrix(E.map_angles(main_df, mapping))
        blurred = E.gaussian_blur_nan_aware(mat, 1.0)
        out[f"data_{i}"] = np.frombuffer(lzma.compress(bytes(by), preset=9 | lzma.PRESET_EXTREME), np.uint8)
        out[f"blurred_{i}"] = np.asarray(blurred, np.float64)
        index.append(dict(name=os.path.basename(p), text_sha256=hashlib.sha256(raw).hexdigest(), n_text=len(raw),
                          valid=valid, discarded=disc, rows_sha256=sha(arr), groups=int(df["_gid"].max()) + 1,
                          baselines=n_bl, kept=len(kept), kept_sha256=sha(kept), aod=[float(v) for v in aod],
                          aoa=[float(v) for v in aoa], matrix_sha256=sha(np.asarray(mat, np.float64))))
        print(i, index[-1]["name"], valid, disc, index[-1]["groups"], n_bl, len(kept), mat.shape, flush=True)
    out["index"] = np.frombuffer(json.dumps(index, ensure_ascii=False).encode("utf-8"), np.uint8)
    path = os.path.join(HERE, "captures_all.npz")
    np.savez(path, **out)
    print(os.path.getsize(path))


if __name__ == "__main__":
    main()
