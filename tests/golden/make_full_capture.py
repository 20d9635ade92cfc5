"""One complete capture of the reference as a fixture (the smallest one, 622 KB of text; a data file of the reference's
repository), with the counters the reference's own functions give for it.

Run in the build container:  python tests/golden/make_full_capture.py
Writes full_164520_text.npz: the raw text, valid / discarded of _parse_groups, group and kept-row counts of the
process_excel_filtered loop, and a digest of the kept rows.  Its pivot sums, matrix, blur and colours are already in
heatmap_full_164520.npz (make_golden.py)."""
import hashlib
import logging
import os
import sys
from pathlib import Path

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import serial_hex_to_excel_v3 as S  # noqa: E402
import bs_beam_correction as B  # noqa: E402

SRC = "/root/reference/debugDoc/256-0x16-Serial Debug 2026-01-26 164520.txt"


def main():
    lg = logging.getLogger("full")
    lg.addHandler(logging.NullHandler())
    lg.propagate = False
    logging.getLogger().setLevel(logging.ERROR)
    rows, valid, disc = S._parse_groups(S._read_bytes(Path(SRC), lg), lg)
    df = B._coerce_types(pd.DataFrame(rows, columns=[B.FLAG_COL, B.UE_BEAM_COL, B.BS_BEAM_COL, B.RSS_COL, B.CLK_COL]))
    df["_gid"] = B._detect_groups(df)
    pieces, n_bl = [], 0
    for g, sub in df.groupby("_gid", sort=True):
        bl = B._identify_baselines(sub)
        n_bl += len(bl)
        f = B._filter_group_corrected(sub, bl)
        if not f.empty:
            pieces.append(f)
    kept = pd.concat(pieces, ignore_index=True).to_numpy().astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "full_164520_text.npz"), text=np.frombuffer(open(SRC, "rb").read(), np.uint8),
                        valid=valid, discarded=disc, groups=int(df["_gid"].max()) + 1, baselines=n_bl, kept=len(kept),
                        kept_sha256=hashlib.sha256(np.ascontiguousarray(kept).tobytes()).hexdigest())
    print(valid, disc, int(df["_gid"].max()) + 1, n_bl, len(kept), os.path.getsize(os.path.join(HERE, "full_164520_text.npz")))


if __name__ == "__main__":
    main()
