"""Golden results for tables that hold empty (NaN) cells, from the reference's correction drivers.

Run in the build container:  python tests/golden/make_null_tables.py
Each case is a small FLAG/UE/BS/RSS/CLK table with some cells null.  For both drivers -- the loop of
process_excel_filtered (bs_beam_correction.py:185-198, "filtered") and of process_excel (:124-134, "inplace")
-- the reference's own functions give either the output rows or the exception they raise, plus the root
logger's messages up to that point.  Also the group ids (_detect_groups :43-47).  Writes null_tables.json.
"""
import json
import logging
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import bs_beam_correction as B  # noqa: E402

COLS = [B.FLAG_COL, B.UE_BEAM_COL, B.BS_BEAM_COL, B.RSS_COL, B.CLK_COL]


class Capture(logging.Handler):
    def __init__(self):
        super().__init__(logging.DEBUG)
        self.lines = []

    def emit(self, record):
        self.lines.append(f"{record.levelname} {record.getMessage()}")


def frame(rows):
    return pd.DataFrame([[np.nan if v is None else v for v in r] for r in rows], columns=COLS, dtype=object)


def run(rows, filtered):
    df = B._coerce_types(frame(rows))
    df["_gid"] = B._detect_groups(df)
    pieces = []
    for g, sub in df.groupby("_gid", sort=True):
        bl = B._identify_baselines(sub)
        if not bl:
            logging.warning("组%d 未识别到基准值，已跳过组内所有预测行" if filtered else "组%d 未识别到基准值", g)
        out = (B._filter_group_corrected if filtered else B._correct_group)(sub, bl)
        if not filtered or not out.empty:
            pieces.append(out)
    if filtered:
        if not pieces:
            raise ValueError("未产生任何可修正数据行，输出为空")
        res = pd.concat(pieces, ignore_index=True)
        return [[int(v) for v in r] for r in res[[B.OUT_UE, B.OUT_BS, B.OUT_RSS, B.OUT_CLK]].to_numpy().tolist()]
    res = pd.concat(pieces).sort_index()
    return [None if pd.isna(v) else int(v) for v in res[B.OUT_COL].tolist()]


def base_table(rng, n_groups, per_group):
    """Groups of rising UE; handovers (FLAG 1, RSS copied) every few rows; clocks on and off the cycle grid."""
    rows = []
    for g in range(n_groups):
        c0 = int(rng.integers(1_000_000, 50_000_000))
        n = int(rng.integers(max(2, per_group // 2), per_group + 1))
        rss = int(rng.integers(15000, 90000))
        for i in range(n):
            ue = min(63, i * 64 // n)
            clk = c0 + i * B.CYCLE + int(rng.integers(-700, 700))
            if i % 4 == 3:
                rows.append([1, ue, int(rng.integers(0, 64)), rss, clk])
            else:
                rss = int(rng.integers(15000, 90000)) if i % 4 != 2 else rss
                rows.append([0, ue, 63, rss, clk])
        # make the handovers true baselines: the row before carries the same RSS
        for i in range(len(rows) - n, len(rows)):
            if rows[i][0] == 1 and i > len(rows) - n:
                rows[i][3] = rows[i - 1][3]
    return rows


def cases():
    rng = np.random.default_rng(20260126)
    out = {}
    c0 = 1_000_000
    g = [[0, 0, 10, 42, c0], [1, 1, 12, 42, c0 + 100], [0, 2, 99, 42, c0 + B.CYCLE + 50],
         [0, 3, 99, 42, c0 + 2 * B.CYCLE - 480], [0, 4, 99, 42, c0 + 3 * B.CYCLE + 600], [0, 5, 99, 42, c0 - B.CYCLE + 100]]

    def with_null(r, c):
        t = [list(x) for x in g]
        t[r][c] = None
        return t

    for r in range(len(g)):
        for c in range(5):
            out[f"unit_r{r}_c{c}"] = with_null(r, c)
    two = [list(x) for x in g] + [[0, 0, 7, 9, 5_000_000], [0, 1, 7, 9, 5_000_000 + B.CYCLE]]
    out["second_group_without_baselines"] = two
    for c in range(5):
        t = [list(x) for x in two]
        t[len(g)][c] = None
        out[f"second_group_first_row_c{c}"] = t
        t = [list(x) for x in two]
        t[len(g) - 1][c] = None
        out[f"first_group_last_row_c{c}"] = t
    t = [list(x) for x in two]
    t[2][1] = None
    t[3][1] = None
    out["two_null_ue_in_a_row"] = t
    t = [list(x) for x in two]
    t[0][3] = None
    t[1][3] = None
    out["both_rss_of_a_handover_null"] = t
    out["all_null_row"] = [list(x) for x in g[:3]] + [[None] * 5] + [list(x) for x in g[3:]]
    out["only_null_rows"] = [[None] * 5, [None] * 5]
    out["null_flag_before_handover"] = [[None, 0, 10, 42, c0], [1, 1, 12, 42, c0 + 100], [0, 2, 9, 42, c0 + B.CYCLE]]
    out["null_flag_as_handover"] = [[0, 0, 10, 42, c0], [None, 1, 12, 42, c0 + 100], [0, 2, 9, 42, c0 + B.CYCLE]]
    for k in range(40):
        rows = base_table(rng, int(rng.integers(1, 5)), int(rng.integers(4, 14)))
        n_null = int(rng.integers(1, 4))
        for _ in range(n_null):
            r = int(rng.integers(0, len(rows)))
            # mostly FLAG / CLK / BS of plain rows, which the reference tolerates; sometimes anything
            c = int(rng.choice([0, 4, 2, 0, 4, 1, 3])) if rng.random() < 0.8 else int(rng.integers(0, 5))
            rows[r][c] = None
        out[f"random_{k:02d}"] = rows
    return out


def main():
    root = logging.getLogger()
    root.setLevel(logging.DEBUG)
    doc = {}
    tally = {}
    for name, rows in cases().items():
        e = dict(rows=rows)
        root.handlers[:] = [Capture()]
        df = B._coerce_types(frame(rows))
        e["gid"] = [int(v) for v in B._detect_groups(df).tolist()]
        for key, filt in (("filtered", True), ("inplace", False)):
            cap = Capture()
            root.handlers[:] = [cap]
            try:
                e[key] = dict(result=run(rows, filt))
            except Exception as ex:  # the reference's own failure is the expected outcome
                e[key] = dict(error=type(ex).__name__, message=str(ex))
            e[key]["log"] = cap.lines
            tally[(key, "error" if "error" in e[key] else "ok")] = tally.get((key, "error" if "error" in e[key] else "ok"), 0) + 1
        doc[name] = e
    json.dump(doc, open(os.path.join(HERE, "null_tables.json"), "w", encoding="utf-8"), ensure_ascii=False)
    print(len(doc), tally)


if __name__ == "__main__":
    main()
