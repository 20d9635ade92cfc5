"""Golden per-group log lines of the reference's correction drivers (bs_beam_correction.py:125-134, :186-195).

Run in the build container:  python tests/golden/make_group_log_lines.py
For every table of unit_tables.json: the root-logger messages of the loop of process_excel_filtered ("filtered")
and of process_excel ("inplace"), captured from the reference's own functions.  Writes group_log_lines.json.
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


class Capture(logging.Handler):
    def __init__(self):
        super().__init__(logging.DEBUG)
        self.lines = []

    def emit(self, record):
        self.lines.append(f"{record.levelname} {record.getMessage()}")


def run(rows, filtered):
    df = pd.DataFrame(np.array(rows, np.int64), columns=[B.FLAG_COL, B.UE_BEAM_COL, B.BS_BEAM_COL, B.RSS_COL, B.CLK_COL])
    df = B._coerce_types(df)
    df["_gid"] = B._detect_groups(df)
    for g, sub in df.groupby("_gid", sort=True):
        bl = B._identify_baselines(sub)
        if not bl:
            logging.warning("组%d 未识别到基准值，已跳过组内所有预测行" if filtered else "组%d 未识别到基准值", g)
        (B._filter_group_corrected if filtered else B._correct_group)(sub, bl)


def main():
    root = logging.getLogger()
    root.setLevel(logging.DEBUG)
    tables = json.load(open(os.path.join(HERE, "unit_tables.json")))
    # one more table: groups without any baseline, a one-row group
    tables["no_baseline_groups"] = dict(rows=[[0, 9, 1, 5, 100], [0, 3, 1, 5, 200], [0, 4, 1, 5, 300], [0, 1, 2, 6, 50],
                                              [1, 2, 7, 6, 60], [0, 3, 2, 6, 50 + 61000], [0, 0, 0, 1, 10]])
    doc = {}
    for name, t in tables.items():
        doc[name] = dict(rows=t["rows"])
        for key, filt in (("filtered", True), ("inplace", False)):
            cap = Capture()
            root.handlers[:] = [cap]
            run(t["rows"], filt)
            doc[name][key] = cap.lines
    json.dump(doc, open(os.path.join(HERE, "group_log_lines.json"), "w", encoding="utf-8"), ensure_ascii=False)
    print({k: (len(v["filtered"]), len(v["inplace"])) for k, v in doc.items()})


if __name__ == "__main__":
    main()
