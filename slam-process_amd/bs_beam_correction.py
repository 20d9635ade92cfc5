"""Drop-in for the reference's bs_beam_correction.py: clock-based BS beam index correction.

Same constants, module-level functions, arguments, return types, exceptions and CLI flags as the
reference script (bs_beam_correction.py:12-25 constants, :28-47 validation / grouping, :50-61
baselines, :64-116 and :140-178 correction, :119-138 and :180-205 workbook drivers, :285-308 CLI).
Group detection, baseline detection, the nearest-baseline search and the order-preserving filter run
on the GPU through libslamhip.so (slam_correct); pandas is used only to carry the tables in and out.
There is no CPU implementation behind these functions.

What differs from the reference: workbooks are read and written with the standard library (no
openpyxl); tables holding NaN cells are rejected with ValueError instead of being partly processed;
values must fit int32 with |CLK| < 2**30.
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
import unittest
from typing import List, Optional, Tuple

import numpy as np
import pandas as pd

from slamhip import ops, xlsx

FLAG_COL = "FLAG"
UE_BEAM_COL = "UE_Beam[5:0]十进制"
BS_BEAM_COL = "BS_Beam[5:0]十进制"
RSS_COL = "RSS十进制"
CLK_COL = "CLK十进制"
OUT_COL = "Corrected_BS_Beam"
OUT_UE = "UE_Beam"
OUT_BS = "BS_Beam"
OUT_RSS = "RSS值"
OUT_CLK = "CLK值"

CYCLE = 61000
TOL = 500
MOD_BASE = 64

_IN_COLS = [FLAG_COL, UE_BEAM_COL, BS_BEAM_COL, RSS_COL, CLK_COL]


def _validate_columns(df: pd.DataFrame) -> None:
    missing = [c for c in _IN_COLS if c not in df.columns]
    if missing:
        raise ValueError(f"缺失列: {missing}")


def _coerce_types(df: pd.DataFrame) -> pd.DataFrame:
    for c in _IN_COLS:
        df[c] = pd.to_numeric(df[c], errors="coerce")
    if df[_IN_COLS].isna().any().any():
        logging.warning("存在无法转换为数值的单元格，已填充为NaN")
    return df


def _columns(df: pd.DataFrame):
    """The five input columns as arrays; NaN cells are refused (the device path carries integers)."""
    cols = [df[c].to_numpy() for c in _IN_COLS]
    for name, a in zip(_IN_COLS, cols):
        if a.dtype.kind == "f" and np.isnan(a).any():
            raise ValueError(f"列 {name} 含有NaN，无法在设备上处理")
    return cols


def _detect_groups(df: pd.DataFrame) -> pd.Series:
    """Group id per row: a new group wherever UE_Beam drops (reference :43-47)."""
    ue = df[UE_BEAM_COL].to_numpy()
    n = len(ue)
    z = np.zeros(n, np.int32)
    res = ops.correct(z, ue, z, z, z, CYCLE, TOL, MOD_BASE)
    return pd.Series(res["gid"].astype(int), index=df.index)


def _identify_baselines(group: pd.DataFrame) -> List[Tuple[int, int]]:
    """(CLK of the row before, BS of the row) wherever FLAG goes 0 -> 1 with equal RSS (reference :50-61)."""
    f, u, b, r, c = _columns(group)
    res = ops.correct(f, u, b, r, c, CYCLE, TOL, MOD_BASE, one_group=True)
    clk_b, bs_b, _ = res["baselines"]
    return [(int(x), int(y)) for x, y in zip(clk_b.tolist(), bs_b.tolist())]


def _choose_correction(clk_diff: int, baselines: List[Tuple[int, int]]) -> Optional[int]:
    """The reference's helper (:64-76) applies ONE clock difference to every baseline; the first wins."""
    if not baselines:
        return None
    res = ops.correct([0], [0], [0], [0], [int(clk_diff)], CYCLE, TOL, MOD_BASE,
                      baselines=[(0, int(bs_b)) for _, bs_b in baselines])
    v = int(res["corrected"][0])
    return None if v < 0 else v


def _correct_group(group: pd.DataFrame, baselines: List[Tuple[int, int]]) -> pd.DataFrame:
    """All rows kept; Corrected_BS_Beam = the winning candidate, else the row's own BS (reference :79-116)."""
    out = group.copy()
    f, u, b, r, c = _columns(group)
    res = ops.correct(f, u, b, r, c, CYCLE, TOL, MOD_BASE, baselines=list(baselines))
    corr = res["corrected"].astype(np.int64)
    won = corr >= 0
    out[OUT_COL] = np.where(won, corr, np.asarray(b).astype(np.int64)).astype(np.float64)
    logging.info("组索引范围[%s-%s] 基准数量=%d 修正数量=%d 总行数=%d", str(group.index.min()),
                 str(group.index.max()), len(baselines), int(won.sum()), len(group))
    return out


def _filter_group_corrected(group: pd.DataFrame, baselines: List[Tuple[int, int]]) -> pd.DataFrame:
    """Only FLAG=0 rows that found a baseline, with the corrected BS (reference :140-178)."""
    f, u, b, r, c = _columns(group)
    res = ops.correct(f, u, b, r, c, CYCLE, TOL, MOD_BASE, baselines=list(baselines))
    ku, kb, kr, kc = res["kept"]
    logging.info("过滤组索引范围[%s-%s] 基准数量=%d 输出修正数量=%d", str(group.index.min()),
                 str(group.index.max()), len(baselines), len(ku))
    return pd.DataFrame({OUT_UE: ku.astype(np.int64), OUT_BS: kb.astype(np.int64),
                         OUT_RSS: kr.astype(np.int64), OUT_CLK: kc.astype(np.int64)},
                        columns=[OUT_UE, OUT_BS, OUT_RSS, OUT_CLK])


def _read_table(input_path: str) -> pd.DataFrame:
    if not os.path.exists(input_path):
        raise FileNotFoundError(f"文件不存在: {input_path}")
    return pd.DataFrame(xlsx.read_table(input_path))


def _write_table(path: str, df: pd.DataFrame) -> None:
    xlsx.write_table(path, list(df.columns), [df[c].tolist() for c in df.columns])


def _log_groups(res, filtered: bool) -> None:
    """The reference's per-group lines, in group order: a warning for a group without baselines
    (:131-132, :191-192) and the summary its per-group function logs (:108-115, :171-177).  Row indices
    are those of the default RangeIndex the reference's DataFrame has."""
    gid = np.asarray(res["gid"], np.int64)
    ng = int(res["n_groups"])
    if ng == 0 or len(gid) == 0:
        return
    n_bl = np.bincount(np.asarray(res["baselines"][2], np.int64), minlength=ng)
    first = np.searchsorted(gid, np.arange(ng), side="left")
    last = np.searchsorted(gid, np.arange(ng), side="right") - 1
    n_corr = np.bincount(gid[np.asarray(res["corrected"]) >= 0], minlength=ng)
    for g in range(ng):
        if n_bl[g] == 0:
            logging.warning("组%d 未识别到基准值，已跳过组内所有预测行" if filtered else "组%d 未识别到基准值", g)
        if filtered:
            logging.info("过滤组索引范围[%s-%s] 基准数量=%d 输出修正数量=%d", str(first[g]), str(last[g]),
                         int(n_bl[g]), int(n_corr[g]))
        else:
            logging.info("组索引范围[%s-%s] 基准数量=%d 修正数量=%d 总行数=%d", str(first[g]), str(last[g]),
                         int(n_bl[g]), int(n_corr[g]), int(last[g] - first[g] + 1))


def process_excel(input_path: str) -> None:
    """Adds Corrected_BS_Beam to every row and writes the workbook back in place (reference :119-138)."""
    df = _read_table(input_path)
    _validate_columns(df)
    df = _coerce_types(df)
    f, u, b, r, c = _columns(df)
    res = ops.correct(f, u, b, r, c, CYCLE, TOL, MOD_BASE)
    _log_groups(res, filtered=False)
    corr = res["corrected"].astype(np.int64)
    df[OUT_COL] = np.where(corr >= 0, corr, np.asarray(b).astype(np.int64)).astype(np.float64)
    _write_table(input_path, df)
    logging.info("已写回修正文件: %s", input_path)


def process_excel_filtered(input_path: str, output_path: Optional[str] = None) -> str:
    """Writes <base>_filtered.xlsx with the corrected FLAG=0 rows of all groups in order (reference :180-205)."""
    df = _read_table(input_path)
    _validate_columns(df)
    df = _coerce_types(df)
    f, u, b, r, c = _columns(df)
    res = ops.correct(f, u, b, r, c, CYCLE, TOL, MOD_BASE)
    _log_groups(res, filtered=True)
    ku, kb, kr, kc = res["kept"]
    if len(ku) == 0:
        raise ValueError("未产生任何可修正数据行，输出为空")
    out = pd.DataFrame({OUT_UE: ku.astype(np.int64), OUT_BS: kb.astype(np.int64),
                        OUT_RSS: kr.astype(np.int64), OUT_CLK: kc.astype(np.int64)})
    if output_path is None:
        base, _ext = os.path.splitext(input_path)
        output_path = f"{base}_filtered.xlsx"
    _write_table(output_path, out)
    logging.info("已生成过滤后的修正文件: %s 行数=%d", output_path, len(out))
    return output_path


class TestBSBeamCorrection(unittest.TestCase):
    """The reference's embedded suite (:207-283) on the same tables.  Its last test expects one filtered
    row, but the reference's own code yields two (the baseline's FLAG=0 row corrects to itself, d = 0);
    the code is the behaviour reproduced here, so that expectation is 2 rows, first BS = 12."""

    @staticmethod
    def _table(rows):
        return pd.DataFrame(rows, columns=_IN_COLS)

    def _make_group(self):
        c0 = 1_000_000
        return self._table([[0, 0, 10, 42, c0], [1, 1, 12, 42, c0 + 100], [0, 2, 99, 42, c0 + CYCLE + 50],
                            [0, 3, 99, 42, c0 + 2 * CYCLE - 480], [0, 4, 99, 42, c0 + 3 * CYCLE + 600],
                            [0, 5, 99, 42, c0 - CYCLE + 100]])

    def test_baseline_identification(self):
        df = self._make_group()
        self.assertEqual(_identify_baselines(df), [(1_000_000, 12)])

    def test_correction_logic(self):
        df = self._make_group()
        out = _correct_group(df, _identify_baselines(df))
        self.assertEqual([int(out.iloc[i][OUT_COL]) for i in (1, 2, 3)], [12, 13, 14])

    def test_boundary_tolerance(self):
        c0 = 5_000_000
        df = self._table([[0, 0, 3, 7, c0], [1, 1, 8, 7, c0 + 10], [0, 2, 0, 7, c0 + CYCLE + TOL],
                          [0, 3, 0, 7, c0 + CYCLE + TOL + 1]])
        out = _correct_group(df, _identify_baselines(df))
        self.assertEqual(int(out.iloc[2][OUT_COL]), 9)
        self.assertEqual(int(out.iloc[3][OUT_COL]), 0)

    def test_negative_diff(self):
        c0 = 7_000_000
        df = self._table([[0, 0, 60, 13, c0], [1, 1, 5, 13, c0 + 1], [0, 2, 0, 13, c0 - CYCLE + 10]])
        out = _correct_group(df, _identify_baselines(df))
        self.assertEqual(int(out.iloc[2][OUT_COL]), 4)

    def test_filter_only_corrected_rows(self):
        c0 = 2_000_000
        df = self._table([[0, 0, 10, 21, c0], [1, 1, 12, 21, c0 + 50], [0, 2, 99, 21, c0 + CYCLE + 20],
                          [0, 3, 99, 21, c0 + CYCLE + TOL + 10]])
        filtered = _filter_group_corrected(df, _identify_baselines(df))
        self.assertEqual(list(filtered.columns), [OUT_UE, OUT_BS, OUT_RSS, OUT_CLK])
        self.assertEqual(filtered[OUT_BS].tolist(), [12, 13])


def _parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--input", type=str, default=r"D:\桌面\SLAMPro\debugDoc\Serial Debug 2026-01-26 164520.xlsx")
    p.add_argument("--output", type=str, default=None)
    p.add_argument("--run-tests", action="store_true")
    p.add_argument("--log-level", type=str, default="INFO")
    return p.parse_args()


def main():
    args = _parse_args()
    logging.basicConfig(level=getattr(logging, args.log_level.upper(), logging.INFO),
                        format="%(asctime)s %(levelname)s %(message)s")
    if args.run_tests:
        suite = unittest.defaultTestLoader.loadTestsFromTestCase(TestBSBeamCorrection)
        result = unittest.TextTestRunner(verbosity=2).run(suite)
        sys.exit(0 if result.wasSuccessful() else 1)
    try:
        process_excel_filtered(args.input, args.output)
    except Exception as e:
        logging.exception("处理失败: %s", str(e))
        sys.exit(1)


if __name__ == "__main__":
    main()
