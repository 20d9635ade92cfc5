"""Drop-in for the reference's bs_beam_correction.py: clock-based BS beam index correction.

Same constants, module-level functions, arguments, return types, exceptions and CLI flags as the
reference script (bs_beam_correction.py:12-25 constants, :28-47 validation / grouping, :50-61
baselines, :64-116 and :140-178 correction, :119-138 and :180-205 workbook drivers, :285-308 CLI).
Group detection, baseline detection, the nearest-baseline search and the order-preserving filter run
on the GPU through libslamhip.so (slam_correct); pandas is used only to carry the tables in and out.
There is no CPU implementation behind these functions.

Empty cells (NaN after _coerce_types) behave as in the reference -- which rows they split, skip or fail
on -- by handing the device stand-in integers and settling the rest from the null masks (_encode).

What differs from the reference: workbooks are read and written with the standard library (no
openpyxl); values must be integers that fit int32 with |CLK| < 2**30.
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


_NAN_INT = "cannot convert float NaN to integer"      # what int(nan) says; the reference's failure on a null cell
_I32_MAX, _I32_MIN = 2 ** 31 - 1, -(2 ** 31)


def _fill_ue(a: np.ndarray, null: np.ndarray) -> np.ndarray:
    """UE with null cells replaced so that integer comparison reproduces the reference's rule (:45): a null
    UE never starts a group (it is above every real value) and the row after it always does (a run of
    nulls descends from the top of int32)."""
    n = len(a)
    idx = np.arange(n, dtype=np.int64)
    start = null & ~np.concatenate(([False], null[:-1]))
    run_pos = idx - np.maximum.accumulate(np.where(start, idx, 0))
    real = a[~null]
    if real.size and real.max() >= _I32_MAX - n:
        raise ValueError(f"列 {UE_BEAM_COL} 的数值过大，无法与空单元格同时处理")
    return np.where(null, _I32_MAX - run_pos, a).astype(np.float64)


def _encode(df: pd.DataFrame):
    """The five input columns as arrays for the device, and the masks of their null (NaN) cells (None when
    there are none).  Null cells get stand-in integers under which the device's integer rules give what
    the reference's NaN rules give: FLAG 2 (neither 0 nor 1, :53; not 1, :84/:144), UE as in _fill_ue, RSS
    two alternating values at the bottom of int32 (equal to nothing, :53), BS and CLK 0 -- rows whose BS
    or CLK is null are settled on the host from the masks, the way the reference's int() calls do."""
    cols = [np.asarray(df[c].to_numpy(), dtype=np.float64) if df[c].dtype.kind not in "iu" else df[c].to_numpy()
            for c in _IN_COLS]
    nulls = [np.isnan(a) if a.dtype.kind == "f" else np.zeros(len(a), bool) for a in cols]
    if not any(m.any() for m in nulls):
        return cols, None
    f, u, b, r, c = cols
    nf, nu, nb, nr, nc = nulls
    real_r = r[~nr]
    if real_r.size and real_r.min() <= _I32_MIN + 1:
        raise ValueError(f"列 {RSS_COL} 的数值过小，无法与空单元格同时处理")
    idx = np.arange(len(f), dtype=np.int64)
    filled = [np.where(nf, 2, f), _fill_ue(u, nu), np.where(nb, 0, b), np.where(nr, _I32_MIN + (idx & 1), r),
              np.where(nc, 0, c)]
    return filled, nulls


def _correct_table(df: pd.DataFrame, filtered: bool, baselines=None, one_group: bool = False):
    """One device pass over a table (or one group), then the reference's treatment of null cells.

    Adds to the device's result: `won` (rows that received a corrected BS), `kept` without the rows whose
    CLK is null (:91-94, :149-151), `baseline_fail` (a baseline's CLK or BS cell is null, :58-59) and `fail` = None or (group id, the group has no baselines) for the
    first group in which the reference would meet int(NaN): at a baseline (:58-59), at a kept row's UE or
    RSS (:164-166), or -- in the in-place driver -- at the BS of a row that keeps its own BS (:86-107)."""
    cols, nulls = _encode(df)
    res = ops.correct(*cols, CYCLE, TOL, MOD_BASE, baselines=baselines, one_group=one_group)
    res["won"] = np.asarray(res["corrected"]) >= 0
    res["fail"] = None
    res["baseline_fail"] = False
    res["b_filled"] = np.asarray(cols[2])
    if nulls is None:
        return res
    nf, nu, nb, nr, nc = nulls
    n = len(nf)
    single = baselines is not None or one_group
    gid = np.zeros(n, np.int64) if single else np.asarray(res["gid"], np.int64)
    bad_groups = []
    if baselines is None:
        # Which rows are baselines: with the row index as the clock, each baseline's clock is the index of the row before it.
        probe = ops.correct(cols[0], cols[1], cols[2], cols[3], np.arange(n, dtype=np.int64), CYCLE, TOL, MOD_BASE,
                            one_group=one_group)
        prev = np.asarray(probe["baselines"][0], np.int64)
        bl_bad = nc[prev] | nb[prev + 1]
        bad_groups.append(gid[prev[bl_bad] + 1])
        res["baseline_fail"] = bool(bl_bad.any())
    won_dev = res["won"]
    won = won_dev & ~nc
    bad_groups.append(gid[(won & (nu | nr)) if filtered else (~won & nb)])
    bad = np.concatenate(bad_groups)
    if bad.size:
        g = int(bad.min())
        n_bl = len(baselines) if baselines is not None else int(np.count_nonzero(np.asarray(res["baselines"][2]) == g))
        res["fail"] = (g, n_bl == 0)
    keep = ~nc[np.flatnonzero(won_dev)]
    res["kept"] = tuple(np.asarray(k)[keep] for k in res["kept"])
    res["won"] = won
    return res


def _raise_if_failed(res) -> None:
    if res["fail"] is not None:
        raise ValueError(_NAN_INT)


def _detect_groups(df: pd.DataFrame) -> pd.Series:
    """Group id per row: a new group wherever UE_Beam drops, and after a null UE_Beam (reference :43-47)."""
    s = df[UE_BEAM_COL]
    ue = np.asarray(s.to_numpy(), dtype=np.float64) if s.dtype.kind not in "iu" else s.to_numpy()
    if ue.dtype.kind == "f" and np.isnan(ue).any():
        ue = _fill_ue(ue, np.isnan(ue))
    n = len(ue)
    z = np.zeros(n, np.int32)
    res = ops.correct(z, ue, z, z, z, CYCLE, TOL, MOD_BASE)
    return pd.Series(res["gid"].astype(int), index=df.index)


def _identify_baselines(group: pd.DataFrame) -> List[Tuple[int, int]]:
    """(CLK of the row before, BS of the row) wherever FLAG goes 0 -> 1 with equal RSS (reference :50-61)."""
    res = _correct_table(group, True, one_group=True)
    if res["baseline_fail"]:
        raise ValueError(_NAN_INT)
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


def _corrected_column(res) -> np.ndarray:
    corr = np.asarray(res["corrected"]).astype(np.int64)
    return np.where(res["won"], corr, res["b_filled"].astype(np.int64)).astype(np.float64)


def _correct_group(group: pd.DataFrame, baselines: List[Tuple[int, int]]) -> pd.DataFrame:
    """All rows kept; Corrected_BS_Beam = the winning candidate, else the row's own BS (reference :79-116)."""
    out = group.copy()
    res = _correct_table(group, False, baselines=list(baselines))
    _raise_if_failed(res)
    out[OUT_COL] = _corrected_column(res)
    logging.info("组索引范围[%s-%s] 基准数量=%d 修正数量=%d 总行数=%d", str(group.index.min()),
                 str(group.index.max()), len(baselines), int(res["won"].sum()), len(group))
    return out


def _kept_frame(res) -> pd.DataFrame:
    ku, kb, kr, kc = res["kept"]
    return pd.DataFrame({OUT_UE: ku.astype(np.int64), OUT_BS: kb.astype(np.int64),
                         OUT_RSS: kr.astype(np.int64), OUT_CLK: kc.astype(np.int64)},
                        columns=[OUT_UE, OUT_BS, OUT_RSS, OUT_CLK])


def _filter_group_corrected(group: pd.DataFrame, baselines: List[Tuple[int, int]]) -> pd.DataFrame:
    """Only FLAG=0 rows that found a baseline, with the corrected BS (reference :140-178)."""
    res = _correct_table(group, True, baselines=list(baselines))
    _raise_if_failed(res)
    out = _kept_frame(res)
    logging.info("过滤组索引范围[%s-%s] 基准数量=%d 输出修正数量=%d", str(group.index.min()),
                 str(group.index.max()), len(baselines), len(out))
    return out


def _read_table(input_path: str) -> pd.DataFrame:
    if not os.path.exists(input_path):
        raise FileNotFoundError(f"文件不存在: {input_path}")
    return pd.DataFrame(xlsx.read_table(input_path))


def _write_table(path: str, df: pd.DataFrame) -> None:
    xlsx.write_table(path, list(df.columns), [df[c].tolist() for c in df.columns])


def _log_groups(res, filtered: bool) -> None:
    """The reference's per-group lines, in group order: a warning for a group without baselines
    (:131-132, :191-192) and the summary its per-group function logs (:108-115, :171-177).  Row indices
    are those of the default RangeIndex the reference's DataFrame has.  When a group fails on a null cell
    the lines stop where the reference's do: before that group, or after its warning."""
    gid = np.asarray(res["gid"], np.int64)
    ng = int(res["n_groups"])
    if ng == 0 or len(gid) == 0:
        return
    n_bl = np.bincount(np.asarray(res["baselines"][2], np.int64), minlength=ng)
    first = np.searchsorted(gid, np.arange(ng), side="left")
    last = np.searchsorted(gid, np.arange(ng), side="right") - 1
    n_corr = np.bincount(gid[res["won"]], minlength=ng)
    stop, stop_warns = res["fail"] if res["fail"] is not None else (ng, False)
    for g in range(ng):
        if g == stop and not stop_warns:
            return
        if n_bl[g] == 0:
            logging.warning("组%d 未识别到基准值，已跳过组内所有预测行" if filtered else "组%d 未识别到基准值", g)
        if g == stop:
            return
        if filtered:
            logging.info("过滤组索引范围[%s-%s] 基准数量=%d 输出修正数量=%d", str(first[g]), str(last[g]),
                         int(n_bl[g]), int(n_corr[g]))
        else:
            logging.info("组索引范围[%s-%s] 基准数量=%d 修正数量=%d 总行数=%d", str(first[g]), str(last[g]),
                         int(n_bl[g]), int(n_corr[g]), int(last[g] - first[g] + 1))


def _correct_frame(df: pd.DataFrame, filtered: bool):
    """The group loop of process_excel_filtered (:185-198) / process_excel (:124-134) over a coerced table, with
    its log lines; returns the kept-rows frame or the Corrected_BS_Beam column."""
    res = _correct_table(df, filtered)
    _log_groups(res, filtered)
    _raise_if_failed(res)
    if filtered:
        out = _kept_frame(res)
        if len(out) == 0:
            raise ValueError("未产生任何可修正数据行，输出为空")
        return out
    return _corrected_column(res)


def process_excel(input_path: str) -> None:
    """Adds Corrected_BS_Beam to every row and writes the workbook back in place (reference :119-138)."""
    df = _read_table(input_path)
    _validate_columns(df)
    df = _coerce_types(df)
    df[OUT_COL] = _correct_frame(df, filtered=False)
    _write_table(input_path, df)
    logging.info("已写回修正文件: %s", input_path)


def process_excel_filtered(input_path: str, output_path: Optional[str] = None) -> str:
    """Writes <base>_filtered.xlsx with the corrected FLAG=0 rows of all groups in order (reference :180-205)."""
    df = _read_table(input_path)
    _validate_columns(df)
    df = _coerce_types(df)
    out = _correct_frame(df, filtered=True)
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
