"""Pins the CPU oracle on the reference itself (build container only: skipped where /root/reference is absent).

1. the reference's own functions, imported, on full captures;
2. the 19 committed stage outputs debugDoc/<name>.xlsx and <name>_filtered.xlsx (golden vectors).
"""
import glob
import logging
import os
import sys

import numpy as np
import pytest

from conftest import REFERENCE, needs_reference
from oracle import slam_oracle as O

pytestmark = needs_reference
D = os.path.join(REFERENCE, "debugDoc")


def _logs():
    return sorted(glob.glob(os.path.join(D, "*.txt")))


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, REFERENCE)
    import matplotlib
    matplotlib.use("Agg")
    import serial_hex_to_excel_v3 as S
    sys.path.remove(REFERENCE)
    logging.disable(logging.CRITICAL)
    yield S
    logging.disable(logging.NOTSET)
    for m in ("serial_hex_to_excel_v3",):
        sys.modules.pop(m, None)


def test_all_committed_stage_outputs():
    from slamhip import xlsx
    logs = _logs()
    assert len(logs) == 19
    for p in logs:
        raw = open(p, "rb").read()
        by = O.read_bytes(raw)
        rows, v, d = O.parse_groups(by)
        mine = np.stack([rows[k].astype(np.int64) for k in ("flag", "ue", "bs", "rss", "clk")], 1)
        base = os.path.basename(p).replace("256-0x16-", "")[:-4]
        tab = xlsx.read_table(os.path.join(D, base + ".xlsx"))
        gold = np.array([tab["FLAG"], tab["UE_Beam[5:0]十进制"], tab["BS_Beam[5:0]十进制"], tab["RSS十进制"],
                         tab["CLK十进制"]], dtype=np.int64).T
        off = len(mine) - len(gold)          # 170305.xlsx had its first 9 rows deleted by hand
        assert off in (0, 9) and np.array_equal(mine[off:], gold), base
        kept, ng, nb = O.filter_corrected(*mine.T)
        ft = xlsx.read_table(os.path.join(D, base + "_filtered.xlsx"))
        rk = [k for k in ft if k.startswith("RSS")][0]
        gf = np.array([ft["UE_Beam"], ft["BS_Beam"], ft[rk], ft["CLK值"]], dtype=np.int64).T
        assert np.array_equal(np.stack([kept[k] for k in ("ue", "bs", "rss", "clk")], 1), gf), base


def test_reference_functions_on_full_captures(ref):
    from pathlib import Path
    lg = logging.getLogger("pin")
    for p in _logs()[:2] + _logs()[-1:]:
        by_ref = ref._read_bytes(Path(p), lg)
        by = O.read_bytes(open(p, "rb").read())
        assert by.tolist() == by_ref
        rows_ref, v_ref, d_ref = ref._parse_groups(by_ref, lg)
        rows, v, d = O.parse_groups(by)
        assert (v, d) == (v_ref, d_ref)
        arr = np.array(rows_ref, dtype=np.int64)
        for j, k in enumerate(("flag", "ue", "bs", "rss", "clk")):
            assert np.array_equal(rows[k].astype(np.int64), arr[:, j])


@pytest.mark.parametrize("shape", [(8, 1024, 3), (3, 4096, 0), (1, 16384, 5)])
def test_reference_functions_on_a_synthetic_frame(ref, tmp_path, shape):
    """The benchmark's kind of data (slamhip/synth.py, 8 sweeps of 1024 records) through the reference's own
    _read_bytes, _parse_groups and the group loop of process_excel_filtered, against the oracle: the frames the
    throughput is measured on are covered by the same pin as the captures."""
    import importlib.util
    import pandas as pd
    from pathlib import Path
    from slamhip import synth
    spec = importlib.util.spec_from_file_location("ref_bs_beam_correction_syn", os.path.join(REFERENCE, "bs_beam_correction.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    lg = logging.getLogger("pin")
    B_rows, N_rec, frame_id = shape        # the second shape has the benchmark's sweep length: 63 baselines per group
    text = synth.make_frame_text(B_rows, N_rec, frame_id).tobytes()
    p = tmp_path / "f.txt"
    p.write_bytes(text)
    by_ref = ref._read_bytes(Path(p), lg)
    rows_ref, v_ref, d_ref = ref._parse_groups(by_ref, lg)
    by = O.read_bytes(text)
    rows, v, d = O.parse_groups(by)
    assert by.tolist() == by_ref and (v, d) == (v_ref, d_ref)
    arr = np.array(rows_ref, dtype=np.int64)
    for j, k in enumerate(("flag", "ue", "bs", "rss", "clk")):
        assert np.array_equal(rows[k].astype(np.int64), arr[:, j])
    df = B._coerce_types(pd.DataFrame(rows_ref, columns=[B.FLAG_COL, B.UE_BEAM_COL, B.BS_BEAM_COL, B.RSS_COL, B.CLK_COL]))
    df["_gid"] = B._detect_groups(df)
    pieces = []
    for g, sub in df.groupby("_gid", sort=True):
        f = B._filter_group_corrected(sub, B._identify_baselines(sub))
        if not f.empty:
            pieces.append(f)
    res = pd.concat(pieces, ignore_index=True).to_numpy().astype(np.int64)
    kept, ng, nb = O.filter_corrected(*arr.T)
    assert ng == int(df["_gid"].max()) + 1 == B_rows and nb >= 15 * B_rows * (N_rec // 1024)
    assert np.array_equal(np.stack([kept[k] for k in ("ue", "bs", "rss", "clk")], 1), res)


def test_parse_and_correction_fuzz_against_the_reference(ref, tmp_path):
    """Random streams from the record grammar the GPU fuzz uses (sweeps, handovers, damaged records, stray flag bytes,
    noise, record-free stretches, cuts) through the reference's own _read_bytes, _parse_groups and the group loop of
    process_excel_filtered, against the oracle: the generator the HIP path is fuzzed with is covered by the same pin."""
    import importlib.util
    import pandas as pd
    from pathlib import Path
    from test_gpu_parity import _grammar_stream
    spec = importlib.util.spec_from_file_location("ref_bs_beam_correction_fuzz", os.path.join(REFERENCE, "bs_beam_correction.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    lg = logging.getLogger("pin")
    rng = np.random.default_rng(31337)
    n_rows = n_kept = 0
    for trial in range(int(os.environ.get("SLAMHIP_FUZZ_REF_STREAMS", "40"))):
        stream = _grammar_stream(rng)
        sep = [" ", " ", "\n", "  ", "\t"]
        text = ("« " if rng.random() < 0.5 else "") + "".join(
            (f"{b:02X}" if rng.random() < 0.98 else f"{b:02x}") + sep[int(rng.integers(0, len(sep)))] for b in stream)
        if rng.random() < 0.3:
            text += "zz 1 123 0x41 G7 "                                  # tokens the reader skips
        p = tmp_path / f"s{trial}.txt"
        p.write_bytes(text.encode("utf-8"))
        by_ref = ref._read_bytes(Path(p), lg)
        by = O.read_bytes(text.encode("utf-8"))
        assert by.tolist() == by_ref, trial
        rows_ref, v_ref, d_ref = ref._parse_groups(by_ref, lg)
        rows, v, d = O.parse_groups(by)
        assert (v, d) == (v_ref, d_ref), trial
        if not v:
            continue
        arr = np.array(rows_ref, dtype=np.int64)
        for j, k in enumerate(("flag", "ue", "bs", "rss", "clk")):
            assert np.array_equal(rows[k].astype(np.int64), arr[:, j]), (trial, k)
        df = B._coerce_types(pd.DataFrame(rows_ref, columns=[B.FLAG_COL, B.UE_BEAM_COL, B.BS_BEAM_COL, B.RSS_COL, B.CLK_COL]))
        df["_gid"] = B._detect_groups(df)
        pieces, nb_ref = [], 0
        for g, sub in df.groupby("_gid", sort=True):
            bl = B._identify_baselines(sub)
            nb_ref += len(bl)
            f = B._filter_group_corrected(sub, bl)
            if not f.empty:
                pieces.append(f)
        kept, ng, nb = O.filter_corrected(*arr.T)
        assert (ng, nb) == (int(df["_gid"].max()) + 1, nb_ref), trial
        res = pd.concat(pieces, ignore_index=True).to_numpy().astype(np.int64) if pieces else np.zeros((0, 4), np.int64)
        assert np.array_equal(np.stack([kept[k] for k in ("ue", "bs", "rss", "clk")], 1), res), trial
        n_rows += v
        n_kept += len(res)
    assert n_rows > 5000 and n_kept > 500, (n_rows, n_kept)


def test_blur_against_the_reference_for_wide_and_narrow_kernels():
    """gaussian_blur_nan_aware of the reference on small, non-square matrices with holes, kernels from 3 x 3 to wider
    than the matrix (edge padding repeats the border) against the oracle's restatement."""
    import importlib.util
    import matplotlib
    matplotlib.use("Agg")
    spec = importlib.util.spec_from_file_location("ref_excel_heatmap_v3_blur", os.path.join(REFERENCE, "excel_heatmap_v3.py"))
    E = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(E)
    rng = np.random.default_rng(8)
    for shape in ((1, 1), (1, 9), (3, 3), (5, 12), (17, 4)):
        for sigma in (0.2, 0.5, 1.0, 3.0, 5.0):
            m = rng.integers(15000, 93000, shape).astype(np.float64)
            m[rng.random(shape) < 0.3] = np.nan
            want = E.gaussian_blur_nan_aware(m.copy(), sigma)
            got = O.blur_nan_aware(m, sigma)
            assert np.array_equal(np.isnan(got), np.isnan(want)), (shape, sigma)
            fin = np.isfinite(want)
            if fin.any():
                assert np.max(np.abs(got[fin] - want[fin]) / np.abs(want[fin])) <= 1e-13, (shape, sigma)
            assert np.allclose(O.gaussian_kernel(sigma), E.gaussian_kernel(sigma), rtol=1e-15, atol=0)


def test_null_rules_fuzz_against_the_reference():
    """The oracle's restatement of the empty-cell rules against the reference's own drivers on 200 random tables
    (the same generator the GPU fuzz uses): result rows, Corrected_BS_Beam column or the exception's text."""
    import importlib.util
    import pandas as pd
    spec = importlib.util.spec_from_file_location("ref_bs_beam_correction_nullfuzz", os.path.join(REFERENCE, "bs_beam_correction.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    cols = [B.FLAG_COL, B.UE_BEAM_COL, B.BS_BEAM_COL, B.RSS_COL, B.CLK_COL]
    rng = np.random.default_rng(404)

    def reference(rows, filtered):
        df = B._coerce_types(pd.DataFrame([[np.nan if v is None else v for v in r] for r in rows], columns=cols, dtype=object))
        gid = B._detect_groups(df)
        df["_gid"] = gid
        try:
            pieces = []
            for g, sub in df.groupby("_gid", sort=True):
                out = (B._filter_group_corrected if filtered else B._correct_group)(sub, B._identify_baselines(sub))
                if not filtered or not out.empty:
                    pieces.append(out)
            if filtered:
                if not pieces:
                    raise ValueError("未产生任何可修正数据行，输出为空")
                res = pd.concat(pieces, ignore_index=True)
                return gid.tolist(), dict(result=[[int(v) for v in r] for r in res.to_numpy().tolist()])
            res = pd.concat(pieces).sort_index()
            return gid.tolist(), dict(result=[int(v) for v in res[B.OUT_COL].tolist()])
        except ValueError as ex:
            return gid.tolist(), dict(error="ValueError", message=str(ex))

    for trial in range(200):
        rows = []
        for g in range(int(rng.integers(1, 5))):
            n = int(rng.integers(2, 18))
            c0 = int(rng.integers(1_000_000, 50_000_000))
            rss = int(rng.integers(15000, 90000))
            for i in range(n):
                fl = int(rng.random() < 0.25)
                if not fl or rng.random() < 0.5:
                    rss = int(rng.integers(15000, 90000)) if rng.random() < 0.6 else rss
                rows.append([fl, min(63, i * 64 // n), int(rng.integers(0, 64)), rss, c0 + i * B.CYCLE + int(rng.integers(-700, 700))])
        for _ in range(int(rng.integers(1, 5))):
            rows[int(rng.integers(0, len(rows)))][int(rng.integers(0, 5))] = None
        for filtered in (True, False):
            assert O.correct_table_nulls_py(rows, filtered) == reference(rows, filtered), (trial, filtered)


def test_heatmap_command_line_options_equal_the_references():
    """build_arg_parser of the drop-in against the reference's: the same options, destinations, types and defaults
    (help texts are free), and the same '参数:' log line for one parsed command line."""
    import importlib.util
    import matplotlib
    matplotlib.use("Agg")
    from conftest import PKG

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    R = load(os.path.join(REFERENCE, "excel_heatmap_v3.py"), "ref_excel_heatmap_v3_cli")
    M = load(os.path.join(PKG, "excel_heatmap_v3.py"), "own_excel_heatmap_v3_cli")

    def options(parser):
        return sorted((tuple(a.option_strings), a.dest, repr(a.default), getattr(a, "type", None), type(a).__name__, a.nargs)
                      for a in parser._actions)

    assert options(R.build_arg_parser()) == options(M.build_arg_parser())


def test_float_pivot_against_the_reference():
    """compute_rssi_matrix of the reference (pandas pivot_table mean) on random tables of fractional RSS against the
    oracle's restatement (compensated sums in row order): equal bit for bit, whatever the row order."""
    import importlib.util
    import warnings
    import matplotlib
    matplotlib.use("Agg")
    import pandas as pd
    spec = importlib.util.spec_from_file_location("ref_excel_heatmap_v3_fp", os.path.join(REFERENCE, "excel_heatmap_v3.py"))
    E = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(E)
    rng = np.random.default_rng(99)
    mapping = {i: -43.6 + 1.4 * i for i in range(64) if i % 9}          # some beams unmapped
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for trial in range(12):
            n = int(rng.integers(1, 4000))
            ue, bs = rng.integers(0, 64, n), rng.integers(0, 64, n)
            rss = rng.normal(-70, 20, n) * rng.choice([1.0, 1e-5, 1e7], n)
            df = pd.DataFrame({"ue_beam": ue, "bs_beam": bs, "rss": rss})
            try:
                want, aod, aoa = E.compute_rssi_matrix(E.map_angles(df, mapping))
            except ValueError:
                continue
            got, aod_o, aoa_o, s, c = O.pivot_mean_float(ue, bs, rss, mapping)
            assert aod == aod_o and aoa == aoa_o
            assert np.array_equal(np.isnan(got), np.isnan(want))
            assert np.array_equal(got[~np.isnan(got)].view(np.uint64), want[~np.isnan(want)].view(np.uint64)), trial
