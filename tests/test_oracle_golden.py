"""The CPU oracle against the committed golden fixtures (outputs of the reference's own functions)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import slam_oracle as O


@pytest.fixture(scope="module")
def mapping():
    return {int(k): v for k, v in json.load(open(os.path.join(GOLDEN, "beam_angle.json"))).items()}


@pytest.fixture(scope="module")
def lut():
    return np.load(os.path.join(GOLDEN, "viridis_lut.npy"))


def test_edge_streams():
    edge = json.load(open(os.path.join(GOLDEN, "edge_streams.json")))
    assert len(edge) >= 20
    for name, e in edge.items():
        text = bytes(e["text"])
        by = O.read_bytes(text)
        assert by.tolist() == e["bytes"], name
        assert O.read_bytes_py(text).tolist() == e["bytes"], name
        rows, v, d = O.parse_groups(by)
        assert (v, d) == (e["valid"], e["discarded"]), name
        got = np.stack([rows[k].astype(np.int64) for k in ("flag", "ue", "bs", "rss", "clk")], 1)
        assert got.tolist() == e["rows"], name
        rows_py, v2, d2 = O.parse_groups_py(by)
        assert (v2, d2) == (v, d) and [list(r) for r in rows_py] == e["rows"], name


@pytest.mark.parametrize("i", [0, 1, 2])
def test_truncated_captures(i):
    z = np.load(os.path.join(GOLDEN, f"log{i}.npz"))
    by = O.read_bytes(z["text"].tobytes())
    assert np.array_equal(by, z["bytes"])
    rows, v, d = O.parse_groups(by)
    assert (v, d) == (int(z["valid"]), int(z["discarded"]))
    r = z["rows"]
    for j, k in enumerate(("flag", "ue", "bs", "rss", "clk")):
        assert np.array_equal(rows[k].astype(np.int64), r[:, j])
    assert np.array_equal(O.detect_groups(r[:, 1]), z["gid"])
    c, b, g = O.identify_baselines(*r.T)
    assert np.array_equal(np.stack([c, b, g], 1), z["baselines"])
    kept, ng, nb = O.filter_corrected(*r.T)
    assert np.array_equal(np.stack([kept[k] for k in ("ue", "bs", "rss", "clk")], 1), z["kept"])
    ci, _ = O.correct_inplace(*r.T)
    assert np.array_equal(ci, z["corrected_inplace"])


def test_unit_tables_follow_the_code_not_the_failing_test():
    ut = json.load(open(os.path.join(GOLDEN, "unit_tables.json")))
    for name, t in ut.items():
        r = np.array(t["rows"], np.int64)
        kept, ng, nb = O.filter_corrected(*r.T)
        assert np.stack([kept[k] for k in ("ue", "bs", "rss", "clk")], 1).tolist() == t["kept"], name
        assert O.detect_groups(r[:, 1]).tolist() == t["gid"]
        c, b, g = O.identify_baselines(*r.T)
        assert np.stack([c, b, g], 1).tolist() == t["baselines"]
        assert O.correct_inplace(*r.T)[0].tolist() == t["corrected_inplace"]
    # the reference's test_filter_only_corrected_rows expects 1 row; its code returns 2 (first BS = 12)
    assert [k[1] for k in ut["filter_only"]["kept"]] == [12, 13]


def test_tokenizer_matches_cpython_on_random_bytes():
    rng = np.random.default_rng(1)
    alphabet = np.frombuffer(bytes(range(256)) + b"0123456789abcdefABCDEFx  \n\t" * 12, np.uint8)
    for _ in range(300):
        t = alphabet[rng.integers(0, len(alphabet), int(rng.integers(0, 400)))].tobytes()
        assert np.array_equal(O.read_bytes(t), O.read_bytes_py(t)), t


def test_scan_matches_literal_loop_on_random_bytes():
    rng = np.random.default_rng(2)
    pool = np.array([0x33, 0xCC, 0x05, 0xC7, 0x41, 0x7F, 0x81, 0xBF, 0x00, 0xFF], np.uint8)
    for _ in range(200):
        b = pool[rng.integers(0, len(pool), int(rng.integers(0, 300)))]
        rows, v, d = O.parse_groups(b)
        rp, v2, d2 = O.parse_groups_py(b)
        assert (v, d) == (v2, d2)
        got = np.stack([rows[k].astype(np.int64) for k in ("flag", "ue", "bs", "rss", "clk")], 1).tolist() if v else []
        assert got == [list(r) for r in rp]


@pytest.mark.parametrize("case", ["log0_lin", "log0_log", "log0_noblur", "log0_sigma2", "full_164520", "full_165358"])
def test_heatmap_stages(mapping, lut, case):
    z = np.load(os.path.join(GOLDEN, f"heatmap_{case}.npz"))
    s, c = z["sum"], z["cnt"]
    rows = np.flatnonzero(c.sum(1) > 0)
    cols = np.flatnonzero(c.sum(0) > 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        mat = np.where(c > 0, s / np.maximum(c, 1), np.nan)[np.ix_(rows, cols)]
    assert np.array_equal(mat, z["matrix"], equal_nan=True), "mean = exact integer sum / count"
    ang = np.array([mapping[i] for i in range(64)])
    assert np.array_equal(ang[rows], z["aod"]) and np.array_equal(ang[cols], z["aoa"])
    use_log, sigma = bool(z["use_log"]), float(z["sigma"])
    r = O.render(mat, z["aod"], z["aoa"], lut, use_log=use_log, blur_sigma=sigma, W=64, H=64)
    bl = z["blurred"]
    assert np.array_equal(np.isnan(bl), np.isnan(r["blurred"]))
    fin = np.isfinite(bl)
    assert np.max(np.abs(bl[fin] - r["blurred"][fin]) / np.abs(bl[fin])) < 1e-12
    assert abs(r["vmin"] - float(z["vmin"])) <= 1e-12 * abs(float(z["vmin"]))
    assert abs(r["vmax"] - float(z["vmax"])) <= 1e-12 * abs(float(z["vmax"]))
    assert np.array_equal(O.edges(z["aoa"]), z["x_edges"]) and np.array_equal(O.edges(z["aod"]), z["y_edges"])
    # cell colours against the reference's saved PNG sampled at cell centres: everything except the
    # cells a dashed grid line runs through
    cp, mine = z["cells_png"], r["cells"]
    fin = mine[..., 3] == 255
    bad = ((cp[..., :3].astype(int) != mine[..., :3].astype(int)).any(-1) & fin).sum()
    assert bad <= 0.05 * fin.sum(), (bad, fin.sum())


def test_gaussian_kernels():
    z = np.load(os.path.join(GOLDEN, "gaussian_kernels.npz"))
    for key in z.files:
        sigma = float(key[1:].replace("_", "."))
        k = O.gaussian_kernel(sigma)
        assert k.shape == z[key].shape and np.allclose(k, z[key], rtol=1e-14, atol=0)


def test_pivot_rows_with_bad_or_unmapped_beams_are_dropped(mapping):
    ue = np.array([0, 1, 64, -1, 5, 5])
    bs = np.array([0, 70, 3, 3, 6, 6])
    rss = np.array([10, 20, 30, 40, 50, 70])
    mp = {k: v for k, v in mapping.items() if k != 1}
    mat, aod, aoa, s, c = O.pivot_mean(ue, bs, rss, mp)
    assert c.sum() == 3 and mat.shape == (2, 2)
    assert mat[1, 1] == 60.0 and mat[0, 0] == 10.0 and np.isnan(mat[0, 1])


def test_reject_positions_and_log_lines_match_reference(golden_dir):
    """Oracle discard positions + the drop-in module's line formatter reproduce the reference's per-reject log
    lines (serial_hex_to_excel_v3.py:55-57, :64, :69, :77, :85) on every golden stream."""
    import importlib.util
    import json
    from conftest import PKG
    # under a private name: the reference's module of the same name is imported by other tests
    spec = importlib.util.spec_from_file_location("dropin_serial_hex", os.path.join(PKG, "serial_hex_to_excel_v3.py"))
    drop = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drop)
    doc = json.load(open(os.path.join(golden_dir, "reject_lines.json"), encoding="utf-8"))
    assert len(doc) >= 25
    for name, e in doc.items():
        by = np.array(e["bytes"], np.uint8)
        rows, v, d, pos, reason = O.parse_groups(by, want_rejects=True)
        assert (v, d) == (e["valid"], e["discarded"]), name
        assert len(pos) == d
        assert drop.reject_lines(by.tolist(), pos) == e["lines"], name


def test_oracle_null_cells_match_the_reference():
    """Tables with empty cells: group ids, outputs and failures as the reference's drivers gave them
    (tests/golden/make_null_tables.py)."""
    doc = json.load(open(os.path.join(GOLDEN, "null_tables.json"), encoding="utf-8"))
    assert len(doc) >= 80
    for name, e in doc.items():
        for key, filt in (("filtered", True), ("inplace", False)):
            gid, got = O.correct_table_nulls_py(e["rows"], filt)
            assert gid == e["gid"], name
            want = {k: v for k, v in e[key].items() if k != "log"}
            assert got == want, (name, key)


def test_oracle_skipped_tokens_match_the_reference_log():
    """The tokens _read_bytes skips and their enumerate index, as logged by the reference (tests/golden/make_token_lines.py)."""
    doc = json.load(open(os.path.join(GOLDEN, "token_lines.json"), encoding="utf-8"))
    assert len(doc) >= 40
    for name, e in doc.items():
        raw = bytes(e["text"])
        lines = ["跳过非16进制单字节token[%d]='%s'" % it for it in O.skipped_tokens_py(raw)]
        assert lines == e["lines"], name
        assert len(O.read_bytes(raw)) == e["n_bytes"], name


def test_oracle_colour_limits_match_the_reference_figure():
    """Given, equal, reversed and invalid vmin / vmax on both scales: limits of the mesh's norm, cell colours and
    failures as the reference's generate_heatmap produced them (tests/golden/make_limit_cases.py)."""
    lut = np.load(os.path.join(GOLDEN, "viridis_lut.npy"))
    doc = json.load(open(os.path.join(GOLDEN, "limit_cases.json")))
    assert len(doc) == 72
    for k, e in doc.items():
        mat = np.array([[np.nan if v is None else v for v in r] for r in e["matrix"]])
        if "error" in e:
            with pytest.raises(ValueError, match=e["message"]):
                O.render(mat, e["aod"], e["aoa"], lut, use_log=e["use_log"], vmin=e["vmin_in"], vmax=e["vmax_in"], blur_sigma=0.0, W=8, H=8)
            continue
        r = O.render(mat, e["aod"], e["aoa"], lut, use_log=e["use_log"], vmin=e["vmin_in"], vmax=e["vmax_in"], blur_sigma=0.0, W=8, H=8)
        assert (r["vmin"], r["vmax"]) == (e["vmin"], e["vmax"]), k
        assert np.array_equal(r["cells"].astype(int), np.array(e["cells"])), k


def test_oracle_on_a_complete_capture():
    """The smallest complete capture of the reference (622 KB of text, 8 groups, 414 baselines): counters and kept rows
    as the reference's functions gave them, pivot sums as in its heatmap fixture."""
    import hashlib
    z = np.load(os.path.join(GOLDEN, "full_164520_text.npz"))
    rows, v, d = O.parse_groups(O.read_bytes(z["text"].tobytes()))
    assert (v, d) == (int(z["valid"]), int(z["discarded"]))
    kept, ng, nb = O.filter_corrected(rows["flag"], rows["ue"], rows["bs"], rows["rss"], rows["clk"])
    assert (ng, nb, len(kept["ue"])) == (int(z["groups"]), int(z["baselines"]), int(z["kept"]))
    arr = np.ascontiguousarray(np.stack([kept[k] for k in ("ue", "bs", "rss", "clk")], 1).astype(np.int64))
    assert hashlib.sha256(arr.tobytes()).hexdigest() == str(z["kept_sha256"])
    mapping = {int(k): v for k, v in json.load(open(os.path.join(GOLDEN, "beam_angle.json"))).items()}
    mat, aod, aoa, so, co = O.pivot_mean(kept["ue"], kept["bs"], kept["rss"], mapping)
    hz = np.load(os.path.join(GOLDEN, "heatmap_full_164520.npz"))
    assert np.array_equal(mat, hz["matrix"], equal_nan=True)


def load_all_captures():
    """(index entry, text bytes, reference's blurred matrix) per capture of tests/golden/captures_all.npz; the text is
    rebuilt from the stored data bytes and checked against the digest of the reference's file."""
    import hashlib
    import lzma
    z = np.load(os.path.join(GOLDEN, "captures_all.npz"))
    index = json.loads(z["index"].tobytes().decode("utf-8"))
    out = []
    for i, e in enumerate(index):
        data = lzma.decompress(z[f"data_{i}"].tobytes())
        text = "« ".encode("utf-8") + "".join(f"{b:02X} " for b in data).encode("ascii")
        assert len(text) == e["n_text"] and hashlib.sha256(text).hexdigest() == e["text_sha256"], e["name"]
        out.append((e, text, z[f"blurred_{i}"]))
    return out


def test_oracle_on_every_capture_of_the_reference():
    """All 19 capture logs: counters, decoded rows, kept rows and the mean matrix exactly as the reference's own
    functions gave them (digests), its blurred matrix to the last bits."""
    import hashlib
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()      # noqa: E731
    mapping = {int(k): v for k, v in json.load(open(os.path.join(GOLDEN, "beam_angle.json"))).items()}
    caps = load_all_captures()
    assert len(caps) == 19
    for e, text, blurred in caps:
        rows, v, d = O.parse_groups(O.read_bytes(text))
        assert (v, d) == (e["valid"], e["discarded"]), e["name"]
        assert sha(np.stack([rows[k].astype(np.int64) for k in ("flag", "ue", "bs", "rss", "clk")], 1)) == e["rows_sha256"]
        kept, ng, nb = O.filter_corrected(rows["flag"], rows["ue"], rows["bs"], rows["rss"], rows["clk"])
        assert (ng, nb, len(kept["ue"])) == (e["groups"], e["baselines"], e["kept"]), e["name"]
        assert sha(np.stack([kept[k] for k in ("ue", "bs", "rss", "clk")], 1).astype(np.int64)) == e["kept_sha256"]
        mat, aod, aoa, so, co = O.pivot_mean(kept["ue"], kept["bs"], kept["rss"], mapping)
        assert list(aod) == e["aod"] and list(aoa) == e["aoa"] and sha(mat.astype(np.float64)) == e["matrix_sha256"]
        mine = O.blur_nan_aware(mat, 1.0)
        fin = np.isfinite(blurred)
        assert np.array_equal(np.isfinite(mine), fin)
        assert np.max(np.abs(mine[fin] - blurred[fin]) / np.abs(blurred[fin])) <= 1e-14, e["name"]


def test_oracle_colours_against_the_heatmaps_the_reference_committed():
    """The three figures under debugDoc/heatmap_outputs (excel_heatmap_v3.py:425-471, CLI defaults), sampled at every
    mesh cell centre by tests/golden/make_committed_pngs.py next to the rows of the matching _filtered.xlsx: the
    oracle's pivot -> blur -> scale -> colour chain gives the same colour in every cell except along the row and the
    column the figure's dashed grid line at 0 degrees runs through."""
    from conftest import mesh_cells_agree
    z = np.load(os.path.join(GOLDEN, "committed_png_cells.npz"))
    mapping = {int(k): v for k, v in json.load(open(os.path.join(GOLDEN, "beam_angle.json"))).items()}
    lut = np.load(os.path.join(GOLDEN, "viridis_lut.npy"))
    for k in range(3):
        rows = z[f"f{k}_rows"].astype(np.int64)
        mat, aod, aoa, s, c = O.pivot_mean(rows[:, 0], rows[:, 1], rows[:, 2], mapping)
        assert aod == z[f"f{k}_aod"].tolist() and aoa == z[f"f{k}_aoa"].tolist()
        r = O.render(mat, aod, aoa, lut, use_log=False, blur_sigma=1.0, W=8, H=8)
        frac, bad_rows, bad_cols, rest_equal = mesh_cells_agree(r["cells"][..., :3], z[f"f{k}_cells"][..., :3])
        assert frac >= 0.97 and len(bad_rows) <= 1 and len(bad_cols) <= 1 and rest_equal, (k, frac, bad_rows, bad_cols)
        assert (z[f"f{k}_cells"][..., 3] == 255).all() and tuple(z[f"f{k}_size"]) == (1200, 1500)


FLOAT_TABLES = ("dbm", "mixed_scales", "few_cells", "infinities", "halves")


def test_oracle_float_pivot_against_the_reference_fixture():
    """RSS values that are not whole numbers: the oracle's restatement of pandas' compensated, row-ordered cell sums
    against matrices the reference's own compute_rssi_matrix returned (tests/golden/make_float_pivot.py), bit for bit."""
    z = np.load(os.path.join(GOLDEN, "float_pivot.npz"))
    mapping = {int(k): v for k, v in json.load(open(os.path.join(GOLDEN, "beam_angle.json"))).items()}
    for name in FLOAT_TABLES:
        mat, aod, aoa, s, c = O.pivot_mean_float(z[name + "_ue"], z[name + "_bs"], z[name + "_rss"], mapping)
        assert aod == z[name + "_aod"].tolist() and aoa == z[name + "_aoa"].tolist(), name
        want = z[name + "_matrix"]
        assert np.array_equal(np.isnan(mat), np.isnan(want)), name
        assert np.array_equal(mat.view(np.uint64)[~np.isnan(mat)], want.view(np.uint64)[~np.isnan(want)]), name
        # the lock-step form used for large tables: same bits, sums and counts included
        mat2, aod2, aoa2, s2, c2 = O.pivot_mean_float_lockstep(z[name + "_ue"], z[name + "_bs"], z[name + "_rss"], mapping)
        assert aod2 == aod and aoa2 == aoa and np.array_equal(c2, c), name
        assert np.array_equal(s2.view(np.uint64), s.view(np.uint64)) and np.array_equal(mat2.view(np.uint64), mat.view(np.uint64)), name
    rng = np.random.default_rng(11)
    mp = {i: float(i // 2) for i in range(0, 64) if i % 7}               # duplicate angles and unmapped beams
    for trial in range(4):
        n = int(rng.integers(0, 20000))
        ue, bs = rng.integers(-2, 66, n), rng.integers(-2, 66, n)
        rss = rng.normal(-70, 20, n) * rng.choice([1.0, 1e-5, 1e7, 1e300], n)
        rss[rng.random(n) < 0.01] = np.nan
        rss[rng.random(n) < 0.002] = np.inf
        a, b = O.pivot_mean_float(ue, bs, rss, mp), O.pivot_mean_float_lockstep(ue, bs, rss, mp)
        assert a[1] == b[1] and a[2] == b[2] and np.array_equal(a[4], b[4]), trial
        assert np.array_equal(a[3].view(np.uint64), b[3].view(np.uint64)), trial
    # whole numbers: the compensated sum is the exact integer sum
    rows = np.load(os.path.join(GOLDEN, "log0.npz"))["kept"]
    a = O.pivot_mean_float(rows[:, 0], rows[:, 1], rows[:, 2].astype(np.float64), mapping)
    b = O.pivot_mean(rows[:, 0], rows[:, 1], rows[:, 2], mapping)
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[4], b[4])


def test_oracle_dictionary_correlation_against_the_reference_fixture():
    """The path estimators' dictionaries and their correlation with a matrix (heatmap_gemini_v4.py:175-209, :271-304):
    the oracle's restatement against what the reference's own LoSEstimator / NLoSEstimator objects held
    (tests/golden/make_dict_corr.py).  Dictionaries and grids bit for bit (the same NumPy expressions); the correlation
    to 1e-13 of its largest element (BLAS may order the sums differently from run to run), and the written-out loops
    agree with the matrix products to the same bar."""
    z = np.load(os.path.join(GOLDEN, "dict_corr.npz"))
    for name in ("capture", "random"):
        ue, bs, m, res = z[name + "_ue_angles"], z[name + "_bs_angles"], z[name + "_matrix"], float(z[name + "_grid_res"])
        for kind, build in (("los", O.los_dictionary), ("nlos", O.nlos_dictionary)):
            aoa, aod, prx, ptx = build(ue, bs, grid_res=res, beam_width=1.4)
            assert np.array_equal(aoa, z[f"{name}_{kind}_aoa"]) and np.array_equal(aod, z[f"{name}_{kind}_aod"]), (name, kind)
            if name == "capture":
                assert np.array_equal(prx, z[f"{name}_{kind}_phi_rx"]) and np.array_equal(ptx, z[f"{name}_{kind}_phi_tx"]), kind
            want = z[f"{name}_{kind}_corr"]
            got = O.dict_correlation(prx, m, ptx)
            assert got.shape == want.shape and np.abs(got - want).max() <= 1e-13 * np.abs(want).max(), (name, kind)
    small = O.dict_correlation_loops(prx[:9, :11], m[:9, :7], ptx[:7, :5])
    assert np.abs(small - O.dict_correlation(prx[:9, :11], m[:9, :7], ptx[:7, :5])).max() <= 1e-13 * np.abs(small).max()
