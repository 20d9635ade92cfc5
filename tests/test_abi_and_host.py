"""CPU-side checks: the C-ABI library loads and exports what the header declares; host logic."""
import hashlib
import inspect
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, PKG, REFERENCE, ROOT, needs_reference


def test_library_exports_every_declared_symbol():
    from slamhip import _lib
    hdr = open(os.path.join(ROOT, "include", "slamhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(slam_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = _lib.load()
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == declared
    assert lib.slam_version().startswith(b"slamhip")
    assert lib.slam_synth_text_len(256, 4096) == 3 + 3 * 256 * (7 + 11 * 4096)


def test_batch_struct_matches_the_header():
    """ctypes mirror of struct slam_batch: same field order as the C declaration."""
    from slamhip import _lib
    hdr = open(os.path.join(ROOT, "include", "slamhip.h")).read()
    body = hdr[hdr.index("typedef struct slam_batch {"):hdr.index("} slam_batch;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part)[-1])
    assert names == [f[0] for f in _lib.Batch._fields_]


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from slamhip import _lib, ops
    with pytest.raises(_lib.SlamError):
        ops.parse_records(np.zeros(22, np.uint8))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert "slam_oracle" not in src and "liboracle" not in src, os.path.join(dirpath, f)


def test_kernel_resources():
    """The compiler's own report of the library as built: the hot kernels use no scratch (no register spills to
    memory) and keep the waves per SIMD their design counts on, by registers and by LDS."""
    from slamhip import build
    build.build(force=False, verbose=False)               # rebuilds (and rewrites the report) when a source is newer
    rep = build.read_report()
    assert len(rep) > 40
    committed = os.path.join(ROOT, "profiles", "r3_kernel_resources.txt")
    assert open(committed).read() == open(build.REPORT).read(), "profiles/r3_kernel_resources.txt is not this build's report"

    def waves(k):
        threads = 1024 if k["kernel"].startswith("k_pivot<") else 256
        if k["kernel"].startswith("k_frame_fused"):
            threads = 512
        by_lds = (160 * 1024 // k["lds"]) * (threads // 64) // 4 if k["lds"] else 8
        return min(k["waves"], by_lds, 8)

    need = {"k_frame_onepass<0>": 4, "k_frame_onepass<1>": 4, "k_frame_fused<0>": 4, "k_frame_fused<1>": 4,
            "k_rows_finish<unsigned char, unsigned int, 0, 3>": 6, "k_rows_finish<unsigned char, unsigned int, 1, 3>": 6,
            "k_rows_finish<unsigned char, unsigned int, 2, 3>": 4, "k_pivot<unsigned char, unsigned int>": 4,
            "k_raster": 8, "k_bl_sort_small": 6}
    by_name = {k["kernel"]: k for k in rep}
    for name, w in need.items():
        assert name in by_name, name
        k = by_name[name]
        assert k["scratch"] == 0 and k["vgpr_spill"] == 0, (name, k)
        assert waves(k) >= w, (name, k, waves(k))
    # no kernel of the library spills vector registers at all
    assert [k["kernel"] for k in rep if k["scratch"] or k["vgpr_spill"]] == []


def test_xlsx_round_trip(tmp_path):
    from slamhip import xlsx
    p = str(tmp_path / "t.xlsx")
    xlsx.write_table(p, ["FLAG", "UE_Beam[5:0]十进制", "x<y&z"], [[0, 1, 1], [5, 63, 0], [1.5, None, 2e10]], sheet="Parsed")
    assert xlsx.sheet_names(p) == ["Parsed"]
    t = xlsx.read_table(p)
    assert t["FLAG"] == [0, 1, 1] and t["UE_Beam[5:0]十进制"] == [5, 63, 0] and t["x<y&z"] == [1.5, None, 2e10]
    import pandas as pd
    try:
        import openpyxl  # noqa: F401
    except ImportError:
        return
    assert pd.read_excel(p)["FLAG"].tolist() == [0, 1, 1]


@needs_reference
def test_xlsx_reads_the_reference_workbooks():
    from slamhip import xlsx
    t = xlsx.read_table(os.path.join(REFERENCE, "beam_angle.xlsx"))          # shared strings
    assert t["BeamID"] == list(range(64)) and t["Angle"][0] == -43.6 and t["Angle"][-1] == 45
    gold = {int(k): v for k, v in json.load(open(os.path.join(GOLDEN, "beam_angle.json"))).items()}
    assert {i: float(a) for i, a in zip(t["BeamID"], t["Angle"])} == gold
    t = xlsx.read_table(os.path.join(REFERENCE, "debugDoc", "Serial Debug 2026-02-06 092721.xlsx"))   # inline strings
    assert len(t["FLAG"]) == 19527


def test_synth_generator_is_pinned():
    from slamhip import synth
    t = synth.make_frame_text(4, 128, 3)
    assert len(t) == synth.text_len(4, 128) and bytes(t[:3]) == "« ".encode()
    assert hashlib.sha256(t.tobytes()).hexdigest() == json.load(open(os.path.join(GOLDEN, "synth_digest.json")))["4x128_f3"]
    assert not np.array_equal(t, synth.make_frame_text(4, 128, 4))
    # the generator's frames exercise resync, baselines, ties and drops
    from oracle import slam_oracle as O
    by = synth.make_frame_bytes(8, 1024, 0)
    rows, v, d = O.parse_groups(by)
    assert 0.99 * 8192 < v < 8192 and d >= 8 * 5
    kept, ng, nb = O.filter_corrected(rows["flag"], rows["ue"], rows["bs"], rows["rss"], rows["clk"])
    assert ng == 8 and nb > 8 * 10 and 0.5 < len(kept["ue"]) / v < 0.7


@needs_reference
def test_drop_in_modules_mirror_the_reference_signatures():
    """Same public names and the same parameter lists as the reference's three scripts."""
    import importlib.util
    import matplotlib
    matplotlib.use("Agg")

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    for mod in ("serial_hex_to_excel_v3", "bs_beam_correction", "excel_heatmap_v3"):
        ref = load(os.path.join(REFERENCE, mod + ".py"), "ref_" + mod)
        mine = load(os.path.join(PKG, mod + ".py"), "mine_" + mod)
        for name, obj in vars(ref).items():
            if inspect.isfunction(obj) and obj.__module__ == ref.__name__:
                assert hasattr(mine, name), f"{mod}.{name} missing"
                ps_ref = [(p.name, p.default) for p in inspect.signature(obj).parameters.values()]
                ps_mine = [(p.name, p.default) for p in inspect.signature(getattr(mine, name)).parameters.values()]
                assert ps_ref == ps_mine, f"{mod}.{name}: {ps_ref} != {ps_mine}"
            elif name.isupper() and isinstance(obj, (int, str)):
                assert getattr(mine, name) == obj, f"{mod}.{name}"


def test_stand_ins_for_empty_cells_reproduce_the_group_rule():
    """Host logic of the correction drop-in: the integers that stand in for null UE / RSS / FLAG cells make plain
    integer rules (previous UE > UE; RSS equal; FLAG == 0 / == 1) agree with the reference's NaN rules on every
    table of the null-cell fixture."""
    import importlib.util
    import pandas as pd
    spec = importlib.util.spec_from_file_location("mine_bs_beam_correction_enc", os.path.join(PKG, "bs_beam_correction.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    doc = json.load(open(os.path.join(GOLDEN, "null_tables.json"), encoding="utf-8"))
    for name, e in doc.items():
        cols = [[np.nan if v is None else v for v in c] for c in zip(*e["rows"])]
        df = pd.DataFrame(dict(zip(B._IN_COLS, cols)), dtype=np.float64)
        (f, u, b, r, c), nulls = B._encode(df)
        assert nulls is not None or not any(v is None for row in e["rows"] for v in row), name
        u = u.astype(np.int64)
        gid = np.cumsum(np.concatenate(([True], u[:-1] > u[1:]))) - 1
        assert gid.tolist() == e["gid"], name
        if nulls is None:
            continue
        nf, nu, nb, nr, nc = nulls
        same = r[1:] == r[:-1]
        assert not (same & (nr[1:] | nr[:-1])).any(), name          # a null RSS equals nothing
        assert not np.isin(f[nf], (0, 1)).any(), name               # a null FLAG is neither 0 nor 1
        assert all(np.isfinite(x).all() and -2 ** 31 <= x.min() and x.max() < 2 ** 31 for x in (f, u, b, r, c)), name


def test_png_writer_round_trips():
    """The stdlib PNG writer: Pillow reads back the same RGBA bytes."""
    import io
    from PIL import Image
    from slamhip import png
    rng = np.random.default_rng(5)
    for shape in ((1, 1, 4), (7, 13, 4), (256, 256, 4)):
        t = rng.integers(0, 256, shape, dtype=np.uint8)
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(png.encode_rgba(t)))), t)
    with pytest.raises(ValueError):
        png.encode_rgba(np.zeros((4, 4, 3), np.uint8))


def test_file_balancing_and_sharding():
    from slamhip import dist as D
    assert D.shard_frames(10, 1, 4) == [1, 5, 9]
    sizes = [622083, 3346947, 3348483, 2391555, 2865411, 2639619, 3112707, 2824707, 945411, 1158147]
    bins = D.balance_files(sizes, 4)
    assert sorted(i for b in bins for i in b) == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in b) for b in bins]
    assert max(loads) - min(loads) <= max(sizes)


def test_gather_of_tiles_two_ranks_gloo(tmp_path):
    """world_size 2 on CPU (gloo): tiles gathered from sharded frames equal the single-rank result."""
    script = tmp_path / "w.py"
    script.write_text(f'''
import os, sys
sys.path.insert(0, {PKG!r})
import torch, torch.distributed as dist
from slamhip import dist as D
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_frames, H, W = 7, 4, 6
def tile_of(f):
    return (torch.arange(H * W * 4, dtype=torch.int64).reshape(H, W, 4) * (f + 3) % 251).to(torch.uint8)
mine = D.shard_frames(n_frames, rank, world)
per = (n_frames + world - 1) // world
tiles = torch.zeros((per, H, W, 4), dtype=torch.uint8)
for i, f in enumerate(mine):
    tiles[i] = tile_of(f)
got = D.gather_tiles(tiles, dst=0)
if rank == 0:
    full = D.interleave_gathered(got, n_frames)
    ref = torch.stack([tile_of(f) for f in range(n_frames)])
    assert torch.equal(full, ref)
    open({str(tmp_path / "ok")!r}, "w").write("ok")
else:
    assert got is None
dist.destroy_process_group()
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    import socket
    with socket.socket() as sk:                              # a port that is free right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (tmp_path / "ok").exists()


def test_sharding_and_gather_order_eight_ranks_gloo(tmp_path):
    """world_size 8 on CPU (gloo), the size of the node the benchmark is specified for: frames dealt out by
    shard_frames and files dealt out by balance_files come back from gather_tiles in the right order, every frame and
    every file exactly once, with frame counts and file counts that do not divide by the number of ranks."""
    script = tmp_path / "w8.py"
    script.write_text(f"""
import os, sys
sys.path.insert(0, {PKG!r})
import torch, torch.distributed as dist
from slamhip import dist as D
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 8
H, W = 3, 5
def tile_of(f):
    return (torch.arange(H * W * 4, dtype=torch.int64).reshape(H, W, 4) * (7 * f + 3) % 253).to(torch.uint8)
# --- frames: f -> rank f mod 8
for n_frames in (29, 8, 5):
    mine = D.shard_frames(n_frames, rank, world)
    assert all(f % world == rank for f in mine)
    per = (n_frames + world - 1) // world
    tiles = torch.zeros((per, H, W, 4), dtype=torch.uint8)
    for i, f in enumerate(mine):
        tiles[i] = tile_of(f)
    got = D.gather_tiles(tiles, dst=0)
    if rank == 0:
        full = D.interleave_gathered(got, n_frames)
        assert torch.equal(full, torch.stack([tile_of(f) for f in range(n_frames)])), n_frames
    else:
        assert got is None
# --- files: the 19 capture sizes of the reference's debugDoc (bytes), greedy by size
sizes = [3321603, 3308091, 1841667, 1841475, 1828611, 1822659, 1816707, 1816515, 1810755, 1810563, 1810563, 1810371,
         1804611, 1804419, 1798467, 1798275, 1792515, 1792323, 622083]
bins = D.balance_files(sizes, world)
assert sorted(i for b in bins for i in b) == list(range(len(sizes)))
loads = [sum(sizes[i] for i in b) for b in bins]
assert max(loads) - min(loads) <= max(sizes)            # greedy by size: no rank more than one file ahead
most = max(len(b) for b in bins)
tiles = torch.zeros((most, H, W, 4), dtype=torch.uint8)
ids = torch.full((most,), -1, dtype=torch.int64)
for i, fi in enumerate(bins[rank]):
    tiles[i] = tile_of(100 + fi)
    ids[i] = fi
got = D.gather_tiles(tiles, dst=0)
got_ids = [torch.empty_like(ids) for _ in range(world)] if rank == 0 else None
dist.gather(ids, got_ids, dst=0)
if rank == 0:
    seen = {{}}
    for r in range(world):
        assert [int(x) for x in got_ids[r] if x >= 0] == bins[r]
        for i, fi in enumerate(bins[r]):
            assert fi not in seen
            seen[fi] = got[r][i]
    assert sorted(seen) == list(range(len(sizes)))
    assert all(torch.equal(seen[fi], tile_of(100 + fi)) for fi in seen)
    open({str(tmp_path / "ok8")!r}, "w").write("ok")
dist.destroy_process_group()
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    import socket
    with socket.socket() as sk:                              # a port that is free right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert (tmp_path / "ok8").exists()


def test_bench_refuses_a_world_that_is_not_the_one_asked_for():
    """bench.py --gpus N inside a launcher that started another number of ranks prints no line and fails, before it
    touches a GPU: a line over fewer GPUs than asked for would be read as the N-GPU result."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-frames", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 3 and r.stdout.strip() == "" and "--gpus 2" in r.stderr


def test_heatmap_reader_cleaning_rules(tmp_path):
    """read_main_data's cleaning (excel_heatmap_v3.py:139-182), host-only: rows with a bad RSS cell go; beams outside
    0..63 go; FLAG filters only when asked; a beam column with an empty or non-numeric cell turns float and then fails
    validate_beam_id (integers only) on every row, so the table comes out empty and the function raises."""
    import importlib.util
    import matplotlib
    matplotlib.use("Agg")
    from slamhip import xlsx
    spec = importlib.util.spec_from_file_location("mine_excel_heatmap_v3_clean", os.path.join(PKG, "excel_heatmap_v3.py"))
    E = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(E)
    hdr = ["UE_Beam", "BS_Beam", "RSS值", "FLAG"]
    p = str(tmp_path / "a.xlsx")
    xlsx.write_table(p, hdr, [[1, 2, 64, 3, -1, 5], [7, 8, 9, 10, 11, 63], [100, "x", 300, None, 500, 600.5], [1, 0, 1, 1, 1, 1]])
    df = E.read_main_data(p)
    assert df[["ue_beam", "bs_beam"]].to_numpy().tolist() == [[1, 7], [5, 63]] and df["rss"].tolist() == [100.0, 600.5]
    assert df["ue_beam"].dtype.kind == "i"
    df = E.read_main_data(p, flag_col="FLAG")
    assert df[["ue_beam", "bs_beam", "flag"]].to_numpy().tolist() == [[1, 7, 1], [5, 63, 1]]
    q = str(tmp_path / "b.xlsx")
    xlsx.write_table(q, hdr, [[1, None, 3], [7, 8, 9], [100, 200, 300], [1, 1, 1]])
    with pytest.raises(ValueError, match="清洗后主数据为空"):
        E.read_main_data(q)
    r = str(tmp_path / "c.xlsx")
    xlsx.write_table(r, hdr, [[1, 2, 3], [7, "beam", 9], [100, 200, 300], [1, 1, 1]])
    with pytest.raises(ValueError, match="清洗后主数据为空"):
        E.read_main_data(r)
    m = str(tmp_path / "m.xlsx")
    xlsx.write_table(m, ["BeamID", "Angle"], [[0, 1, 64, -1, None, 2], [-10.0, None, 5.0, 6.0, 7.0, "30"]])
    mp = E.read_mapping(m)
    assert set(mp) == {0, 1, 2} and mp[0] == -10.0 and np.isnan(mp[1]) and mp[2] == 30.0


def test_parser_refuses_values_that_are_not_bytes():
    """_parse_groups takes what _read_bytes yields (0..255); other integers are refused before any device call."""
    import importlib.util
    import logging
    spec = importlib.util.spec_from_file_location("mine_serial_hex_to_excel_v3_rng", os.path.join(PKG, "serial_hex_to_excel_v3.py"))
    S = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(S)
    for bad in ([0x33, 307], [-205, 1], [0.5, 1]):
        with pytest.raises(ValueError, match="0..255"):
            S._parse_groups(bad, logging.getLogger("t"))


def test_upscaled_axis_table_matches_the_oracles_definition():
    """The up-scaled beam table (host logic, no GPU): the product's table and beam positions equal the oracle's
    restatement for several factors and for sparse mappings; the table's own angles sit at every factor-th entry."""
    from oracle import slam_oracle as O
    from slamhip import _lib, ops
    from slamhip.pipeline import default_mapping
    maps = [default_mapping(), {3: 1.5}, {i: float(i // 2) for i in range(64) if i % 7}, {5: -1.0, 9: 4.0, 70: 9.0}]
    for mp in maps:
        b2a1, ang1 = ops.axis_from_mapping(mp)
        for factor in (1, 2, 3, 4):
            if len(ang1) * factor > _lib.MAX_AXIS:
                continue
            b2a, ang = ops.upscaled_axis(mp, factor)
            ob2a, oang = O.upscaled_axis(mp, factor)
            assert b2a.dtype == np.int16 and np.array_equal(b2a, ob2a) and np.array_equal(ang, oang)
            assert np.array_equal(ang[::factor], ang1) and len(ang) == factor * len(ang1)
            assert np.all(np.diff(ang) > 0) and np.array_equal(b2a[b2a1 >= 0], b2a1[b2a1 >= 0] * factor)
    with pytest.raises(ValueError):
        ops.upscaled_axis(default_mapping(), 5)              # 320 entries: beyond the library's 256
    assert _lib.cells_per_frame(64) == 4096 and _lib.cells_per_frame(65) == 4225 and _lib.cells_per_frame(256) == 65536
    assert _lib.load().slam_cells_per_frame(256) == 65536 and _lib.load().slam_cells_per_frame(8) == 4096


def test_bench_kernel_accounting_groups_the_builds_and_never_exceeds_the_peak():
    """bench.py's per-kernel table (host logic): the builds of the correction kernel are priced together against one
    set of algorithmic bytes, a launch too short for its nominal bytes gets no rate, the roofline entry names the
    traffic file of the shape that ran or says that there is none."""
    sys.path.insert(0, ROOT)
    import bench
    nbytes, valid, kept, tile_bytes, F, K = 46.1e6, 4.19e6, 2.5e6, 4194304, 200, 5           # a 256 x 16384 frame
    kern = {"frame_onepass": (K, 40.0), "rows_finish": (K, 0.3), "rows_finish_mid": (K, 16.0), "rows_finish_rest": (K, 0.3),
            "pivot": (K, 3.5), "raster": (K, 0.0005), "bl_sort_small": (K, 1.5)}
    table, roof, ach = bench.kernel_accounting(kern, nbytes, valid, kept, tile_bytes, lambda launches: F * K / launches, 256, 16384, 1)
    assert "rows_finish_mid" not in table and set(table["rows_finish"]["parts_avg_ms"]) == {"rows_finish", "rows_finish_mid", "rows_finish_rest"}
    assert abs(table["rows_finish"]["avg_ms"] - (0.3 + 16.0 + 0.3) / K) < 1e-4
    want = (10.0 * valid + 10.0 * kept) * F / ((16.6 / K) * 1e-3) / 1e9
    assert abs(table["rows_finish"]["alg_GBps"] - want) < 1.0
    assert table["raster"]["alg_GBps"] is None and "alg_note" in table["raster"]          # would be far above the peak
    assert all(v.get("alg_GBps") is None or v["alg_GBps"] <= bench.HBM_PEAK_GBS for v in table.values())
    assert roof["kernel"] == "frame_onepass" and roof["peak"] == 8000.0 and roof["frac"] == round(roof["achieved"] / 8000.0, 4)
    # the traffic comes from the committed counter passes of the very shape, per launch like the algorithmic bytes
    assert "256x16384" in roof["traffic_source"] and roof["traffic_source"].startswith("profiles/")
    assert roof["alg_bytes_per_launch"] <= roof["traffic"] < 1.2 * roof["alg_bytes_per_launch"]
    assert roof["pipeline_traffic_per_frame"] > 3.0 * nbytes + 11.0 * valid + 10.0 * kept
    _t, roof_none, _a = bench.kernel_accounting(kern, nbytes, valid, kept, tile_bytes, lambda launches: F * K / launches, 128, 2048, 1)
    assert roof_none["traffic"] is None and "128x2048" in roof_none["traffic_source"] and roof_none["pipeline_traffic_per_frame"] is None
    _t, roof2, _a = bench.kernel_accounting(kern, nbytes, valid, kept, tile_bytes, lambda launches: 800 * K / launches, 256, 4096, 1)
    assert roof2["traffic"] and roof2["traffic_source"].startswith("profiles/")
