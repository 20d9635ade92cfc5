"""Regenerates tests/golden/*.npz|json from the reference (build container only).

Imports the reference's three hot-path modules from /root/reference, runs their
own functions on small inputs, and stores inputs + outputs as data fixtures.
Nothing of the reference's source is stored.  Run:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import io
import json
import logging
import os
import sys
from pathlib import Path

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "slam-process_amd"))

import matplotlib  # noqa: E402

matplotlib.use("Agg")
import pandas as pd  # noqa: E402
import serial_hex_to_excel_v3 as S  # noqa: E402
import bs_beam_correction as B  # noqa: E402
import excel_heatmap_v3 as E  # noqa: E402
from slamhip import xlsx  # noqa: E402

logging.disable(logging.CRITICAL)
LG = logging.getLogger("golden")
D = os.path.join(REF, "debugDoc")


def ref_parse_text(raw: bytes):
    tmp = Path("/tmp/_golden_in.txt")
    tmp.write_bytes(raw)
    by = S._read_bytes(tmp, LG)
    rows, v, d = S._parse_groups(by, LG)
    return np.array(by, np.uint8), np.array(rows, np.int64).reshape(-1, 5), v, d


def ref_filter(rows: np.ndarray):
    df = pd.DataFrame(rows, columns=[B.FLAG_COL, B.UE_BEAM_COL, B.BS_BEAM_COL, B.RSS_COL, B.CLK_COL])
    df = B._coerce_types(df)
    gid = B._detect_groups(df)
    df["_gid"] = gid
    pieces, bl_all = [], []
    for g, sub in df.groupby("_gid", sort=True):
        bl = B._identify_baselines(sub)
        bl_all += [(c, b, int(g)) for c, b in bl]
        f = B._filter_group_corrected(sub, bl)
        if not f.empty:
            pieces.append(f)
    kept = pd.concat(pieces, ignore_index=True).to_numpy(np.int64) if pieces else np.zeros((0, 4), np.int64)
    return gid.to_numpy(np.int64), np.array(bl_all, np.int64).reshape(-1, 3), kept


def ref_correct_inplace(rows: np.ndarray):
    df = pd.DataFrame(rows, columns=[B.FLAG_COL, B.UE_BEAM_COL, B.BS_BEAM_COL, B.RSS_COL, B.CLK_COL])
    gid = B._detect_groups(df)
    df["_gid"] = gid
    out = []
    for g, sub in df.groupby("_gid", sort=True):
        out.append(B._correct_group(sub, B._identify_baselines(sub)))
    res = pd.concat(out).sort_index()
    return res[B.OUT_COL].to_numpy(np.float64).astype(np.int64)


def mapping_dict():
    t = xlsx.read_table(os.path.join(REF, "beam_angle.xlsx"))
    return {int(i): float(a) for i, a in zip(t["BeamID"], t["Angle"]) if 0 <= int(i) <= 63}


def ref_heatmap_stages(kept: np.ndarray, mapping, use_log: bool, sigma: float, tag: str):
    """Runs map_angles -> compute_rssi_matrix -> generate_heatmap and captures the mesh's data path."""
    df = pd.DataFrame({"ue_beam": kept[:, 0], "bs_beam": kept[:, 1], "rss": kept[:, 2].astype(np.float64)})
    dfm = E.map_angles(df, mapping)
    mat, aod, aoa = E.compute_rssi_matrix(dfm)
    blurred = E.gaussian_blur_nan_aware(mat.copy(), sigma) if sigma > 0 else mat.copy()
    figs = []
    orig_close = E.plt.close
    E.plt.close = lambda fig=None: figs.append(fig)
    out_png = f"/tmp/_golden_{tag}.png"
    try:
        E.generate_heatmap(mat, aod, aoa, out_png, "t", colormap="viridis", use_log=use_log,
                           vmin=None, vmax=None, blur_sigma=sigma, logger=None)
    finally:
        E.plt.close = orig_close
    fig = figs[0]
    ax = fig.axes[0]
    mesh = ax.collections[0]
    vmin, vmax = float(mesh.norm.vmin), float(mesh.norm.vmax)
    # sample the saved PNG at every cell centre (figure saved at dpi 150, canvas built at dpi 120)
    from PIL import Image
    img = np.array(Image.open(out_png).convert("RGBA"))
    Himg = img.shape[0]
    xe = np.array(mesh.get_coordinates()[0, :, 0])
    ye = np.array(mesh.get_coordinates()[:, 0, 1])
    xc = (xe[:-1] + xe[1:]) / 2
    yc = (ye[:-1] + ye[1:]) / 2
    XX, YY = np.meshgrid(xc, yc)
    pix = ax.transData.transform(np.stack([XX.ravel(), YY.ravel()], 1)) * (150.0 / fig.dpi)
    px = np.clip(np.round(pix[:, 0] - 0.5).astype(int), 0, img.shape[1] - 1)
    py = np.clip(np.round(Himg - pix[:, 1] - 0.5).astype(int), 0, Himg - 1)
    cells_png = img[py, px].reshape(len(yc), len(xc), 4)
    orig_close(fig)
    return dict(matrix=mat, aod=np.array(aod), aoa=np.array(aoa), blurred=blurred,
                vmin=vmin, vmax=vmax, x_edges=xe, y_edges=ye, cells_png=cells_png)


def main():
    out = {}
    mapping = mapping_dict()
    with open(os.path.join(HERE, "beam_angle.json"), "w") as fh:
        json.dump({str(k): v for k, v in sorted(mapping.items())}, fh)
    # the Agg renderer turns float colours into bytes by rounding: the saved PNG shows round(c*255)
    lut = np.rint(np.array(matplotlib.colormaps["viridis"].colors) * 255).astype(np.uint8)
    np.save(os.path.join(HERE, "viridis_lut.npy"), lut)

    # 1. truncated real captures ------------------------------------------------------------
    logs = ["256-0x16-Serial Debug 2026-01-26 164520.txt",
            "Serial Debug 2026-01-27 114647.txt",
            "Serial Debug 2026-02-06 092721.txt"]
    for i, name in enumerate(logs):
        raw = open(os.path.join(D, name), "rb").read()[: 3 + 3 * 30000 + (i % 2)]   # ragged cut on odd i
        by, rows, v, d = ref_parse_text(raw)
        gid, bl, kept = ref_filter(rows)
        np.savez_compressed(os.path.join(HERE, f"log{i}.npz"), text=np.frombuffer(raw, np.uint8), bytes=by,
                            rows=rows, valid=v, discarded=d, gid=gid, baselines=bl, kept=kept,
                            corrected_inplace=ref_correct_inplace(rows))
        out[f"log{i}"] = dict(source=name, text_bytes=len(raw), n_bytes=len(by), valid=v, discarded=d,
                              groups=int(gid.max()) + 1, baselines=len(bl), kept=len(kept))

    # 2. handcrafted tokenizer / framing edge streams ------------------------------------------
    rec = bytes([0x33, 0x05, 0xC7, 0x41, 0x42, 0x43, 0x44, 0x45, 0x81, 0x82, 0x83])
    rec2 = bytes([0xCC, 0x33, 0xCC, 0x7F, 0x40, 0x55, 0x6A, 0x41, 0xBF, 0x80, 0x95])   # ue=0x33, bs=0xCC look like flags
    hexs = lambda b: " ".join(f"{x:02X}" for x in b)
    edge_texts = {
        "empty": b"",
        "only_prefix": "« ".encode(),
        "one_record": ("« " + hexs(rec) + " ").encode(),
        "no_trailing_space": ("« " + hexs(rec + rec2)).encode(),
        "lower_and_0x": (" ".join(f"0x{x:02x}" for x in rec) + " " + hexs(rec2).lower()).encode(),
        "upper_0X_rejected": b"0X33 33 05 C7 41 42 43 44 45 81 82 83",
        "bad_tokens": b"333 3 GG 0x 0x1 0x123 33 g3 3g 33\n05\tC7\r\n41 42 43 44 45 81 82 83",
        "unicode_ws": ("33 05　05 C7\u008541 42 43 44 45 81 82 83").encode(),
        "nul_no_split": b"33\x0005 33 05 C7 41 42 43 44 45 81 82 83",
        "fs_gs_rs_us_split": b"33\x1c05\x1dC7\x1e41\x1f42 43 44 45 81 82 83",
        "invalid_utf8_joins": b"3\xff3 0\xc05 C\x807 41 42 43 44 45 81 82 83 \xe2\x28\xa1 \xf0\x9f 4\xe2\x824",
        "truncated_utf8_tail": b"33 05 C7 41 42 43 44 45 81 82 8\xe2\x82",
        "bom": b"\xef\xbb\xbf33 05 C7 41 42 43 44 45 81 82 83 33",
        "tail_short": (hexs(rec) + " " + hexs(rec[:7])).encode(),
        "tail_flag_last": (hexs(rec) + " 00 33").encode(),
        "resync_garbage": (hexs(b"\x33\x33\x33\x33\x33\x2f\xe2" + rec + b"\xcc\x00" + rec2 + rec[:5] + rec + rec2)).encode(),
        "all_flags": hexs(b"\x33" * 40 + b"\xcc" * 40).encode(),
        "no_flags": hexs(bytes(range(0x40, 0x90))).encode(),
        "tag_flip_each": hexs(b"".join(bytes(rec[j] ^ (0x40 if j == k else 0) for j in range(11)) for k in range(11)) + rec).encode(),
        "interleaved_alignment": hexs((rec2[:1] + rec2) * 12).encode(),
    }
    edge = {}
    for k, raw in edge_texts.items():
        by, rows, v, d = ref_parse_text(raw)
        edge[k] = dict(text=list(raw), bytes=by.tolist(), rows=rows.tolist(), valid=v, discarded=d)
    with open(os.path.join(HERE, "edge_streams.json"), "w") as fh:
        json.dump(edge, fh)
    out["edge_streams"] = {k: (e["valid"], e["discarded"]) for k, e in edge.items()}

    # 3. the reference's embedded unit-test tables, with the CODE's outputs ------------------------
    C = B.CYCLE
    T = B.TOL
    tables = {
        "make_group": [[0, 0, 10, 42, 1_000_000], [1, 1, 12, 42, 1_000_100], [0, 2, 99, 42, 1_000_000 + C + 50],
                       [0, 3, 99, 42, 1_000_000 + 2 * C - 480], [0, 4, 99, 42, 1_000_000 + 3 * C + 600],
                       [0, 5, 99, 42, 1_000_000 - C + 100]],
        "boundary_tolerance": [[0, 0, 3, 7, 5_000_000], [1, 1, 8, 7, 5_000_010], [0, 2, 0, 7, 5_000_000 + C + T],
                               [0, 3, 0, 7, 5_000_000 + C + T + 1]],
        "negative_diff": [[0, 0, 60, 13, 7_000_000], [1, 1, 5, 13, 7_000_001], [0, 2, 0, 13, 7_000_000 - C + 10]],
        "filter_only": [[0, 0, 10, 21, 2_000_000], [1, 1, 12, 21, 2_000_050], [0, 2, 99, 21, 2_000_000 + C + 20],
                        [0, 3, 99, 21, 2_000_000 + C + T + 10]],
        "two_groups_ties": [[0, 5, 1, 9, 100], [1, 6, 2, 9, 150], [0, 7, 3, 9, 100 + C - 20], [1, 8, 40, 9, 100 + C],
                            [0, 9, 3, 9, 100 + C - 10], [0, 9, 3, 9, 100 + 5 * C + 10],
                            [0, 2, 1, 9, 100], [0, 3, 1, 9, 100 + C], [1, 4, 63, 5, 200], [1, 5, 63, 5, 300],
                            [0, 6, 7, 5, 400], [1, 7, 9, 5, 500], [0, 8, 7, 5, 400 + 70 * C + 500], [0, 8, 7, 5, 400 - 3 * C - 500]],
    }
    ut = {}
    for k, t in tables.items():
        rows = np.array(t, np.int64)
        gid, bl, kept = ref_filter(rows)
        ut[k] = dict(rows=rows.tolist(), gid=gid.tolist(), baselines=bl.tolist(), kept=kept.tolist(),
                     corrected_inplace=ref_correct_inplace(rows).tolist())
    with open(os.path.join(HERE, "unit_tables.json"), "w") as fh:
        json.dump(ut, fh)

    # 4. heatmap stages ------------------------------------------------------------------------
    hm = {}
    z = np.load(os.path.join(HERE, "log0.npz"))
    cases = {"log0_lin": (z["kept"], False, 1.0), "log0_log": (z["kept"], True, 1.0), "log0_noblur": (z["kept"], False, 0.0),
             "log0_sigma2": (z["kept"], False, 2.0)}
    # full committed filtered outputs (the three with committed PNGs): store only the 64x64-scale arrays
    for nm in ["Serial Debug 2026-01-26 164520", "Serial Debug 2026-01-26 165358"]:
        ft = xlsx.read_table(os.path.join(D, nm + "_filtered.xlsx"))
        rk = [k for k in ft if k.startswith("RSS")][0]
        kept = np.array([ft["UE_Beam"], ft["BS_Beam"], ft[rk], ft["CLK值"]], np.int64).T
        cases["full_" + nm[-6:]] = (kept, False, 1.0)
    for k, (kept, use_log, sigma) in cases.items():
        r = ref_heatmap_stages(kept, mapping, use_log, sigma, k)
        # pivot inputs as exact integer sums/counts over beam ids, so the fixture stays small
        s = np.zeros((64, 64), np.int64)
        c = np.zeros((64, 64), np.int64)
        np.add.at(s, (kept[:, 1], kept[:, 0]), kept[:, 2])
        np.add.at(c, (kept[:, 1], kept[:, 0]), 1)
        hm[k] = dict(use_log=use_log, sigma=sigma)
        np.savez_compressed(os.path.join(HERE, f"heatmap_{k}.npz"), sum=s, cnt=c, use_log=use_log, sigma=sigma, **r)
    out["heatmap"] = hm
    # gaussian kernels
    np.savez_compressed(os.path.join(HERE, "gaussian_kernels.npz"),
                        **{f"s{str(s).replace('.', '_')}": E.gaussian_kernel(s) for s in (0.0, 0.3, 0.5, 1.0, 1.5, 2.0)})
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as fh:
        json.dump(out, fh, indent=1, ensure_ascii=False)
    print(json.dumps(out, indent=1, ensure_ascii=False))


if __name__ == "__main__":
    main()
