"""Samples the three heatmaps the reference's authors committed (debugDoc/heatmap_outputs/*_filtered_heatmap.png,
written by excel_heatmap_v3.py:425-471 with the CLI defaults: linear scale, viridis, blur sigma 1.0) next to the
rows they were drawn from (the matching debugDoc/*_filtered.xlsx).  Build container only.

Per figure the fixture holds DATA only: the UE / BS / RSS columns of the workbook, the figure's size, the axes box
found from its spines, and the RGBA value of the pixel at the centre of every mesh cell.  Cell centres follow from
the workbook through the reference's own map_angles / compute_rssi_matrix (imported here) and its _edges rule.
Run:  python tests/golden/make_committed_pngs.py   ->  tests/golden/committed_png_cells.npz
"""
from __future__ import annotations

import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "slam-process_amd"))      # for slamhip.xlsx (a standard-library workbook reader)
sys.path.insert(0, REF)                                           # the reference's excel_heatmap_v3 comes first

import matplotlib  # noqa: E402

matplotlib.use("Agg")
import pandas as pd  # noqa: E402
from PIL import Image  # noqa: E402
import excel_heatmap_v3 as E  # noqa: E402  (the reference's, first on the path)
from slamhip import xlsx  # noqa: E402

D = os.path.join(REF, "debugDoc")
NAMES = ["Serial Debug 2026-01-26 164520", "Serial Debug 2026-01-26 165358", "Serial Debug 2026-01-26 170305"]


def axes_box(img: np.ndarray):
    """(left, right, top, bottom) of the axes in pixels, from the four spines: each is a dark line two pixels wide
    centred on the axes edge, so the edge is the boundary between its two pixel columns / rows."""
    g = img[..., :3].astype(int).sum(-1)
    dark = g < 200
    rows = np.flatnonzero(dark.sum(1) > img.shape[1] // 3)
    cols = np.flatnonzero(dark.sum(0) > img.shape[0] // 3)
    # the mesh area lies between the first pair of dark columns and the next pair (the colour bar's come after)
    pairs_c = [c for c in cols if c + 1 in cols]
    pairs_r = [r for r in rows if r + 1 in rows]
    return float(pairs_c[0] + 1), float(pairs_c[1] + 1), float(pairs_r[0] + 1), float(pairs_r[-1] + 1)


def edges(v):
    v = np.asarray(v, np.float64)                     # excel_heatmap_v3.py:430-440
    if len(v) == 1:
        return np.array([v[0] - 0.5, v[0] + 0.5])
    e = np.empty(len(v) + 1)
    e[1:-1] = (v[:-1] + v[1:]) / 2.0
    e[0] = v[0] - (v[1] - v[0]) / 2.0
    e[-1] = v[-1] + (v[-1] - v[-2]) / 2.0
    return e


def cell_centre_pixels(box, aod, aoa):
    left, right, top, bottom = box
    xe, ye = edges(aoa), edges(aod)
    xc, yc = (xe[:-1] + xe[1:]) / 2, (ye[:-1] + ye[1:]) / 2
    px = np.floor(left + (xc - xe[0]) / (xe[-1] - xe[0]) * (right - left)).astype(int)
    py = np.floor(bottom - (yc - ye[0]) / (ye[-1] - ye[0]) * (bottom - top)).astype(int)
    return px, py


def main():
    t = xlsx.read_table(os.path.join(REF, "beam_angle.xlsx"))
    mapping = {int(i): float(a) for i, a in zip(t["BeamID"], t["Angle"]) if 0 <= int(i) <= 63}
    out = {}
    for k, name in enumerate(NAMES):
        tb = xlsx.read_table(os.path.join(D, name + "_filtered.xlsx"))
        rss_key = [c for c in tb if c.startswith("RSS")][0]
        ue, bs, rss = (np.asarray(tb[c], np.int64) for c in ("UE_Beam", "BS_Beam", rss_key))
        df = pd.DataFrame({"ue_beam": ue, "bs_beam": bs, "rss": rss.astype(np.float64)})
        mat, aod, aoa = E.compute_rssi_matrix(E.map_angles(df, mapping))
        img = np.array(Image.open(os.path.join(D, "heatmap_outputs", name + "_filtered_heatmap.png")).convert("RGBA"))
        box = axes_box(img)
        px, py = cell_centre_pixels(box, aod, aoa)
        cells = img[py[:, None], px[None, :]]
        out[f"f{k}_name"] = np.array(name)
        out[f"f{k}_rows"] = np.stack([ue, bs, rss], 1).astype(np.int32)
        out[f"f{k}_size"] = np.array(img.shape[:2])
        out[f"f{k}_box"] = np.array(box)
        out[f"f{k}_aod"] = np.array(aod)
        out[f"f{k}_aoa"] = np.array(aoa)
        out[f"f{k}_cells"] = cells.astype(np.uint8)
        print(name, "rows", len(ue), "matrix", mat.shape, "image", img.shape, "axes box", box)
    np.savez_compressed(os.path.join(HERE, "committed_png_cells.npz"), **out)


if __name__ == "__main__":
    main()
