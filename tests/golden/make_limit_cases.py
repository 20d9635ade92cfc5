"""Golden colour-scale results for given vmin / vmax, from the reference's generate_heatmap (excel_heatmap_v3.py:374-475).

Run in the build container:  python tests/golden/make_limit_cases.py
Small matrices with holes; limits inside, outside, equal and reversed, on the linear and on the shifted log scale.
For every case either the exception the reference raises (from matplotlib's norm) or the limits of the mesh's norm
and its face colours after drawing, as round(rgba * 255).  Writes limit_cases.json.
"""
import json
import os
import sys
import warnings

import numpy as np
import matplotlib

matplotlib.use("Agg")
warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import excel_heatmap_v3 as E  # noqa: E402


def run(mat, aod, aoa, **kw):
    figs = []
    orig_close = E.plt.close
    E.plt.close = lambda fig=None: figs.append(fig)
    try:
        E.generate_heatmap(mat, aod, aoa, "/tmp/_limit_case.png", "t", colormap="viridis", blur_sigma=0.0, logger=None, **kw)
    finally:
        E.plt.close = orig_close
    fig = figs[0]
    mesh = fig.axes[0].collections[0]
    fc = np.asarray(mesh.get_facecolors())
    out = dict(vmin=float(mesh.norm.vmin), vmax=float(mesh.norm.vmax),
               cells=np.round(fc * 255).astype(int).reshape(mat.shape[0], mat.shape[1], 4).tolist())
    orig_close(fig)
    return out


def main():
    rng = np.random.default_rng(12)
    mats = {}
    m = rng.integers(15000, 93000, (5, 7)).astype(np.float64)
    m[1, 2] = m[3, 3] = m[4, 0] = np.nan
    mats["a"] = m
    mats["b"] = np.array([[100.0, 200.0, np.nan], [300.0, 400.0, 500.0]])
    mats["flat"] = np.full((3, 3), 42.0)
    doc = {}
    for name, mat in mats.items():
        lo, hi = float(np.nanmin(mat)), float(np.nanmax(mat))
        mid = (lo + hi) / 2
        limits = [dict(), dict(vmin=lo, vmax=hi), dict(vmin=mid), dict(vmax=mid), dict(vmin=lo - 10, vmax=hi + 10),
                  dict(vmin=mid, vmax=mid), dict(vmin=hi, vmax=lo), dict(vmin=hi + 5), dict(vmax=lo - 5), dict(vmin=lo - 1),
                  dict(vmax=lo), dict(vmin=mid, vmax=mid + 1e-3)]
        aod = [float(i) * 2.5 - 3 for i in range(mat.shape[0])]
        aoa = [float(j) * 1.5 - 4 for j in range(mat.shape[1])]
        for use_log in (False, True):
            for k, lim in enumerate(limits):
                key = f"{name}_{'log' if use_log else 'lin'}_{k:02d}"
                e = dict(matrix=[[None if np.isnan(v) else v for v in r] for r in mat.tolist()], aod=aod, aoa=aoa, use_log=use_log,
                         vmin_in=lim.get("vmin"), vmax_in=lim.get("vmax"))
                try:
                    e.update(run(mat, aod, aoa, use_log=use_log, **lim))
                except Exception as ex:
                    e.update(error=type(ex).__name__, message=str(ex))
                doc[key] = e
    json.dump(doc, open(os.path.join(HERE, "limit_cases.json"), "w"))
    print(len(doc), {k: (v.get("error"), v.get("message")) for k, v in doc.items() if "error" in v})


if __name__ == "__main__":
    main()
