"""Fixture for RSS values that are not whole numbers: tables of (ue_beam, bs_beam, rss) with fractional, negative,
tiny, huge and infinite RSS through the reference's own map_angles + compute_rssi_matrix (excel_heatmap_v3.py:264-314,
imported in the build container), stored as inputs and the matrices it returned.
Run:  python tests/golden/make_float_pivot.py  ->  tests/golden/float_pivot.npz
"""
import importlib.util
import json
import os
import warnings

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def tables():
    rng = np.random.default_rng(20260211)
    out = {}
    n = 6000
    ue, bs = rng.integers(0, 64, n), rng.integers(0, 64, n)
    out["dbm"] = (ue, bs, np.round(rng.normal(-71.3, 14.0, n), 2))                       # two-decimal dBm readings
    out["mixed_scales"] = (ue, bs, rng.normal(-70, 15, n) * rng.choice([1.0, 1e-6, 1e9], n))   # order matters most here
    ue2, bs2 = rng.integers(10, 14, 3000), rng.integers(20, 23, 3000)                   # few cells, long sums
    out["few_cells"] = (ue2, bs2, rng.uniform(-1, 1, 3000) * 10.0 ** rng.integers(-8, 9, 3000))
    r3 = rng.normal(0, 1, 500)
    r3[::50] = np.inf
    r3[7::90] = -np.inf
    out["infinities"] = (rng.integers(0, 6, 500), rng.integers(0, 6, 500), r3)
    out["halves"] = (rng.integers(0, 64, 2000), rng.integers(0, 64, 2000), rng.integers(30000, 180000, 2000) / 2.0)
    return out


def main():
    import pandas as pd
    import matplotlib
    matplotlib.use("Agg")
    spec = importlib.util.spec_from_file_location("ref_excel_heatmap_v3", os.path.join(REF, "excel_heatmap_v3.py"))
    E = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(E)
    mapping = {int(k): v for k, v in json.load(open(os.path.join(HERE, "beam_angle.json"))).items()}
    doc = {}
    warnings.simplefilter("ignore")
    for name, (ue, bs, rss) in tables().items():
        df = pd.DataFrame({"ue_beam": ue.astype(np.int64), "bs_beam": bs.astype(np.int64), "rss": rss.astype(np.float64)})
        mat, aod, aoa = E.compute_rssi_matrix(E.map_angles(df, mapping))
        doc[name + "_ue"], doc[name + "_bs"], doc[name + "_rss"] = ue.astype(np.int32), bs.astype(np.int32), rss.astype(np.float64)
        doc[name + "_matrix"], doc[name + "_aod"], doc[name + "_aoa"] = mat, np.array(aod), np.array(aoa)
        print(name, mat.shape, int(np.isfinite(mat).sum()), "finite cells")
    np.savez_compressed(os.path.join(HERE, "float_pivot.npz"), **doc)


if __name__ == "__main__":
    main()
