"""Fixture for the dictionary correlation of the reference's path estimators (heatmap_gemini_v4.py, imported in the build
container): LoSEstimator.construct_dictionary + the first line of its estimate_los_path loop (:182-198, :209) and
NLoSEstimator.construct_dictionary + the first line of estimate_nlos_paths (:283-296, :304), on the mean-RSS matrix of a
committed capture as BeamDataProcessor.pivot_data shapes it (UE rows, BS columns, empty cells filled with the minimum,
:117-131) and on a random matrix.  Stored: angles, matrices, the dictionaries and the correlations the reference's own
objects gave (the random case without its dictionaries: the oracle's are held against the capture case's).  Coarse
grids (1 and 2 degrees) keep the file small; the default 0.1-degree grid is covered by the oracle.
Run:  python tests/golden/make_dict_corr.py  ->  tests/golden/dict_corr.npz
"""
import importlib.util
import json
import os

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    import matplotlib
    matplotlib.use("Agg")
    spec = importlib.util.spec_from_file_location("ref_heatmap_gemini_v4", os.path.join(REF, "heatmap_gemini_v4.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    mapping = {int(k): v for k, v in json.load(open(os.path.join(HERE, "beam_angle.json"))).items()}
    kept = np.load(os.path.join(HERE, "log0.npz"))["kept"]            # ue, bs, rss, clk of a capture head, corrected
    import pandas as pd
    df = pd.DataFrame({"UE_Beam": kept[:, 0], "BS_Beam": kept[:, 1], "RSS": kept[:, 2].astype(np.float64)})
    avg = df.groupby(["UE_Beam", "BS_Beam"])["RSS"].mean().reset_index()          # pivot_data, :117-131
    piv = avg.pivot(index="UE_Beam", columns="BS_Beam", values="RSS").fillna(avg["RSS"].min())
    ue_angles = np.array([mapping[int(u)] for u in piv.index.values])
    bs_angles = np.array([mapping[int(b)] for b in piv.columns.values])
    rss = piv.values.astype(np.float64)
    rng = np.random.default_rng(20260212)
    all_angles = np.array([mapping[i] for i in range(64)])
    rnd = rng.uniform(15000.0, 93000.0, (64, 64))
    doc = {}
    for name, ue_a, bs_a, mat, res in (("capture", ue_angles, bs_angles, rss, 1.0), ("random", all_angles, all_angles, rnd, 2.0)):
        los = G.LoSEstimator(ue_a, bs_a, mat)
        los.construct_dictionary(grid_res=res, beam_width=1.4)
        corr_los = los.Phi_RX.T @ los.rss_vector.reshape(len(ue_a), len(bs_a)) @ los.Phi_TX          # :207-209, first round
        nlos = G.NLoSEstimator(beam_width_deg=1.4)
        nlos.construct_dictionary(ue_a, bs_a, grid_res=res)
        corr_nlos = nlos.Phi_RX.T @ mat @ nlos.Phi_TX                                                # :304
        doc.update({name + "_ue_angles": ue_a, name + "_bs_angles": bs_a, name + "_matrix": mat, name + "_grid_res": np.float64(res),
                    name + "_los_phi_rx": los.Phi_RX, name + "_los_phi_tx": los.Phi_TX, name + "_los_aoa": los.aoa_grid,
                    name + "_los_aod": los.aod_grid, name + "_los_corr": corr_los,
                    name + "_nlos_phi_rx": nlos.Phi_RX, name + "_nlos_phi_tx": nlos.Phi_TX, name + "_nlos_aoa": nlos.aoa_grid,
                    name + "_nlos_aod": nlos.aod_grid, name + "_nlos_corr": corr_nlos})
        if name == "random":
            for k in ("_los_phi_rx", "_los_phi_tx", "_nlos_phi_rx", "_nlos_phi_tx"):
                del doc[name + k]
        print(name, mat.shape, los.Phi_RX.shape, los.Phi_TX.shape, corr_los.shape, nlos.Phi_RX.shape, corr_nlos.shape)
    np.savez_compressed(os.path.join(HERE, "dict_corr.npz"), **doc)


if __name__ == "__main__":
    main()
