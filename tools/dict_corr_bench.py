"""Rate of the dictionary correlation (slam_dict_correlate_dev, f64 MFMA) on device-resident operands: F mean-RSS
matrices of 64 x 64 beams against the dictionaries of the reference's default 0.1-degree search grids.
python tools/dict_corr_bench.py [frames] [repeats]  ->  one JSON line (TFLOP/s against the f64 matrix peak, output GB/s)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-process_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from slamhip import _lib, ops  # noqa: E402
from slamhip.pipeline import default_mapping  # noqa: E402

F64_MATRIX_PEAK_TFLOPS = 78.6          # MI355X vector = matrix f64 peak
HBM_PEAK_GBS = 8000.0


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    mapping = default_mapping()
    angles = np.array([mapping[i] for i in range(64)])
    aoa, aod, prx, ptx = ops.los_dictionary(angles, angles)
    ga, gd = len(aoa), len(aod)
    dev = torch.device("cuda", 0)
    c = _lib.ctx(0)
    c.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    a, b = torch.from_numpy(prx.copy()).to(dev), torch.from_numpy(ptx.copy()).to(dev)
    m = torch.randn((F, 64, 64), dtype=torch.float64, device=dev) * 3.0e4
    out = torch.empty((F, ga, gd), dtype=torch.float64, device=dev)
    call = lambda: c.check(c.lib.slam_dict_correlate_dev(c.h, C.c_void_p(a.data_ptr()), 64, ga, C.c_void_p(b.data_ptr()), 64, gd,   # noqa: E731
                                                         C.c_void_p(m.data_ptr()), F, C.c_void_p(out.data_ptr())))
    call()
    torch.cuda.synchronize(dev)
    want = (a.T @ m[:2] @ b).cpu().numpy()
    err = float(np.abs(out[:2].cpu().numpy() - want).max() / np.abs(want).max())
    c.timing_reset()
    c.timing(True)
    for _ in range(reps):
        call()
    c.timing(False)
    launches, ms = c.timing_read()["dict_corr"]
    per = ms / launches * 1e-3
    flop = F * 2.0 * (ga * 64 * 64 + ga * 64 * gd)
    wbytes = F * ga * gd * 8.0
    print(json.dumps(dict(frames_per_launch=F, grid=[ga, gd], launches=launches, ms_per_launch=round(per * 1e3, 4),
                          us_per_frame=round(per / F * 1e6, 3), TFLOPs=round(flop / per / 1e12, 2),
                          frac_of_f64_matrix_peak=round(flop / per / 1e12 / F64_MATRIX_PEAK_TFLOPS, 4),
                          output_GBps=round(wbytes / per / 1e9, 1), frac_of_hbm_peak=round(wbytes / per / 1e9 / HBM_PEAK_GBS, 4),
                          flop_per_frame=flop / F, output_bytes_per_frame=wbytes / F,
                          max_rel_diff_vs_torch_f64=err)))


if __name__ == "__main__":
    main()
