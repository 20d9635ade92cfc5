"""NumPy-level wrappers of the host-pointer entry points of libslamhip.so.

Each function moves NumPy arrays through one C-ABI call; all arithmetic happens
in the HIP kernels.  The drop-in modules (serial_hex_to_excel_v3,
bs_beam_correction, excel_heatmap_v3) are thin layers over these.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import ptr

CYCLE, TOL, MOD_BASE = 61000, 500, 64      # bs_beam_correction.py:23-25


def _u8(buf) -> np.ndarray:
    if isinstance(buf, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(buf), dtype=np.uint8)
    return np.ascontiguousarray(np.asarray(buf), dtype=np.uint8)


def hex_unpack(text, device: int = 0) -> np.ndarray:
    """_read_bytes' data path: raw file bytes -> uint8 array of kept tokens."""
    c = _lib.ctx(device)
    t = _u8(text)
    out = np.empty(len(t) // 2 + 2, np.uint8)
    n = C.c_size_t(0)
    c.check(c.lib.slam_hex_unpack(c.h, ptr(t) if len(t) else None, len(t), ptr(out), len(out), C.byref(n)))
    return out[: n.value].copy()


def token_rejects(text, device: int = 0):
    """The tokens _read_bytes skips: (index among all tokens, byte begin, byte end) uint32 arrays, in text order."""
    c = _lib.ctx(device)
    t = _u8(text)
    n = C.c_size_t(0)
    cap = 1024
    while True:
        idx, beg, end = (np.empty(cap, np.uint32) for _ in range(3))
        rc = c.lib.slam_token_rejects(c.h, ptr(t) if len(t) else None, len(t), ptr(idx), ptr(beg), ptr(end), cap, C.byref(n))
        if rc == _lib.E_CAPACITY and n.value > cap:
            cap = n.value
            continue
        c.check(rc)
        k = n.value
        return idx[:k].copy(), beg[:k].copy(), end[:k].copy()


def _parse(fn_name: str, src: np.ndarray, cap: int, device: int):
    c = _lib.ctx(device)
    flag = np.empty(cap, np.uint8)
    ue = np.empty(cap, np.uint8)
    bs = np.empty(cap, np.uint8)
    rss = np.empty(cap, np.uint32)
    clk = np.empty(cap, np.uint32)
    nv, nd = C.c_size_t(0), C.c_size_t(0)
    fn = getattr(c.lib, fn_name)
    c.check(fn(c.h, ptr(src) if len(src) else None, len(src), ptr(flag), ptr(ue), ptr(bs), ptr(rss), ptr(clk),
               cap, C.byref(nv), C.byref(nd)))
    v = nv.value
    return dict(flag=flag[:v], ue=ue[:v], bs=bs[:v], rss=rss[:v], clk=clk[:v]), v, nd.value


def parse_records(data, device: int = 0):
    """_parse_groups' data path: bytes -> (columns dict, valid, discarded)."""
    b = _u8(data)
    return _parse("slam_parse_records", b, len(b) // 11 + 2, device)


def parse_rejects(data, n_discarded: Optional[int] = None, device: int = 0) -> np.ndarray:
    """Byte indices at which _parse_groups discards a group, in scan order (uint32 array)."""
    b = _u8(data)
    c = _lib.ctx(device)
    cap = len(b) + 1 if n_discarded is None else int(n_discarded) + 1
    pos = np.empty(cap, np.uint32)
    n = C.c_size_t()
    c.check(c.lib.slam_parse_rejects(c.h, _lib.ptr(b), len(b), _lib.ptr(pos), cap, C.byref(n)))
    return pos[: n.value].copy()


def parse_text(text, device: int = 0):
    """_read_bytes + _parse_groups fused: raw file bytes -> (columns dict, valid, discarded)."""
    t = _u8(text)
    return _parse("slam_parse_text", t, len(t) // 22 + 2, device)


def _i32(a, name: str) -> np.ndarray:
    arr = np.asarray(a)
    if arr.dtype.kind == "f":
        if not np.all(np.isfinite(arr)):
            raise ValueError(f"column {name} holds NaN/inf: not representable on the device")
        if not np.all(arr == np.round(arr)):
            raise ValueError(f"column {name} holds non-integers")
    a64 = arr.astype(np.int64)
    if a64.size and (a64.min() < -(2 ** 31) or a64.max() >= 2 ** 31):
        raise ValueError(f"column {name} does not fit int32")
    return np.ascontiguousarray(a64.astype(np.int32))


def correct(flag, ue, bs, rss, clk, cycle: int = CYCLE, tol: int = TOL, mod_base: int = MOD_BASE,
            baselines: Optional[Sequence[Tuple[int, int]]] = None, one_group: bool = False, device: int = 0):
    """Correction over a whole table (groups found from UE wraps), over the table as ONE group
    (`one_group`), or over one group with explicit `baselines`.

    Returns dict(kept=(ue, bs, rss, clk) arrays, gid, corrected (-1 = none), baselines=(clk_b, bs_b, gid),
    n_groups).
    """
    c = _lib.ctx(device)
    f, u, b, r, k = (_i32(x, n) for x, n in ((flag, "FLAG"), (ue, "UE"), (bs, "BS"), (rss, "RSS"), (clk, "CLK")))
    n = len(f)
    if not (len(u) == len(b) == len(r) == len(k) == n):
        raise ValueError("columns differ in length")
    ou, ob, orr, oc = (np.empty(n, np.int32) for _ in range(4))
    gid = np.empty(n, np.int32)
    corr = np.empty(n, np.int32)
    if baselines is not None:
        blc_in = _i32([x[0] for x in baselines], "baseline clk")
        blb_in = _i32([x[1] for x in baselines], "baseline bs")
        bl_n = len(baselines)
    else:
        blc_in = blb_in = None
        bl_n = -1
    cap_bl = max(n // 2 + 2, max(bl_n, 0) + 2)
    blc, blb, blg = (np.empty(cap_bl, np.int32) for _ in range(3))
    nk, nb, ng = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    c.check(c.lib.slam_correct(c.h, ptr(f), ptr(u), ptr(b), ptr(r), ptr(k), n, int(cycle), int(tol), int(mod_base),
                               1 if one_group else 0, ptr(blc_in), ptr(blb_in), bl_n,
                               ptr(ou), ptr(ob), ptr(orr), ptr(oc), C.byref(nk), ptr(gid), ptr(corr),
                               ptr(blc), ptr(blb), ptr(blg), C.byref(nb), C.byref(ng)))
    kk, bb = nk.value, nb.value
    return dict(kept=(ou[:kk], ob[:kk], orr[:kk], oc[:kk]), gid=gid, corrected=corr,
                baselines=(blc[:bb], blb[:bb], blg[:bb]), n_groups=ng.value)


def axis_from_mapping(mapping: Dict[int, float]):
    """BeamID->angle dict -> (beam_to_axis int16[64], sorted distinct angles f64).  Ids outside 0..63 are ignored
    (excel_heatmap_v3.py:246-247)."""
    items = [(int(k), float(a)) for k, a in mapping.items() if 0 <= int(k) <= 63 and not math.isnan(float(a))]
    angles = sorted({a for _, a in items})
    pos = {a: i for i, a in enumerate(angles)}
    b2a = np.full(64, -1, np.int16)
    for k, a in items:
        b2a[k] = pos[a]
    return b2a, np.array(angles, np.float64)


def upscaled_axis(mapping: Dict[int, float], factor: int = 4):
    """The up-scaled beam table: the mapping's sorted distinct angles interpolated linearly to `factor` times as many
    entries (entry factor*i is angle i; the entries past the last angle continue its last step), and every beam at
    entry factor * (its index among the distinct angles).  Returns (beam_to_axis int16[64], angles f64).

    The reference has no such mode (its validate_beam_id admits 64 beams, excel_heatmap_v3.py:68-72): this is the
    benchmark configuration "4x upscaled beam table", rendered over the full grid (slam_batch.full_grid)."""
    b2a, angles = axis_from_mapping(mapping)
    factor = int(factor)
    if factor < 1 or len(angles) * factor > _lib.MAX_AXIS:
        raise ValueError(f"up-scaled table of {len(angles)} x {factor} entries exceeds {_lib.MAX_AXIS}")
    if factor == 1 or len(angles) == 0:
        return b2a, angles
    n = len(angles)
    j = np.arange(n * factor, dtype=np.float64) / factor
    if n == 1:
        up = angles[0] + j                      # a lone angle: unit steps, like _edges' +-0.5 around a single value
    else:
        i0 = np.minimum(np.floor(j).astype(np.int64), n - 2)
        up = angles[i0] + (j - i0) * (angles[i0 + 1] - angles[i0])
    up[::factor] = angles                       # the table's own angles exactly
    return np.where(b2a >= 0, b2a * factor, -1).astype(np.int16), np.ascontiguousarray(up, np.float64)


def pivot_mean(ue, bs, rss, beam_to_axis: np.ndarray, n_axis: int, device: int = 0):
    """map_angles + compute_rssi_matrix's data path: (sum int64, cnt int64, mean f64) on the full axis grid."""
    c = _lib.ctx(device)
    u, b, r = (_i32(x, n) for x, n in ((ue, "ue_beam"), (bs, "bs_beam"), (rss, "rss")))
    b2a = np.ascontiguousarray(beam_to_axis, dtype=np.int16)
    s = np.zeros((n_axis, n_axis), np.int64)
    k = np.zeros((n_axis, n_axis), np.int64)
    m = np.empty((n_axis, n_axis), np.float64)
    c.check(c.lib.slam_pivot_mean(c.h, ptr(u), ptr(b), ptr(r), len(u), ptr(b2a), int(n_axis), ptr(s), ptr(k), ptr(m)))
    return s, k, m


def pivot_mean_f64(ue, bs, rss, beam_to_axis: np.ndarray, n_axis: int, device: int = 0):
    """The pivot for RSS values of any kind (fractions, negative dBm): per cell pandas' compensated sum in row order,
    the count and the mean, (sum f64, cnt int64, mean f64) on the full axis grid."""
    c = _lib.ctx(device)
    u, b = (_i32(x, n) for x, n in ((ue, "ue_beam"), (bs, "bs_beam")))
    r = np.ascontiguousarray(rss, dtype=np.float64)
    if len(r) != len(u):
        raise ValueError("columns of different lengths")
    b2a = np.ascontiguousarray(beam_to_axis, dtype=np.int16)
    s = np.zeros((n_axis, n_axis), np.float64)
    k = np.zeros((n_axis, n_axis), np.int64)
    m = np.empty((n_axis, n_axis), np.float64)
    c.check(c.lib.slam_pivot_mean_f64(c.h, ptr(u), ptr(b), ptr(r), len(u), ptr(b2a), int(n_axis), ptr(s), ptr(k), ptr(m)))
    return s, k, m


def blur_nan(data: np.ndarray, sigma: float, device: int = 0) -> np.ndarray:
    """gaussian_blur_nan_aware's data path."""
    c = _lib.ctx(device)
    a = np.ascontiguousarray(data, dtype=np.float64)
    out = np.empty_like(a)
    c.check(c.lib.slam_blur_nan(c.h, ptr(a), a.shape[0], a.shape[1], float(sigma), ptr(out)))
    return out


def render(mat: np.ndarray, aod, aoa, lut: np.ndarray, use_log: bool = False, vmin: Optional[float] = None,
           vmax: Optional[float] = None, W: int = 1024, H: int = 1024, want_tile: bool = True, device: int = 0):
    """generate_heatmap's data path after the blur: (cells RGBA [h,w,4], tile RGBA [H,W,4] or None, vmin, vmax)."""
    c = _lib.ctx(device)
    m = np.ascontiguousarray(mat, dtype=np.float64)
    h, w = m.shape
    ad = np.ascontiguousarray(aod, dtype=np.float64)
    aa = np.ascontiguousarray(aoa, dtype=np.float64)
    l = np.ascontiguousarray(lut, dtype=np.uint8).reshape(256, 3)
    cells = np.empty((h, w, 4), np.uint8)
    tile = np.empty((H, W, 4), np.uint8) if want_tile else None
    vv = np.empty(2, np.float64)
    nan = float("nan")
    c.check(c.lib.slam_render(c.h, ptr(m), h, w, ptr(ad), ptr(aa), 1 if use_log else 0,
                              nan if vmin is None else float(vmin), nan if vmax is None else float(vmax),
                              ptr(l), ptr(cells), ptr(tile), int(W), int(H), ptr(vv)))
    return cells, tile, float(vv[0]), float(vv[1])


# ---- next to the path: the dictionary correlation of the reference's path estimators ---------------------------
def los_dictionary(ue_angles, bs_angles, grid_res: float = 0.1, beam_width: float = 1.4):
    """LoSEstimator.construct_dictionary (heatmap_gemini_v4.py:182-198) with its _gaussian_beam (:175-180): search grids
    from the smallest to short of the largest angle, Gaussian beams of FWHM beam_width.  Host-side set-up, done once per
    angle set.  Returns (aoa_grid, aod_grid, Phi_RX [n_ue, len(aoa_grid)], Phi_TX [n_bs, len(aod_grid)])."""
    ue, bs = np.asarray(ue_angles, np.float64), np.asarray(bs_angles, np.float64)
    sigma = beam_width / 2.355
    aoa = np.arange(np.min(ue), np.max(ue), grid_res)
    aod = np.arange(np.min(bs), np.max(bs), grid_res)
    beam = lambda x, c: np.exp(-((x - c) ** 2) / (2 * sigma ** 2))          # noqa: E731
    return aoa, aod, beam(ue[:, None], aoa[None, :]), beam(bs[:, None], aod[None, :])


def nlos_dictionary(ue_angles, bs_angles, grid_res: float = 0.1, beam_width: float = 1.4):
    """NLoSEstimator.construct_dictionary (heatmap_gemini_v4.py:283-296) with its _gaussian_beam (:271-281): the grids
    run up to and including the largest angle.  Same return as los_dictionary."""
    ue, bs = np.asarray(ue_angles, np.float64), np.asarray(bs_angles, np.float64)
    sigma = beam_width / 2.355
    aoa = np.arange(np.min(ue), np.max(ue) + grid_res, grid_res)
    aod = np.arange(np.min(bs), np.max(bs) + grid_res, grid_res)

    def beam(query, centers):
        q, c = np.meshgrid(query, centers)
        return np.exp(-((q - c) ** 2) / (2 * sigma ** 2))
    return aoa, aod, beam(aoa, ue), beam(aod, bs)


def dict_correlate(phi_rx: np.ndarray, m: np.ndarray, phi_tx: np.ndarray, device: int = 0) -> np.ndarray:
    """correlation = Phi_RX.T @ res_mat @ Phi_TX (heatmap_gemini_v4.py:209, :304) on the device, float64.
    m [n_ue, n_bs] gives [ga, gd]; a stack m [F, n_ue, n_bs] gives [F, ga, gd]."""
    c = _lib.ctx(device)
    a = np.ascontiguousarray(phi_rx, dtype=np.float64)
    b = np.ascontiguousarray(phi_tx, dtype=np.float64)
    mm = np.ascontiguousarray(m, dtype=np.float64)
    single = mm.ndim == 2
    if single:
        mm = mm[None]
    if a.ndim != 2 or b.ndim != 2 or mm.ndim != 3 or mm.shape[1:] != (a.shape[0], b.shape[0]):
        raise ValueError(f"shapes do not chain: Phi_RX {a.shape}, matrix {mm.shape[1:]}, Phi_TX {b.shape}")
    out = np.empty((mm.shape[0], a.shape[1], b.shape[1]), np.float64)
    c.check(c.lib.slam_dict_correlate(c.h, ptr(a), a.shape[0], a.shape[1], ptr(b), b.shape[0], b.shape[1], ptr(mm),
                                      mm.shape[0], ptr(out)))
    return out[0] if single else out
