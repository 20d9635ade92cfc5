"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

A restatement of the reference's parse -> correct -> render-matrix algorithms
used to check the HIP path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product package never
does.

Integer stages (tokenise, frame, group, baseline, correct, pivot sums) are the
sequential C restatement in slam_oracle.c, reached through ctypes.  Float
stages (mean, Gaussian kernel, NaN-aware blur, edges, normalise, colour, raster
index maps) are NumPy float64.

Pinned against the reference's own functions (imported in the build container
by tests/golden/make_golden.py and by tests that skip when /root/reference is
absent), the committed debugDoc/*.xlsx outputs and the reference's embedded
unit tests.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_SRC = os.path.join(_HERE, "slam_oracle.c")

CYCLE = 61000      # bs_beam_correction.py:23
TOL = 500          # bs_beam_correction.py:24
MOD_BASE = 64      # bs_beam_correction.py:25


def build(force: bool = False) -> str:
    """Compile slam_oracle.c with gcc (also done by `make -C oracle`)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-o", _SO, _SRC, "-lm"]
        )
    return _SO


_lib = None


def _L():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = ctypes.CDLL(_SO)
        i64 = ctypes.c_int64
        p = ctypes.c_void_p
        lib.orc_read_bytes.restype = i64
        lib.orc_read_bytes.argtypes = [p, i64, p]
        lib.orc_parse_groups.restype = None
        lib.orc_parse_groups.argtypes = [p, i64, p, p, p, p, p, p, p, p, p, i64]
        lib.orc_detect_groups.restype = i64
        lib.orc_detect_groups.argtypes = [p, i64, p]
        lib.orc_identify_baselines.restype = i64
        lib.orc_identify_baselines.argtypes = [p, p, p, p, p, i64, p, p, p]
        lib.orc_filter_corrected.restype = i64
        lib.orc_filter_corrected.argtypes = [p, p, p, p, p, i64, i64, i64, i64, p, p, p, p, p, p]
        lib.orc_correct_inplace.restype = None
        lib.orc_correct_inplace.argtypes = [p, p, p, p, p, i64, i64, i64, i64, p, p]
        lib.orc_pivot.restype = None
        lib.orc_pivot.argtypes = [p, p, p, i64, p, ctypes.c_int, p, p]
        _lib = lib
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.int64)


# ---------------------------------------------------------------------------
# a1  serial_hex_to_excel_v3._read_bytes  (:18-36)
# ---------------------------------------------------------------------------
def read_bytes(text: bytes) -> np.ndarray:
    """Raw file bytes -> uint8 array of the kept hex tokens."""
    buf = np.frombuffer(bytes(text), dtype=np.uint8)
    out = np.empty(len(buf) // 2 + 2, dtype=np.uint8)
    n = _L().orc_read_bytes(_ptr(buf) if len(buf) else None, len(buf), _ptr(out))
    return out[:n].copy()


def read_bytes_py(text: bytes) -> np.ndarray:
    """The same through CPython's own decoder/split (small inputs; cross-check of the C tokenizer)."""
    out = []
    for t in bytes(text).decode("utf-8", errors="ignore").split():
        s = t
        if len(s) == 4 and s[:2] == "0x":
            s = s[2:]
        if len(s) == 2 and all(c in "0123456789abcdefABCDEF" for c in s):
            out.append(int(s, 16))
    return np.array(out, dtype=np.uint8)


def skipped_tokens_py(text: bytes) -> List[Tuple[int, str]]:
    """(token index, token) for every token _read_bytes skips (:22-28), through CPython's decoder and split."""
    out = []
    for idx, t in enumerate(bytes(text).decode("utf-8", errors="ignore").split()):
        s = t[2:] if len(t) == 4 and t[:2] == "0x" else t
        if not (len(s) == 2 and all(c in "0123456789abcdefABCDEF" for c in s)):
            out.append((idx, t))
    return out


# ---------------------------------------------------------------------------
# a2  serial_hex_to_excel_v3._parse_groups  (:41-103)
# ---------------------------------------------------------------------------
def parse_groups(data, want_rejects: bool = False):
    """bytes -> dict(flag, ue, bs, rss, clk arrays), valid, discarded[, reject_pos, reject_reason]."""
    b = np.ascontiguousarray(np.asarray(data, dtype=np.uint8))
    n = len(b)
    cap = n // 11 + 1
    flag = np.empty(cap, np.uint8)
    ue = np.empty(cap, np.uint8)
    bs = np.empty(cap, np.uint8)
    rss = np.empty(cap, np.uint32)
    clk = np.empty(cap, np.uint32)
    nv = ctypes.c_int64(0)
    nd = ctypes.c_int64(0)
    rp = np.empty(n + 1 if want_rejects else 1, np.int64)
    rr = np.empty(n + 1 if want_rejects else 1, np.uint8)
    _L().orc_parse_groups(
        _ptr(b) if n else None, n, _ptr(flag), _ptr(ue), _ptr(bs), _ptr(rss), _ptr(clk),
        ctypes.byref(nv), ctypes.byref(nd),
        _ptr(rp) if want_rejects else None, _ptr(rr) if want_rejects else None,
        len(rp) if want_rejects else 0,
    )
    v, d = nv.value, nd.value
    rows = dict(flag=flag[:v].copy(), ue=ue[:v].copy(), bs=bs[:v].copy(),
                rss=rss[:v].copy(), clk=clk[:v].copy())
    if want_rejects:
        return rows, v, d, rp[:d].copy(), rr[:d].copy()
    return rows, v, d


def parse_groups_py(data: Sequence[int]):
    """Literal pure-Python loop of the same scan (tiny inputs; cross-check of the C scan)."""
    b = [int(x) for x in data]
    n = len(b)
    rows: List[Tuple[int, int, int, int, int]] = []
    disc = 0
    i = 0
    while i < n:
        f = b[i]
        if f not in (0xCC, 0x33):
            i += 1
            continue
        if i + 11 > n:
            disc += 1
            break
        g = b[i + 1:i + 11]
        tags = [(x >> 6) & 3 for x in g]
        if tags != [0, 3, 1, 1, 1, 1, 1, 2, 2, 2]:
            disc += 1
            i += 1
            continue
        clk = sum((g[2 + j] & 63) << (6 * j) for j in range(5))
        rss = sum((g[7 + j] & 63) << (6 * j) for j in range(3))
        rows.append((1 if f == 0xCC else 0, g[0] & 63, g[1] & 63, rss, clk))
        i += 11
    return rows, len(rows), disc


# ---------------------------------------------------------------------------
# a4-a6  bs_beam_correction
# ---------------------------------------------------------------------------
def detect_groups(ue) -> np.ndarray:
    """bs_beam_correction._detect_groups (:43-47)."""
    u = _i64(ue)
    gid = np.empty(len(u), np.int64)
    _L().orc_detect_groups(_ptr(u), len(u), _ptr(gid))
    return gid


def identify_baselines(flag, ue, bs, rss, clk):
    """All groups' baselines in order: (clk_b, bs_b, gid) arrays.  _identify_baselines (:50-61)."""
    f, u, b, r, c = map(_i64, (flag, ue, bs, rss, clk))
    n = len(f)
    oc = np.empty(n + 1, np.int64)
    ob = np.empty(n + 1, np.int64)
    og = np.empty(n + 1, np.int64)
    k = _L().orc_identify_baselines(_ptr(f), _ptr(u), _ptr(b), _ptr(r), _ptr(c), n,
                                    _ptr(oc), _ptr(ob), _ptr(og))
    return oc[:k].copy(), ob[:k].copy(), og[:k].copy()


def filter_corrected(flag, ue, bs, rss, clk, cycle=CYCLE, tol=TOL, mod_base=MOD_BASE):
    """process_excel_filtered's data path (:186-198 + :140-178): kept rows (ue, bs', rss, clk), n_groups, n_baselines."""
    f, u, b, r, c = map(_i64, (flag, ue, bs, rss, clk))
    n = len(f)
    ou, ob, orr, oc = (np.empty(n + 1, np.int64) for _ in range(4))
    ng = ctypes.c_int64(0)
    nb = ctypes.c_int64(0)
    k = _L().orc_filter_corrected(_ptr(f), _ptr(u), _ptr(b), _ptr(r), _ptr(c), n,
                                  int(cycle), int(tol), int(mod_base),
                                  _ptr(ou), _ptr(ob), _ptr(orr), _ptr(oc),
                                  ctypes.byref(ng), ctypes.byref(nb))
    return dict(ue=ou[:k].copy(), bs=ob[:k].copy(), rss=orr[:k].copy(), clk=oc[:k].copy()), ng.value, nb.value


def correct_inplace(flag, ue, bs, rss, clk, cycle=CYCLE, tol=TOL, mod_base=MOD_BASE):
    """process_excel's data path (_correct_group :79-116 over all groups): corrected_bs per row, was_corrected mask."""
    f, u, b, r, c = map(_i64, (flag, ue, bs, rss, clk))
    n = len(f)
    out = np.empty(n, np.int64)
    wc = np.empty(n, np.uint8)
    _L().orc_correct_inplace(_ptr(f), _ptr(u), _ptr(b), _ptr(r), _ptr(c), n,
                             int(cycle), int(tol), int(mod_base), _ptr(out), _ptr(wc))
    return out, wc.astype(bool)


def filter_group_py(rows, baselines, cycle=CYCLE, tol=TOL, mod_base=MOD_BASE):
    """Literal loop of _filter_group_corrected (:140-178) for ONE group with explicit baselines (small inputs)."""
    out = []
    for flag, ue, bs, rss, clk in rows:
        if int(flag) == 1 or not baselines:
            continue
        cands = []
        for clk_b, bs_b in baselines:
            d = int(clk) - int(clk_b)
            k = int(round(d / cycle))
            resid = abs(d - k * cycle)
            if resid <= tol:
                cands.append((resid, (int(bs_b) + k) % mod_base))
        if cands:
            cands.sort(key=lambda x: x[0])
            out.append([int(ue), int(cands[0][1]), int(rss), int(clk)])
    return out


def correct_table_nulls_py(rows, filtered: bool, cycle=CYCLE, tol=TOL, mod_base=MOD_BASE):
    """The correction drivers on a table whose cells may be None (a NaN after _coerce_types :35-40), as a
    literal loop (small inputs).  Returns (gid list, dict(result=...) or dict(error="ValueError", message=...)).

    Null semantics of the reference: a row after a null UE starts a group and a null UE never does (:45);
    a null FLAG is neither 0 nor 1 for the baseline mask (:53) and counts as 0 in the row loops (:84, :144);
    a null RSS equals nothing (:53); a null CLK skips the row (:91-94, :149-151); every other int() of a
    null cell raises ValueError (:58-59, :86-107, :164-167).  filtered=True gives the kept rows of
    process_excel_filtered (:186-198), False the Corrected_BS_Beam column of process_excel (:125-134).
    """
    def to_int(v):
        if v is None:
            raise ValueError("cannot convert float NaN to integer")
        return int(v)

    n = len(rows)
    gid, g = [], -1
    for i, r in enumerate(rows):
        prev = rows[i - 1][1] if i else None
        if i == 0 or prev is None or (r[1] is not None and prev > r[1]):
            g += 1
        gid.append(g)
    try:
        kept, col = [], [None] * n
        start = 0
        while start < n:
            end = start
            while end < n and gid[end] == gid[start]:
                end += 1
            bl = []
            for i in range(start + 1, end):
                f, pf, rs, prs = rows[i][0], rows[i - 1][0], rows[i][3], rows[i - 1][3]
                if f == 1 and pf == 0 and rs is not None and prs is not None and rs == prs:
                    clk_b = to_int(rows[i - 1][4])
                    bl.append((clk_b, to_int(rows[i][2])))
            for i in range(start, end):
                flag, ue, bs, rss, clk = rows[i]
                flag = 0 if flag is None else int(flag)
                best = None
                if flag != 1 and bl and clk is not None:
                    for clk_b, bs_b in bl:
                        d = int(clk) - clk_b
                        k = int(round(d / cycle))
                        resid = abs(d - k * cycle)
                        if resid <= tol and (best is None or resid < best[0]):
                            best = (resid, (bs_b + k) % mod_base)
                if filtered:
                    if best is not None:
                        kept.append([to_int(ue), int(best[1]), to_int(rss), to_int(clk)])
                else:
                    col[i] = int(best[1]) if best is not None else to_int(bs)
            start = end
        if filtered and not kept:
            raise ValueError("未产生任何可修正数据行，输出为空")
        return gid, dict(result=kept if filtered else col)
    except ValueError as e:
        return gid, dict(error="ValueError", message=str(e))


# ---------------------------------------------------------------------------
# a8-a10  mapping table, map_angles, compute_rssi_matrix
# ---------------------------------------------------------------------------
def axis_from_mapping(mapping: Dict[int, float]) -> Tuple[np.ndarray, np.ndarray]:
    """BeamID->angle dict (ids 0..63, excel_heatmap_v3.py:246-247) -> (beam_to_axis int16[64], sorted distinct angles)."""
    angles = sorted({float(a) for k, a in mapping.items() if 0 <= int(k) <= 63 and not math.isnan(float(a))})
    pos = {a: i for i, a in enumerate(angles)}
    b2a = np.full(64, -1, np.int16)
    for k, a in mapping.items():
        if 0 <= int(k) <= 63 and not math.isnan(float(a)):
            b2a[int(k)] = pos[float(a)]
    return b2a, np.array(angles, np.float64)


def upscaled_axis(mapping: Dict[int, float], factor: int):
    """The benchmark's "4x upscaled beam table" -- PARITY UNPINNED: the reference has no such mode (validate_beam_id
    admits beams 0..63 only, excel_heatmap_v3.py:68-72), so this restates the build's own definition, not a
    reference function.  The sorted distinct angles of the table are interpolated linearly to `factor` times as many
    entries: entry factor*i is angle i exactly, entry factor*i + k lies k/factor of the way to angle i+1, and the
    entries after the last angle continue its last step.  A beam whose angle is the i-th distinct one sits at entry
    factor*i.  Returns (beam_to_axis int16[64], angles f64)."""
    b2a, angles = axis_from_mapping(mapping)
    n = len(angles)
    if factor == 1 or n == 0:
        return b2a, angles
    up = np.empty(n * factor, np.float64)
    for j in range(n * factor):
        x = np.float64(j) / np.float64(factor)
        if n == 1:
            up[j] = angles[0] + x
        else:
            i0 = min(j // factor, n - 2)
            up[j] = angles[i0] + (x - np.float64(i0)) * (angles[i0 + 1] - angles[i0])
    up[::factor] = angles
    return np.where(b2a >= 0, b2a.astype(np.int64) * factor, -1).astype(np.int16), up


def pivot_mean(ue, bs, rss, mapping: Dict[int, float], upscale: int = 1):
    """map_angles (:264-280) + compute_rssi_matrix (:286-314).

    Returns (matrix f64 [n_aod, n_aoa] with NaN for empty cells, aod_list, aoa_list,
    sum int64 [n_axis,n_axis], cnt int64 [n_axis,n_axis]).  Axes hold only angles that occur.
    upscale > 1 (parity unpinned, see upscaled_axis): the table is the up-scaled one and the matrix is the full
    grid, every entry of the table a row and a column whether data fell on it or not.
    """
    b2a, angles = upscaled_axis(mapping, int(upscale))
    n_axis = len(angles)
    u, b, r = map(_i64, (ue, bs, rss))
    s = np.zeros((n_axis, n_axis), np.int64)
    c = np.zeros((n_axis, n_axis), np.int64)
    if n_axis:
        _L().orc_pivot(_ptr(u), _ptr(b), _ptr(r), len(u), _ptr(b2a), n_axis, _ptr(s), _ptr(c))
    rows = np.flatnonzero(c.sum(axis=1) > 0)
    cols = np.flatnonzero(c.sum(axis=0) > 0)
    if upscale > 1:
        rows = cols = np.arange(n_axis)
    with np.errstate(invalid="ignore", divide="ignore"):
        full = np.where(c > 0, s.astype(np.float64) / c.astype(np.float64), np.nan)
    mat = full[np.ix_(rows, cols)]
    return mat, angles[rows].tolist(), angles[cols].tolist(), s, c


def pivot_mean_float(ue, bs, rss, mapping: Dict[int, float]):
    """compute_rssi_matrix (:286-314) for RSS values that are not whole numbers.  pandas' pivot_table(aggfunc=mean)
    (pandas 2.x group_mean; not vendored in the reference, pinned by tests against the reference's own function) adds
    the values of a cell in row order with Kahan compensation -- y = v - c; t = s + y; c = (t - s) - y, a NaN
    compensation (infinite sums) reset to 0 -- skips NaN values, and divides the sum by the number added.
    Returns (matrix over the angles that occur, aod_list, aoa_list, sum f64, cnt int64 on the full axis grid)."""
    b2a, angles = axis_from_mapping(mapping)
    n_axis = len(angles)
    u, b = _i64(ue), _i64(bs)
    r = np.ascontiguousarray(np.asarray(rss), dtype=np.float64)
    s = np.zeros((n_axis, n_axis), np.float64)
    comp = np.zeros((n_axis, n_axis), np.float64)
    c = np.zeros((n_axis, n_axis), np.int64)
    for i in range(len(u)):
        if not (0 <= u[i] <= 63 and 0 <= b[i] <= 63) or r[i] != r[i]:
            continue
        col, row = int(b2a[u[i]]), int(b2a[b[i]])
        if col < 0 or row < 0:
            continue
        with np.errstate(invalid="ignore", over="ignore"):
            y = r[i] - comp[row, col]
            t = s[row, col] + y
            cc = (t - s[row, col]) - y
        comp[row, col] = 0.0 if cc != cc else cc
        s[row, col] = t
        c[row, col] += 1
    rows = np.flatnonzero(c.sum(axis=1) > 0)
    cols = np.flatnonzero(c.sum(axis=0) > 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        full = np.where(c > 0, s / np.maximum(c, 1), np.nan)
    return full[np.ix_(rows, cols)], angles[rows].tolist(), angles[cols].tolist(), s, c


def pivot_mean_float_lockstep(ue, bs, rss, mapping: Dict[int, float]):
    """pivot_mean_float for large tables: the same recurrence, but the rows are first brought together cell by cell in
    row order (stable argsort) and step k adds the k-th value of every cell at once.  Same bits as the row-by-row loop
    (tests/test_oracle_golden.py holds the two against each other); returns the same tuple."""
    b2a, angles = axis_from_mapping(mapping)
    n_axis = len(angles)
    u, b = _i64(ue), _i64(bs)
    r = np.ascontiguousarray(np.asarray(rss), dtype=np.float64)
    ok = (u >= 0) & (u <= 63) & (b >= 0) & (b <= 63) & ~np.isnan(r)
    col = np.where(ok, b2a[np.clip(u, 0, 63)], -1).astype(np.int64)
    row = np.where(ok, b2a[np.clip(b, 0, 63)], -1).astype(np.int64)
    ok &= (col >= 0) & (row >= 0)
    cell = (row * n_axis + col)[ok]
    vals = r[ok]
    order = np.argsort(cell, kind="stable")
    cell, vals = cell[order], vals[order]
    cnt = np.bincount(cell, minlength=n_axis * n_axis).astype(np.int64)
    start = np.concatenate(([0], np.cumsum(cnt)[:-1]))
    s = np.zeros(n_axis * n_axis, np.float64)
    comp = np.zeros(n_axis * n_axis, np.float64)
    live = np.flatnonzero(cnt > 0)
    k = 0
    with np.errstate(invalid="ignore", over="ignore"):
        while len(live):
            v = vals[start[live] + k]
            y = v - comp[live]
            t = s[live] + y
            cc = (t - s[live]) - y
            comp[live] = np.where(np.isnan(cc), 0.0, cc)
            s[live] = t
            k += 1
            live = live[cnt[live] > k]
    s = s.reshape(n_axis, n_axis)
    c = cnt.reshape(n_axis, n_axis)
    rows = np.flatnonzero(c.sum(axis=1) > 0)
    cols = np.flatnonzero(c.sum(axis=0) > 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        full = np.where(c > 0, s / np.maximum(c, 1), np.nan)
    return full[np.ix_(rows, cols)], angles[rows].tolist(), angles[cols].tolist(), s, c


# ---------------------------------------------------------------------------
# a11-a12  gaussian_kernel (:320-333), gaussian_blur_nan_aware (:336-368)
# ---------------------------------------------------------------------------
def gaussian_kernel(sigma: float) -> np.ndarray:
    if sigma <= 0:
        return np.array([[1.0]], dtype=np.float64)
    size = int(max(3, math.ceil(6 * sigma)))
    if size % 2 == 0:
        size += 1
    c = size // 2
    ax = np.arange(-c, c + 1, dtype=np.int64)
    k = np.exp(-(ax[None, :] * ax[None, :] + ax[:, None] * ax[:, None]) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float64)


def blur_nan_aware(data: np.ndarray, sigma: float) -> np.ndarray:
    """Normalised convolution with edge padding; NaN where the weight sum is <= 1e-12."""
    data = np.asarray(data, dtype=np.float64)
    if sigma <= 0:
        return data
    k = gaussian_kernel(sigma)
    kh, kw = k.shape
    ph, pw = kh // 2, kw // 2
    mask = np.isfinite(data).astype(np.float64)
    filled = np.where(np.isfinite(data), data, 0.0)
    pd_ = np.pad(filled, ((ph, ph), (pw, pw)), mode="edge")
    pm = np.pad(mask, ((ph, ph), (pw, pw)), mode="edge")
    H, W = data.shape
    num = np.zeros((H, W), np.float64)
    den = np.zeros((H, W), np.float64)
    # plain row-major accumulation order over the kernel taps
    for a in range(kh):
        for b in range(kw):
            w = k[a, b] * pm[a:a + H, b:b + W]
            num += pd_[a:a + H, b:b + W] * w
            den += w
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(den > 1e-12, num / den, np.nan)


# ---------------------------------------------------------------------------
# a13  generate_heatmap's data path (:396-456): edges, normalise, colour
# ---------------------------------------------------------------------------
def edges(vals: Sequence[float]) -> np.ndarray:
    """_edges (:430-440): interior midpoints, outer edges half a step out; a single value gives +-0.5."""
    v = np.asarray(vals, dtype=np.float64)
    if len(v) == 1:
        return np.array([v[0] - 0.5, v[0] + 0.5], dtype=np.float64)
    e = np.empty(len(v) + 1, np.float64)
    e[1:-1] = (v[:-1] + v[1:]) / 2.0
    e[0] = v[0] - (v[1] - v[0]) / 2.0
    e[-1] = v[-1] + (v[-1] - v[-2]) / 2.0
    return e


def nonsingular_limits(lo: float, hi: float) -> Tuple[float, float]:
    """What the colour bar's pass through matplotlib's nonsingular(expander=0.1) leaves of the norm's limits
    (the figure has a colour bar, :457): an empty interval grows by a tenth of each end (to +-0.1 around zero),
    reversed limits come out as (0.9 * smaller, larger)."""
    if hi < lo:
        return hi - 0.1 * abs(hi), lo
    big = max(abs(lo), abs(hi))
    if big < 2.2250738585072014e-287:
        return -0.1, 0.1
    if hi - lo <= big * 1e-15:
        return lo - 0.1 * abs(lo), hi + 0.1 * abs(hi)
    return lo, hi


def normalise(mat: np.ndarray, use_log: bool, vmin: Optional[float], vmax: Optional[float]):
    """Returns (norm or NaN, vmin_used, vmax_used) as the figure's colour scale does.

    Linear (:420-423, :449-456): the mesh gets neither a norm nor limits, so vmin / vmax play no part and the
    scale is autoscaled to the finite min/max.  Log (:409-419): shifted = mat - min + 1e-6, LogNorm between the
    limits shifted alike; a limit that is not positive after the shift fails like LogNorm does.
    """
    mat = np.asarray(mat, np.float64)
    fin = np.isfinite(mat)
    if not fin.any():
        raise ValueError("matrix is all NaN")
    if use_log:
        mn = mat[fin].min()
        sh = mat - mn + 1e-6
        lo = (vmin - mn + 1e-6) if vmin is not None else sh[fin].min()
        hi = (vmax - mn + 1e-6) if vmax is not None else sh[fin].max()
        if not (lo > 0 and hi > 0 and math.isfinite(lo) and math.isfinite(hi)):
            raise ValueError("Invalid vmin or vmax")
        lo, hi = nonsingular_limits(float(lo), float(hi))
        with np.errstate(invalid="ignore", divide="ignore"):
            t = np.log10(sh)
        tlo, thi = math.log10(lo), math.log10(hi)
        nrm = (t - tlo) / (thi - tlo)
        return np.where(fin, nrm, np.nan), float(lo), float(hi)
    lo, hi = nonsingular_limits(float(mat[fin].min()), float(mat[fin].max()))
    nrm = (mat - lo) / (hi - lo)
    return np.where(fin, nrm, np.nan), lo, hi


def colour_cells(nrm: np.ndarray, lut: np.ndarray) -> np.ndarray:
    """norm -> RGBA8 per cell: LUT[clip(floor(norm*256), 0, 255)], NaN -> set_bad's (255,255,255,0) (:447).

    lut is uint8 [256, 3]: the colormap's 256 colours as the saved PNG shows them, round(c*255).
    """
    h, w = nrm.shape
    out = np.zeros((h, w, 4), np.uint8)
    fin = np.isfinite(nrm)
    x = np.where(fin, nrm, 0.0) * 256.0
    idx = np.clip(np.floor(x), 0, 255).astype(np.int64)
    out[..., :3] = lut[idx]
    out[..., 3] = 255
    out[~fin] = (255, 255, 255, 0)
    return out


def raster_index_maps(x_edges: np.ndarray, y_edges: np.ndarray, W: int, H: int):
    """Data-only tile geometry: pixel (px, py) -> (col, row) cell indices.

    The tile spans [x_edges[0], x_edges[-1]] x [y_edges[0], y_edges[-1]]; pixel
    centres are sampled; py = 0 is the top (largest AoD), like the figure's axes.
    """
    xe = np.asarray(x_edges, np.float64)
    ye = np.asarray(y_edges, np.float64)
    sx = (xe[-1] - xe[0]) / np.float64(W)
    sy = (ye[-1] - ye[0]) / np.float64(H)
    X = xe[0] + (np.arange(W, dtype=np.float64) + 0.5) * sx
    Y = ye[-1] - (np.arange(H, dtype=np.float64) + 0.5) * sy
    col = np.clip(np.searchsorted(xe, X, side="right") - 1, 0, len(xe) - 2)
    row = np.clip(np.searchsorted(ye, Y, side="right") - 1, 0, len(ye) - 2)
    return col.astype(np.int32), row.astype(np.int32)


def raster(cells_rgba: np.ndarray, x_edges, y_edges, W: int, H: int) -> np.ndarray:
    col, row = raster_index_maps(x_edges, y_edges, W, H)
    return cells_rgba[row[:, None], col[None, :]]


def render(matrix, aod_list, aoa_list, lut, use_log=False, vmin=None, vmax=None,
           blur_sigma=1.0, W=1024, H=1024):
    """generate_heatmap's data path end to end: blur -> scale -> cell colours -> tile."""
    mat = np.array(matrix, dtype=np.float64, copy=True)
    if blur_sigma and blur_sigma > 0:
        mat = blur_nan_aware(mat, blur_sigma)
    nrm, lo, hi = normalise(mat, use_log, vmin, vmax)
    cells = colour_cells(nrm, lut)
    tile = raster(cells, edges(aoa_list), edges(aod_list), W, H)
    return dict(blurred=mat, norm=nrm, vmin=lo, vmax=hi, cells=cells, tile=tile)


# ---------------------------------------------------------------------------
# beyond the three scripts: the dictionary correlation of the path estimators (heatmap_gemini_v4.py)
# ---------------------------------------------------------------------------
def gaussian_beam(x, center, width: float) -> np.ndarray:
    """LoSEstimator._gaussian_beam (heatmap_gemini_v4.py:175-180): exp(-(x - c)^2 / (2 sigma^2)), sigma = FWHM / 2.355."""
    sigma = width / 2.355
    return np.exp(-((np.asarray(x, np.float64) - np.asarray(center, np.float64)) ** 2) / (2 * sigma ** 2))


def los_dictionary(ue_angles, bs_angles, grid_res: float = 0.1, beam_width: float = 1.4):
    """LoSEstimator.construct_dictionary (:182-198): grids from the smallest to (short of) the largest angle in steps
    of grid_res; Phi_RX [n_ue, len(aoa_grid)], Phi_TX [n_bs, len(aod_grid)].  Returns (aoa_grid, aod_grid, Phi_RX, Phi_TX)."""
    ue, bs = np.asarray(ue_angles, np.float64), np.asarray(bs_angles, np.float64)
    aoa = np.arange(np.min(ue), np.max(ue), grid_res)
    aod = np.arange(np.min(bs), np.max(bs), grid_res)
    return aoa, aod, gaussian_beam(ue[:, None], aoa[None, :], beam_width), gaussian_beam(bs[:, None], aod[None, :], beam_width)


def nlos_dictionary(ue_angles, bs_angles, grid_res: float = 0.1, beam_width: float = 1.4):
    """NLoSEstimator.construct_dictionary (:283-296) with its own _gaussian_beam (:271-281): grids up to and including
    the largest angle (arange(min, max + grid_res, grid_res)); Phi[i, j] = beam centred on angle i seen at grid point j."""
    ue, bs = np.asarray(ue_angles, np.float64), np.asarray(bs_angles, np.float64)
    sigma = beam_width / 2.355
    aoa = np.arange(np.min(ue), np.max(ue) + grid_res, grid_res)
    aod = np.arange(np.min(bs), np.max(bs) + grid_res, grid_res)

    def beam(query, centers):
        q, c = np.meshgrid(query, centers)
        return np.exp(-((q - c) ** 2) / (2 * sigma ** 2))
    return aoa, aod, beam(aoa, ue), beam(aod, bs)


def dict_correlation(phi_rx, m, phi_tx) -> np.ndarray:
    """correlation = Phi_RX.T @ res_mat @ Phi_TX (heatmap_gemini_v4.py:209, :304), float64.  m [n_ue, n_bs] or a stack
    [F, n_ue, n_bs]."""
    a, b = np.asarray(phi_rx, np.float64), np.asarray(phi_tx, np.float64)
    return a.T @ np.asarray(m, np.float64) @ b


def dict_correlation_loops(phi_rx, m, phi_tx) -> np.ndarray:
    """The same sums written out element by element in index order (tiny cases only)."""
    a, b, mm = np.asarray(phi_rx, np.float64), np.asarray(phi_tx, np.float64), np.asarray(m, np.float64)
    out = np.zeros((a.shape[1], b.shape[1]))
    for i in range(a.shape[1]):
        for j in range(b.shape[1]):
            acc = 0.0
            for v in range(b.shape[0]):
                t = 0.0
                for u in range(a.shape[0]):
                    t += a[u, i] * mm[u, v]
                acc += t * b[v, j]
            out[i, j] = acc
    return out
