"""Batched, device-resident pipeline: hex text of F frames in HBM -> records, corrected rows, matrices, tiles.

This is the fast entry the reference has no counterpart for: its three scripts hand data to each
other through .xlsx files (serial_hex_to_excel_v3.py:155, bs_beam_correction.py:202-203,
excel_heatmap_v3.py:95); here the same per-frame chain runs back to back on the GPU through
slam_process_frames_dev.  torch is used only to own device memory and the stream.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, ops
from ._lib import Batch

STAGE_DECODE, STAGE_CORRECT, STAGE_RENDER, STAGE_ALL = 1, 2, 4, 7


def default_mapping() -> Dict[int, float]:
    """The reference's beam_angle.xlsx table (64 entries, BeamID -> degrees), shipped as JSON."""
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "beam_angle.json")
    with open(p) as fh:
        return {int(k): float(v) for k, v in json.load(fh).items()}


def viridis_lut() -> np.ndarray:
    """256 x 3 uint8 viridis as the reference's saved PNGs show it (round(c*255))."""
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "viridis_lut.npy")
    return np.load(p)


class FramePipeline:
    """Owns the device buffers for batches of up to `max_frames` frames of up to `max_text_len` text bytes."""

    def __init__(self, max_frames: int, max_text_len: int, device: int = 0, tile_w: int = 1024, tile_h: int = 1024,
                 mapping: Optional[Dict[int, float]] = None, lut: Optional[np.ndarray] = None,
                 blur_sigma: float = 1.0, use_log: bool = False, vmin: Optional[float] = None,
                 vmax: Optional[float] = None, cycle: int = ops.CYCLE, tol: int = ops.TOL,
                 mod_base: int = ops.MOD_BASE, src_is_bytes: bool = False, n_tile_slots: Optional[int] = None,
                 n_compact_slots: int = 1, reserve: bool = True, upscale: int = 1,
                 full_grid: Optional[bool] = None, fused: Optional[bool] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("FramePipeline needs a GPU (the HIP path has no CPU fallback)")
        self.dev = torch.device("cuda", device)
        # A ctx of its own: the library workspace this pipeline reserves is never re-pointed by other callers of the
        # process-wide ctx (ops.*, other pipelines), so graphs captured from run() stay valid as long as it lives.
        self.ctx = _lib.Context(device)
        if fused is not None:             # None: the library's default (or the environment's SLAMHIP_FUSED)
            self.ctx.set_fused(bool(fused))
        self.F = int(max_frames)
        self.max_text_len = int(max_text_len)
        self.row_tiles = int(self.ctx.lib.slam_row_tiles(self.max_text_len, 1 if src_is_bytes else 0))
        self.row_cap = ((self.row_tiles * _lib.TILE_SLOTS + 15) // 16) * 16
        self.tile_w, self.tile_h = int(tile_w), int(tile_h)
        # upscale > 1: the beam table interpolated to `upscale` times as many entries and the full grid rendered
        # (ops.upscaled_axis); 1: the reference's semantics, rows and columns are the angles that occur
        self.upscale = int(upscale)
        self.full_grid = (self.upscale > 1) if full_grid is None else bool(full_grid)
        b2a, angles = ops.upscaled_axis(mapping if mapping is not None else default_mapping(), self.upscale)
        self.n_axis = len(angles)
        self.cells_per_frame = _lib.cells_per_frame(self.n_axis)
        self.axis_angles = angles
        F, rc, d = self.F, self.row_cap, self.dev
        u8, u32 = torch.uint8, torch.int32
        self.b2a = torch.from_numpy(b2a.astype(np.int16)).to(d)
        self.angles = torch.from_numpy(angles.copy()).to(d)
        self.lut = torch.from_numpy(np.ascontiguousarray(lut if lut is not None else viridis_lut(), np.uint8).copy()).to(d)
        self.flag = torch.empty(F * rc, dtype=u8, device=d)
        self.ue = torch.empty(F * rc, dtype=u8, device=d)
        self.bs = torch.empty(F * rc, dtype=u8, device=d)
        self.rss = torch.empty(F * rc, dtype=u32, device=d)
        self.clk = torch.empty(F * rc, dtype=u32, device=d)
        self.tile_cnt = torch.zeros(F * self.row_tiles, dtype=u32, device=d)
        self.kept_blocks = int(self.ctx.lib.slam_kept_blocks(rc, self.max_text_len, 1 if src_is_bytes else 0))
        self.kept_cnt = torch.zeros(F * self.kept_blocks, dtype=u32, device=d)
        self.k_ue = torch.empty(F * rc, dtype=u8, device=d)
        self.k_bs = torch.empty(F * rc, dtype=u8, device=d)
        self.k_rss = torch.empty(F * rc, dtype=u32, device=d)
        self.k_clk = torch.empty(F * rc, dtype=u32, device=d)
        # What a tile is painted from (cell colours, pixel -> cell maps) and the frame's stats words live in one
        # flat buffer per slot, so a rank can ship a batch's results in one 20 KB-per-frame message.
        self.compact_layout = compact_layout(F, self.tile_w, self.tile_h, self.cells_per_frame)
        self.compact = torch.zeros((int(n_compact_slots), self.compact_layout["bytes"]), dtype=u8, device=d)
        self.compact_slot = 0
        self.matrix = torch.empty(F * self.n_axis * self.n_axis, dtype=torch.float64, device=d)
        self.blurred = torch.empty(F * self.n_axis * self.n_axis, dtype=torch.float64, device=d)
        self.n_tile_slots = int(n_tile_slots or F)
        self.tiles = torch.empty((self.n_tile_slots, self.tile_h, self.tile_w, 4), dtype=u8, device=d)
        self.off = torch.zeros(F, dtype=torch.int64, device=d)
        self.len = torch.zeros(F, dtype=u32, device=d)
        nan = float("nan")
        bt = Batch()
        bt.n_frames = F
        bt.max_text_len = self.max_text_len
        bt.src_is_bytes = 1 if src_is_bytes else 0
        bt.cycle, bt.tol, bt.mod_base = int(cycle), int(tol), int(mod_base)
        bt.beam_to_axis_dev = self.b2a.data_ptr()
        bt.axis_angle_dev = self.angles.data_ptr()
        bt.n_axis = self.n_axis
        bt.full_grid = 1 if self.full_grid else 0
        bt.blur_sigma = float(blur_sigma)
        bt.use_log = 1 if use_log else 0
        bt.vmin = nan if vmin is None else float(vmin)
        bt.vmax = nan if vmax is None else float(vmax)
        bt.lut_dev = self.lut.data_ptr()
        bt.tile_w, bt.tile_h = self.tile_w, self.tile_h
        bt.row_cap = rc
        bt.flag_dev, bt.ue_dev, bt.bs_dev = self.flag.data_ptr(), self.ue.data_ptr(), self.bs.data_ptr()
        bt.rss_dev, bt.clk_dev = self.rss.data_ptr(), self.clk.data_ptr()
        bt.tile_cnt_dev = self.tile_cnt.data_ptr()
        bt.kept_ue_dev, bt.kept_bs_dev = self.k_ue.data_ptr(), self.k_bs.data_ptr()
        bt.kept_rss_dev, bt.kept_clk_dev = self.k_rss.data_ptr(), self.k_clk.data_ptr()
        bt.kept_cnt_dev = self.kept_cnt.data_ptr()
        self._point_compact(bt, 0)
        bt.matrix_dev = self.matrix.data_ptr()
        bt.blurred_dev = self.blurred.data_ptr()
        bt.tile_dev = self.tiles.data_ptr()
        bt.text_off_dev = self.off.data_ptr()
        bt.text_len_dev = self.len.data_ptr()
        self.bt = bt
        self._text_keep = None
        # reserve the library workspace once, so steady-state calls allocate nothing (reserve=False: a pipeline that
        # only ever runs on the library's fixed scratch, run(..., noalloc=True))
        bt.text_dev = self.flag.data_ptr()          # any non-null pointer: only the shape matters here
        if reserve:
            self.ctx.check(self.ctx.lib.slam_reserve_workspace(self.ctx.h, C.byref(bt)))

    def _point_compact(self, bt, slot: int) -> None:
        base, lay = self.compact[slot].data_ptr(), self.compact_layout
        bt.cells_dev, bt.colmap_dev = base + lay["cells"], base + lay["colmap"]
        bt.rowmap_dev, bt.stats_dev = base + lay["rowmap"], base + lay["stats"]
        self.compact_slot = slot

    @property
    def stats(self) -> torch.Tensor:
        """int32 view of the current slot's stats words [F * STAT_WORDS]."""
        lay = self.compact_layout
        return self.compact[self.compact_slot][lay["stats"]:lay["stats"] + self.F * _lib.STAT_WORDS * 4].view(torch.int32)

    def set_uniform_layout(self, n_frames: int, frame_stride: int, text_len: int) -> None:
        """Frames of equal length laid out at a fixed stride in the text buffer."""
        if n_frames > self.F or text_len > self.max_text_len:
            raise ValueError("batch exceeds the pipeline's capacity")
        self.off[:n_frames] = torch.arange(n_frames, dtype=torch.int64, device=self.dev) * int(frame_stride)
        self.len[:n_frames] = int(text_len)
        self.bt.n_frames = int(n_frames)

    def set_layout(self, offsets: Sequence[int], lengths: Sequence[int]) -> None:
        n = len(offsets)
        if n > self.F or (n and max(lengths) > self.max_text_len):
            raise ValueError("batch exceeds the pipeline's capacity")
        self.off[:n] = torch.tensor(list(offsets), dtype=torch.int64, device=self.dev)
        self.len[:n] = torch.tensor(list(lengths), dtype=torch.int64, device=self.dev).to(torch.int32)
        self.bt.n_frames = n

    def run(self, text: torch.Tensor, text_base_offset: int = 0, stages: int = STAGE_ALL, tile_slot: int = 0,
            compact_slot: Optional[int] = None, noalloc: bool = False) -> None:
        """Enqueue one batch on torch's current stream (no synchronisation).  noalloc: through
        slam_process_frames_noalloc_dev, on the library's fixed device scratch instead of the ctx workspace."""
        self.ctx.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        if compact_slot is not None and compact_slot != self.compact_slot:
            self._point_compact(self.bt, int(compact_slot))
        self._text_keep = text
        self.bt.text_dev = text.data_ptr() + int(text_base_offset)
        self.bt.tile_dev = self.tiles.data_ptr() + int(tile_slot) * self.tile_h * self.tile_w * 4
        call = self.ctx.lib.slam_process_frames_noalloc_dev if noalloc else self.ctx.lib.slam_process_frames_dev
        self.ctx.check(call(self.ctx.h, C.byref(self.bt), int(stages)))

    # ---- read-back helpers (synchronise) -----------------------------------------------------
    def stats_host(self) -> np.ndarray:
        torch.cuda.synchronize(self.dev)
        return self.stats.cpu().numpy().view(np.uint32).reshape(self.F, _lib.STAT_WORDS)[: self.bt.n_frames]

    def frame_rows(self, f: int, st: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
        """Decoded records of frame f in order (the used entries of its tiles, concatenated)."""
        st = self.stats_host() if st is None else st
        rc, T, S = self.row_cap, self.row_tiles, _lib.TILE_SLOTS
        span = int(st[f, _lib.STAT_ROW_SLOTS])
        cnt = self.tile_cnt[f * T:f * T + span // S].to(torch.int64)
        use = (torch.arange(S, device=self.dev)[None, :] < cnt[:, None]).reshape(-1)
        sl = slice(f * rc, f * rc + span)
        out = dict(flag=self.flag[sl][use].cpu().numpy(), ue=self.ue[sl][use].cpu().numpy(),
                   bs=self.bs[sl][use].cpu().numpy(), rss=self.rss[sl][use].cpu().numpy().view(np.uint32),
                   clk=self.clk[sl][use].cpu().numpy().view(np.uint32))
        if len(out["flag"]) != int(st[f, _lib.STAT_VALID]):
            raise RuntimeError("tile counts and the frame's record count disagree")
        return out

    def frame_kept(self, f: int, st: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
        """Corrected rows of frame f in order (the used entries of its blocks, concatenated)."""
        st = self.stats_host() if st is None else st
        rc, KB, S = self.row_cap, self.kept_blocks, _lib.kept_unit(st[f, _lib.STAT_LAYOUT])
        span = int(st[f, _lib.STAT_ROW_SLOTS])
        nb = (span + S - 1) // S
        cnt = self.kept_cnt[f * KB:f * KB + nb].to(torch.int64)
        use = (torch.arange(S, device=self.dev)[None, :] < cnt[:, None]).reshape(-1)[:min(nb * S, rc)]
        sl = slice(f * rc, f * rc + len(use))
        out = dict(ue=self.k_ue[sl][use].cpu().numpy(), bs=self.k_bs[sl][use].cpu().numpy(),
                   rss=self.k_rss[sl][use].cpu().numpy().view(np.uint32),
                   clk=self.k_clk[sl][use].cpu().numpy().view(np.uint32))
        if len(out["ue"]) != int(st[f, _lib.STAT_KEPT]):
            raise RuntimeError("block counts and the frame's kept count disagree")
        return out

    def frame_matrix(self, f: int, st: Optional[np.ndarray] = None):
        """(full-grid mean [n_axis,n_axis], compact blurred [n_aod,n_aoa], aod angles, aoa angles, vmin, vmax)."""
        st = self.stats_host() if st is None else st
        na = self.n_axis
        full = self.matrix[f * na * na:(f + 1) * na * na].cpu().numpy().reshape(na, na)
        h, w = int(st[f, _lib.STAT_N_AOD]), int(st[f, _lib.STAT_N_AOA])
        bl = self.blurred[f * na * na: f * na * na + h * w].cpu().numpy().reshape(h, w)
        if self.full_grid:            # every entry of the axis table is a row and a column
            cnt_rows = cnt_cols = np.arange(na)
        else:
            cnt_rows = np.flatnonzero(np.isfinite(full).any(axis=1))
            cnt_cols = np.flatnonzero(np.isfinite(full).any(axis=0))
        vv = st[f, _lib.STAT_VMIN_LO:_lib.STAT_VMAX_HI + 1].copy().view(np.float64)
        return full, bl, self.axis_angles[cnt_rows], self.axis_angles[cnt_cols], float(vv[0]), float(vv[1])

    def frame_tile(self, f: int, tile_slot: int = 0) -> np.ndarray:
        torch.cuda.synchronize(self.dev)
        return self.tiles[tile_slot + f].cpu().numpy()


def compact_layout(F: int, W: int, H: int, cells_per_frame: int = 4096) -> Dict[str, int]:
    """Byte offsets of the per-batch result buffer: cells [F*cells_per_frame*4] | colmap [F*2W] | rowmap [F*2H] |
    stats [F*64]."""
    al = lambda x: (x + 255) // 256 * 256
    cells = 0
    colmap = al(cells + F * cells_per_frame * 4)
    rowmap = al(colmap + F * W * 2)
    stats = al(rowmap + F * H * 2)
    return dict(cells=cells, colmap=colmap, rowmap=rowmap, stats=stats, bytes=al(stats + F * _lib.STAT_WORDS * 4))


def expand_compact(buf: torch.Tensor, F: int, W: int, H: int, out: Optional[torch.Tensor] = None,
                   device: Optional[int] = None, cells_per_frame: int = 4096) -> torch.Tensor:
    """Paint the F tiles a compact result buffer (FramePipeline.compact[slot], possibly received from another rank)
    describes: [F, H, W, 4] uint8, bit-identical to the tiles the producing pipeline rendered."""
    dev = buf.device
    c = _lib.ctx(dev.index if device is None else device)
    lay = compact_layout(F, W, H, cells_per_frame)
    if out is None:
        out = torch.empty((F, H, W, 4), dtype=torch.uint8, device=dev)
    c.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    base = buf.data_ptr()
    c.check(c.lib.slam_raster_dev(c.h, C.c_void_p(base + lay["cells"]), C.c_void_p(base + lay["colmap"]),
                                  C.c_void_p(base + lay["rowmap"]), C.c_void_p(base + lay["stats"]), int(F), int(W),
                                  int(H), C.c_void_p(out.data_ptr()), int(cells_per_frame)))
    return out


def synth_frames(n_frames: int, B: int, N: int, first_frame: int = 0, seed: int = 20260126, device: int = 0,
                 out: Optional[torch.Tensor] = None, frame_stride: Optional[int] = None) -> torch.Tensor:
    """Generate synthetic frames on the device (uint8 tensor [n_frames * frame_stride])."""
    c = _lib.ctx(device)
    tl = int(c.lib.slam_synth_text_len(B, N))
    stride = int(frame_stride or ((tl + 255) // 256) * 256)
    dev = torch.device("cuda", device)
    if out is None:
        out = torch.empty(n_frames * stride + 256, dtype=torch.uint8, device=dev)
    c.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    c.check(c.lib.slam_synth_frames_dev(c.h, C.c_void_p(out.data_ptr()), stride, int(first_frame), int(n_frames),
                                        int(B), int(N), int(seed)))
    return out
