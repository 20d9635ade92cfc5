"""ctypes binding of libslamhip.so (include/slamhip.h).

There is no CPU fallback: if the library is missing or no GPU is present the
calls raise.  Importing this module never touches the GPU; `ctx()` does.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLAMHIP_LIB") or os.path.join(_HERE, "libslamhip.so")     # override: tuning builds kept beside the shipped one

OK, E_INVALID, E_HIP, E_CAPACITY, E_EMPTY, E_ALLNAN, E_RANGE, E_LIMITS = 0, -1, -2, -3, -4, -5, -6, -7
STAT_WORDS = 16
(STAT_NBYTES, STAT_NOT_STRICT, STAT_VALID, STAT_DISCARDED, STAT_GROUPS, STAT_BASELINES, STAT_KEPT,
 STAT_N_AOD, STAT_N_AOA, STAT_ALLNAN, STAT_VMIN_LO, STAT_VMIN_HI, STAT_VMAX_LO, STAT_VMAX_HI,
 STAT_ROW_SLOTS, STAT_LAYOUT) = range(16)
LAYOUT_TWO_PASS, LAYOUT_KEPT_TILES, LAYOUT_FUSED_WHY_SHIFT = 1, 2, 8
# SLAM_FUSED_GAVE_UP_*: why the fused decode + correct kernel handed a frame on (layout word >> LAYOUT_FUSED_WHY_SHIFT)
(FUSED_GAVE_UP_TILE_LISTS, FUSED_GAVE_UP_UNSYNC, FUSED_GAVE_UP_FIRST_ROW, FUSED_GAVE_UP_WG_LISTS, FUSED_GAVE_UP_TIMEOUT,
 FUSED_GAVE_UP_NEIGHBOUR, FUSED_GAVE_UP_GROUP_SIZE) = 1, 2, 4, 8, 16, 32, 64
TILE_BYTES, TILE_SLOTS, KEPT_BLOCK = 3024, 276, 1024
ABI_REVISION = 4                      # SLAM_ABI_REVISION of the header this binding was written against


def kept_unit(layout_word: int) -> int:
    """Entries per block of a frame's corrected-row arrays, from its SLAM_STAT_LAYOUT word."""
    return TILE_SLOTS if int(layout_word) & LAYOUT_KEPT_TILES else KEPT_BLOCK
NOALLOC_SCRATCH_BYTES = 64 << 20
MAX_AXIS = 256


def cells_per_frame(n_axis: int) -> int:
    """slam_cells_per_frame: cells reserved per frame in the cell-colour buffers."""
    return 4096 if n_axis <= 64 else n_axis * n_axis

EXPORTS = [
    "slam_ctx_create", "slam_ctx_destroy", "slam_set_stream", "slam_use_own_stream", "slam_sync", "slam_last_error", "slam_version", "slam_abi_revision", "slam_set_fused", "slam_get_fused",
    "slam_hex_unpack", "slam_token_rejects", "slam_parse_records", "slam_parse_rejects", "slam_parse_text", "slam_correct", "slam_pivot_mean", "slam_pivot_mean_f64",
    "slam_blur_nan", "slam_render", "slam_dict_correlate", "slam_dict_correlate_dev", "slam_process_frames_dev", "slam_process_frames_noalloc_dev", "slam_raster_dev", "slam_reserve_workspace", "slam_pin_workspace", "slam_batch_sizeof",
    "slam_row_tiles", "slam_kept_blocks", "slam_cells_per_frame", "slam_synth_text_len", "slam_synth_frames_dev", "slam_copy_probe_dev", "slam_read_probe_dev",
    "slam_timing_enable", "slam_timing_select", "slam_timing_reset", "slam_timing_read",
]


class SlamError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libslamhip error {code}: {msg}")
        self.code = code
        self.msg = msg


class Batch(C.Structure):
    """struct slam_batch"""
    _fields_ = [
        ("text_dev", C.c_void_p), ("text_off_dev", C.c_void_p), ("text_len_dev", C.c_void_p),
        ("n_frames", C.c_uint32), ("max_text_len", C.c_uint32), ("src_is_bytes", C.c_uint32),
        ("cycle", C.c_int32), ("tol", C.c_int32), ("mod_base", C.c_int32),
        ("beam_to_axis_dev", C.c_void_p), ("axis_angle_dev", C.c_void_p), ("n_axis", C.c_int32), ("full_grid", C.c_int32),
        ("blur_sigma", C.c_double), ("use_log", C.c_int32), ("vmin", C.c_double), ("vmax", C.c_double),
        ("lut_dev", C.c_void_p), ("tile_w", C.c_int32), ("tile_h", C.c_int32),
        ("row_cap", C.c_uint32),
        ("flag_dev", C.c_void_p), ("ue_dev", C.c_void_p), ("bs_dev", C.c_void_p),
        ("rss_dev", C.c_void_p), ("clk_dev", C.c_void_p), ("tile_cnt_dev", C.c_void_p),
        ("kept_ue_dev", C.c_void_p), ("kept_bs_dev", C.c_void_p),
        ("kept_rss_dev", C.c_void_p), ("kept_clk_dev", C.c_void_p), ("kept_cnt_dev", C.c_void_p),
        ("stats_dev", C.c_void_p), ("matrix_dev", C.c_void_p), ("blurred_dev", C.c_void_p),
        ("tile_dev", C.c_void_p), ("cells_dev", C.c_void_p), ("colmap_dev", C.c_void_p), ("rowmap_dev", C.c_void_p),
    ]


_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load the shared library and declare every prototype (no GPU call)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python slam-process_amd/slamhip/build.py` "
                "(there is no CPU fallback for this path)")
        # A process must hold ONE HIP runtime.  PyTorch-ROCm ships its own libamdhip64.so.7; loading it
        # first makes the dynamic linker satisfy this library's libamdhip64.so.7 dependency with the same
        # copy, so torch tensors and these kernels share one runtime, one device context, one stream space.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        vp, sz, i32, u32, u64, f64 = C.c_void_p, C.c_size_t, C.c_int32, C.c_uint32, C.c_uint64, C.c_double
        psz = C.POINTER(C.c_size_t)
        lib.slam_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        lib.slam_ctx_destroy.argtypes = [vp]
        lib.slam_set_stream.argtypes = [vp, vp]
        lib.slam_use_own_stream.argtypes = [vp]
        lib.slam_sync.argtypes = [vp]
        lib.slam_last_error.argtypes = [vp]
        lib.slam_last_error.restype = C.c_char_p
        lib.slam_version.restype = C.c_char_p
        if not hasattr(lib, "slam_abi_revision") or lib.slam_abi_revision() != ABI_REVISION:
            raise RuntimeError(f"{LIB_PATH}: the library's binary interface is not revision {ABI_REVISION}, the one this "
                               "binding was written against -- rebuild the library or update the binding")
        lib.slam_set_fused.argtypes = [vp, C.c_int]
        lib.slam_get_fused.argtypes = [vp]
        lib.slam_hex_unpack.argtypes = [vp, vp, sz, vp, sz, psz]
        lib.slam_parse_records.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp, sz, psz, psz]
        lib.slam_parse_text.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp, sz, psz, psz]
        lib.slam_parse_rejects.argtypes = [vp, vp, sz, vp, sz, psz]
        lib.slam_token_rejects.argtypes = [vp, vp, sz, vp, vp, vp, sz, psz]
        lib.slam_correct.argtypes = [vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, u32, vp, vp, C.c_int64,
                                     vp, vp, vp, vp, psz, vp, vp, vp, vp, vp, psz, psz]
        lib.slam_pivot_mean.argtypes = [vp, vp, vp, vp, sz, vp, C.c_int, vp, vp, vp]
        lib.slam_pivot_mean_f64.argtypes = [vp, vp, vp, vp, sz, vp, C.c_int, vp, vp, vp]
        lib.slam_blur_nan.argtypes = [vp, vp, C.c_int, C.c_int, f64, vp]
        lib.slam_render.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, f64, f64, vp, vp, vp,
                                    C.c_int, C.c_int, vp]
        lib.slam_process_frames_dev.argtypes = [vp, C.POINTER(Batch), u32]
        lib.slam_process_frames_noalloc_dev.argtypes = [vp, C.POINTER(Batch), u32]
        lib.slam_reserve_workspace.argtypes = [vp, C.POINTER(Batch)]
        lib.slam_pin_workspace.argtypes = [vp, C.c_int]
        lib.slam_batch_sizeof.restype = sz
        if lib.slam_batch_sizeof() != C.sizeof(Batch):
            raise RuntimeError(f"{LIB_PATH}: slam_batch is {lib.slam_batch_sizeof()} bytes in the library, "
                               f"{C.sizeof(Batch)} in this binding -- rebuild the library or update the binding")
        lib.slam_raster_dev.argtypes = [vp, vp, vp, vp, vp, u32, i32, i32, vp, u32]
        lib.slam_cells_per_frame.argtypes = [i32]
        lib.slam_cells_per_frame.restype = u32
        lib.slam_synth_text_len.argtypes = [u32, u32]
        lib.slam_synth_text_len.restype = sz
        lib.slam_synth_frames_dev.argtypes = [vp, vp, u64, u32, u32, u32, u32, u64]
        lib.slam_copy_probe_dev.argtypes = [vp, vp, vp, sz]
        lib.slam_read_probe_dev.argtypes = [vp, vp, vp, sz, C.c_int]
        lib.slam_dict_correlate.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, u32, vp]
        lib.slam_dict_correlate_dev.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, u32, vp]
        lib.slam_row_tiles.argtypes = [u32, u32]
        lib.slam_row_tiles.restype = u32
        lib.slam_kept_blocks.argtypes = [u32, u32, u32]
        lib.slam_kept_blocks.restype = u32
        lib.slam_timing_enable.argtypes = [vp, C.c_int]
        lib.slam_timing_select.argtypes = [vp, C.c_char_p]
        lib.slam_timing_reset.argtypes = [vp]
        lib.slam_timing_read.argtypes = [vp, vp, vp, vp, C.c_int]
        for name in EXPORTS:
            fn = getattr(lib, name)
            if name not in ("slam_last_error", "slam_version", "slam_synth_text_len", "slam_row_tiles", "slam_kept_blocks", "slam_cells_per_frame"):
                fn.restype = C.c_int
        _lib = lib
        return lib


class Context:
    """A slam_ctx bound to one device."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.slam_ctx_create(int(device), C.byref(h))
        if rc != OK or not h:
            raise SlamError(rc, f"slam_ctx_create(device={device}) failed: no usable GPU (HIP path has no CPU fallback)")
        self.h = h
        self.device = device

    def check(self, rc: int) -> int:
        if rc < 0:
            raise SlamError(rc, (self.lib.slam_last_error(self.h) or b"").decode("utf-8", "replace"))
        return rc

    def set_stream(self, stream_ptr: Optional[int]) -> None:
        """Enqueue on this hipStream_t handle from now on (0 is HIP's null stream = torch's default stream);
        None goes back to the ctx's own stream."""
        if stream_ptr is None:
            self.check(self.lib.slam_use_own_stream(self.h))
        else:
            self.check(self.lib.slam_set_stream(self.h, C.c_void_p(int(stream_ptr))))

    def sync(self) -> None:
        self.check(self.lib.slam_sync(self.h))

    def set_fused(self, on: bool) -> None:
        """Batched calls that decode and correct: one fused kernel (True) or two (False); see slam_set_fused."""
        self.check(self.lib.slam_set_fused(self.h, 1 if on else 0))

    def get_fused(self) -> bool:
        return bool(self.check(self.lib.slam_get_fused(self.h)))

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.slam_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- timing ------------------------------------------------------------------------------
    def timing(self, on: bool) -> None:
        self.check(self.lib.slam_timing_enable(self.h, 1 if on else 0))

    def timing_select(self, names=None) -> None:
        """Time only these kernels (iterable of names as timing_read reports them); None: every kernel."""
        self.check(self.lib.slam_timing_select(self.h, ",".join(names).encode() if names else None))

    def timing_reset(self) -> None:
        self.check(self.lib.slam_timing_reset(self.h))

    def timing_read(self):
        cap = 64
        names = C.create_string_buffer(32 * cap)
        launches = (C.c_uint32 * cap)()
        ms = (C.c_float * cap)()
        n = self.check(self.lib.slam_timing_read(self.h, names, launches, ms, cap))
        out = {}
        for i in range(min(n, cap)):
            nm = names.raw[32 * i:32 * i + 32].split(b"\0", 1)[0].decode()
            out[nm] = (int(launches[i]), float(ms[i]))
        return out


_ctxs = {}


def ctx(device: int = 0) -> Context:
    """Process-wide context per device (created on first use)."""
    c = _ctxs.get(device)
    if c is None:
        c = Context(device)
        _ctxs[device] = c
    return c


def ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
