"""Whole captures in, per-capture results out: the three reference scripts' chain without workbooks.

`process_texts` packs any number of capture texts (different lengths) into one device buffer and runs
slam_process_frames_dev once.  Texts that are not in the strict capture format ("XX XX ..." with an
optional leading "« ") are unpacked by the exact general tokenizer (slam_hex_unpack) and re-run as byte
frames.  `process_files_sharded` does the same for the files one rank owns and gathers the tiles.
"""
from __future__ import annotations

import os
import threading
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, ops
from .pipeline import FramePipeline, STAGE_ALL


# Buffers and pipelines of the default configuration are kept between calls (a pipeline owns a few hundred MB of device
# arrays and a library workspace: building one per call costs more than processing nineteen captures).  Entries are
# keyed by the calling thread as well, so two threads never share a staging buffer or a pipeline; the least recently
# used entry goes first; clear_cache() releases everything.  process_texts itself is re-entrant across threads, not
# within one (a call reuses its thread's buffers).
_CACHE: "OrderedDict[tuple, object]" = OrderedDict()
_CACHE_MAX = 6
_CACHE_LOCK = threading.Lock()


def _cached(key, make):
    key = (threading.get_ident(), os.environ.get("SLAMHIP_FUSED", "")) + tuple(key)
    with _CACHE_LOCK:
        v = _CACHE.get(key)
        if v is not None:
            _CACHE.move_to_end(key)
            return v
        while len(_CACHE) >= _CACHE_MAX:
            _CACHE.popitem(last=False)
    v = make()
    with _CACHE_LOCK:
        _CACHE[key] = v
    return v


def clear_cache() -> None:
    """Drop the staging buffers and pipelines kept between calls (their device memory goes back to torch's allocator
    and the pipelines' library workspaces are freed)."""
    with _CACHE_LOCK:
        _CACHE.clear()


def _pack(chunks: Sequence[np.ndarray], device) -> tuple:
    """The chunks at 256-byte aligned offsets of one device buffer (staged through pinned host memory)."""
    offs, lens, pos = [], [], 0
    for c in chunks:
        offs.append(pos)
        lens.append(len(c))
        pos += (len(c) + 255) // 256 * 256
    cap = max(1 << 20, 1 << (pos + 256 - 1).bit_length())                 # buffers by power-of-two size class
    host = _cached(("pin", cap), lambda: torch.empty(cap, dtype=torch.uint8, pin_memory=True))
    dev = _cached(("dev", str(device), cap), lambda: torch.empty(cap, dtype=torch.uint8, device=device))
    hv = host.numpy()
    for o, c in zip(offs, chunks):
        hv[o:o + len(c)] = c
        hv[o + len(c):(o + len(c) + 255) // 256 * 256 + 32] = 0           # the slack the kernels may read past a frame
    dev[:pos + 256].copy_(host[:pos + 256], non_blocking=True)
    return dev, offs, lens


def _pipeline(n: int, max_len: int, device: int, tile: int, src_is_bytes: bool, kwargs: dict) -> FramePipeline:
    """A pipeline that holds n frames of up to max_len bytes; with default settings, one kept from an earlier call
    whose capacity suffices."""
    if kwargs:
        return FramePipeline(n, max(max_len, 1), device=device, tile_w=tile, tile_h=tile, src_is_bytes=src_is_bytes, **kwargs)
    n_cap = max(8, 1 << (n - 1).bit_length())
    len_cap = max(1 << 20, 1 << (max(max_len, 1) - 1).bit_length())
    return _cached(("pipe", device, n_cap, len_cap, tile, src_is_bytes),
                   lambda: FramePipeline(n_cap, len_cap, device=device, tile_w=tile, tile_h=tile, src_is_bytes=src_is_bytes))


def _collect(pipe: FramePipeline, n: int, rows: bool, tiles: str) -> List[Dict]:
    st = pipe.stats_host()
    out = []
    for f in range(n):
        full, bl, aod, aoa, lo, hi = pipe.frame_matrix(f, st)
        r = dict(stats=st[f].copy(), matrix=full, blurred=bl, aod=aod, aoa=aoa, vmin=lo, vmax=hi)
        if rows:                    # the row arrays are the bulk of a capture's results: fetched only when asked for
            r.update(rows=pipe.frame_rows(f, st), kept=pipe.frame_kept(f, st))
        if tiles == "host":
            r["tile"] = pipe.frame_tile(f)
        else:                       # stays on the device (e.g. for a gather); a copy: the pipeline's buffer is reused
            r["tile_dev"] = pipe.tiles[f].clone()
        out.append(r)
    return out


def process_texts(texts: Sequence[bytes], device: int = 0, tile: int = 1024, rows: bool = True, tiles: str = "host",
                  **pipeline_kwargs) -> List[Dict]:
    """Every text through decode -> correct -> pivot -> blur -> raster on the GPU; one dict per text: the stats
    words, the matrices and their axes, the tile ("tile": host array, or with tiles="device" "tile_dev": device
    tensor), and with rows=True the decoded records and the corrected rows."""
    if tiles not in ("host", "device"):
        raise ValueError("tiles must be 'host' or 'device'")
    if not texts:
        return []
    dev = torch.device("cuda", device)
    arrs = [np.frombuffer(bytes(t), np.uint8) for t in texts]
    buf, offs, lens = _pack(arrs, dev)
    pipe = _pipeline(len(arrs), max(lens), device, tile, False, pipeline_kwargs)
    pipe.set_layout(offs, lens)
    pipe.run(buf, stages=STAGE_ALL)
    res = _collect(pipe, len(arrs), rows, tiles)
    loose = [i for i, r in enumerate(res) if r["stats"][_lib.STAT_NOT_STRICT]]
    if loose:
        data = [ops.hex_unpack(arrs[i], device=device) for i in loose]
        bbuf, boffs, blens = _pack(data, dev)
        bpipe = _pipeline(len(data), max(blens), device, tile, True, pipeline_kwargs)
        bpipe.set_layout(boffs, blens)
        bpipe.run(bbuf, stages=STAGE_ALL)
        for i, r in zip(loose, _collect(bpipe, len(data), rows, tiles)):
            res[i] = r
    return res


def process_files_sharded(paths: Sequence[str], device: int = 0, tile: int = 1024, gather_via_host: bool = False,
                          rows: bool = False, timing: Optional[Dict] = None, **pipeline_kwargs):
    """Under torch.distributed: each rank processes the files `balance_files` gives it; rank 0 receives every
    file's tile (list indexed like `paths`), other ranks None.  Without an initialised process group: all files.
    `gather_via_host` stages the gather through host memory (for a gloo process group); otherwise the tiles stay
    on the device from the raster kernel through the gather.  rows=True also fetches every log's decoded and
    corrected rows.  `timing`, if given, receives this rank's seconds for reading, processing and gathering."""
    import time
    import os
    import torch.distributed as dist
    from . import dist as D
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    sizes = [os.path.getsize(p) for p in paths]
    bins = D.balance_files(sizes, world)
    mine = bins[rank]
    t0 = time.perf_counter()
    texts = [open(paths[i], "rb").read() for i in mine]
    t1 = time.perf_counter()
    res = process_texts(texts, device=device, tile=tile, rows=rows, tiles="device", **pipeline_kwargs)
    torch.cuda.synchronize(device)
    t2 = time.perf_counter()
    if timing is not None:
        timing.update(read_s=t1 - t0, process_s=t2 - t1, text_bytes=sum(len(t) for t in texts), logs=len(texts))
    if world == 1:
        out1 = [r["tile_dev"].cpu().numpy() for r in res]
        if timing is not None:
            timing["gather_s"] = time.perf_counter() - t2
        return out1, res
    per = max(len(b) for b in bins)
    dev = torch.device("cuda", device)
    tiles = torch.zeros((per, tile, tile, 4), dtype=torch.uint8, device=dev)
    for j, r in enumerate(res):
        tiles[j].copy_(r["tile_dev"])               # device to device
    got = D.gather_tiles(tiles.cpu() if gather_via_host else tiles, dst=0)
    if timing is not None:
        torch.cuda.synchronize(device)
        timing["gather_s"] = time.perf_counter() - t2
    if rank != 0:
        return None, res
    out: List[Optional[np.ndarray]] = [None] * len(paths)
    for r, b in enumerate(bins):
        for j, i in enumerate(b):
            out[i] = got[r][j].cpu().numpy()
    return out, res
