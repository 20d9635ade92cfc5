#!/usr/bin/env python3
"""Throughput benchmark of the decode -> correct -> raster hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one batch of F synthetic 256 x 4096 frames (the shape the headline metric is quoted on) through
slam_process_frames_dev: hex text resident in HBM -> decoded records -> corrected rows -> mean matrix ->
blur -> 1024 x 1024 RGBA tile per frame.  With N > 1 (launched by torch.distributed.run, one rank per
GPU) every rank processes its own frames (weak scaling) and every step's results are gathered to rank 0
over RCCL on a side stream while the next step computes.  --gather compact (default) ships what each tile
is painted from (cell colours + pixel->cell maps + stats, 20.6 KB per 4 MiB tile; slam_raster_dev repaints it
bit for bit on the receiver, checked after the timed region against one full-tile gather); --gather tiles
ships the 4 MiB tiles themselves, which at this frame rate is bound by rank 0's xGMI ingest, not by compute.
With N > 1 the other gather mode is timed too, after the headline region, and both rates are reported.

Rank 0 prints ONE JSON line.  `value` is frames/s over all ranks with the text already in HBM when
the clock starts.  `roofline` prices the dominant kernel (algorithmic bytes / its mean duration from
HIP events on the launch stream) against the 8 TB/s HBM peak; `cpu_baseline` is the CPU oracle
(oracle/, a sequential C + NumPy restatement of the reference) timed here on a bounded sample, one
single-threaded worker process per host core of the box's share (16).  Two checks ride along after the clock:
`cpu_baseline.parity_check` holds the device's results for frame 0 against that oracle's for the same frame, and
`roofline.measured_ceilings` gives what plain read, copy and 3:1 read/write streams reach on the same device.
At N = 1, `other_configs` carries short passes over the two other synthetic shapes the benchmark is specified for
(256 x 16384 long sweeps; 1024 x 8192 with the 4x up-scaled beam table), run after the headline region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "slam-process_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--frames-per-step", type=int, default=1600,
                   help="frames per batch (one slam_process_frames_dev call); larger batches amortise launch tails")
    p.add_argument("--beams", type=int, default=256, help="rows (UE sweeps) per frame")
    p.add_argument("--samples", type=int, default=4096, help="records per row")
    p.add_argument("--tile", type=int, default=1024)
    p.add_argument("--upscale", type=int, default=1, help="beam table up-scaled by this factor, full grid rendered")
    p.add_argument("--distinct-frames", type=int, default=1600, help="frames kept resident in HBM (ring)")
    p.add_argument("--gather", choices=("compact", "tiles", "none"), default="compact",
                   help="N>1: what every step sends to rank 0")
    p.add_argument("--no-gather", action="store_true", help="same as --gather none")
    p.add_argument("--cpu-frames", type=int, default=128, help="frames of the CPU-oracle sample (0 = skip)")
    p.add_argument("--no-kernel-timing", action="store_true")
    p.add_argument("--no-other-configs", action="store_true", help="N=1: skip the short passes over the other shapes")
    p.add_argument("--no-pcie-fed", action="store_true", help="N=1: skip the host-fed rate measured after the clock")
    p.add_argument("--fused", type=int, choices=(0, 1), default=None,
                   help="decode + correction as one fused kernel (1) or as separate kernels (0); default: the library's")
    p.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                   help="N>1 process group; gloo (collectives staged through host memory) only to rehearse the "
                        "multi-rank control flow on a box with fewer GPUs than ranks")
    return p.parse_args()


def cpu_baseline(B, N, n_frames, tile):
    """The CPU oracle on the same synthetic frames (the NumPy generator is byte-identical to the device one), in
    worker processes of one core each (oracle/bench_worker.py), started before this process touches the GPU.
    value = frames of all workers / the slowest worker's time; the single-core rate is reported next to it."""
    import subprocess
    cores = max(1, min(16, os.cpu_count() or 1, n_frames))          # the box's CPU share for one GPU is 16
    per = max(1, n_frames // cores)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    worker = os.path.join(ROOT, "oracle", "bench_worker.py")
    import tempfile
    dump = os.path.join(tempfile.mkdtemp(prefix="slam_bench_"), "frame0.npz")     # worker 0 leaves frame 0's results here
    procs = [subprocess.Popen([sys.executable, worker, str(B), str(N), str(w * per), str(per), str(tile)] +
                              ([dump] if w == 0 else []), stdout=subprocess.PIPE, env=env, text=True) for w in range(cores)]
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=600)
        if p.returncode != 0:
            raise RuntimeError("CPU baseline worker failed")
        res.append(json.loads(out.strip().splitlines()[-1]))
    total = sum(r["n"] for r in res)
    slowest = max(r["dt"] for r in res)
    single = sum(r["n"] / r["dt"] for r in res) / len(res)
    ref_offline = None                  # the reference's own Python, timed once in the build container (it cannot travel)
    ref_json = os.path.join(ROOT, "profiles", "reference_cpu_timing.json")
    if os.path.exists(ref_json) and (B, N) == (256, 4096):
        rj = json.load(open(ref_json))
        ref_offline = dict(frames_per_second_one_thread=round(rj["implied_frames_per_second_one_thread"], 5),
                           seconds_per_frame=round(rj["implied_seconds_per_256x4096_frame"], 1), where=rj["what"])
    return dict(value=total / slowest, unit="frames/s", cores=cores, kind="port", single_core_value=round(single, 3),
                reference_offline=ref_offline, _dump=dump,
                sample=f"{total} synthetic {B}x{N} frames through oracle/ (sequential C scan + brute-force baseline "
                       f"search + NumPy blur/raster), {per} per worker process, {cores} single-threaded workers on "
                       f"{cores} of {os.cpu_count()} visible host cores, slowest worker {slowest:.1f} s "
                       f"({sum(r['dt'] for r in res):.0f} core-seconds)")


def parity_check(pipe, text, dump):
    """The device's results for synthetic frame 0 against what the CPU-baseline worker computed for the same frame
    (the checker's side of the cpu_baseline leg): counters, decoded rows, kept rows and the mean matrix must be equal,
    the blurred matrix within 1e-12, the tile's pixels equal.  One extra batch after the clock."""
    z = np.load(dump)
    pipe.run(text, text_base_offset=0)                  # batch 0 holds frames 0 .. F-1 of this rank
    st = pipe.stats_host()
    rows, kept = pipe.frame_rows(0, st), pipe.frame_kept(0, st)
    full, bl, _aod, _aoa, lo, hi = pipe.frame_matrix(0, st)
    rr, cc = np.flatnonzero(np.isfinite(full).any(axis=1)), np.flatnonzero(np.isfinite(full).any(axis=0))
    fin = np.isfinite(z["blurred"])
    blur_err = float(np.max(np.abs(bl[fin] - z["blurred"][fin]) / np.abs(z["blurred"][fin]))) if np.array_equal(np.isfinite(bl), fin) else float("inf")
    out = dict(
        frame=int(z["frame"]),
        counters_equal=[int(st[0, k]) for k in (2, 3, 4, 5, 6)] == z["counters"].tolist(),
        decoded_rows_equal=bool(np.array_equal(np.stack([rows[k].astype(np.int64) for k in ("flag", "ue", "bs", "rss", "clk")], 1), z["rows"])),
        kept_rows_equal=bool(np.array_equal(np.stack([kept[k].astype(np.int64) for k in ("ue", "bs", "rss", "clk")], 1), z["kept"])),
        matrix_equal=bool(np.array_equal(full[np.ix_(rr, cc)], z["matrix"], equal_nan=True)),
        blurred_max_rel_err=blur_err,
        tile_pixels_differing=int((pipe.frame_tile(0) != z["tile"]).any(axis=-1).sum()),
    )
    out["ok"] = bool(out["counters_equal"] and out["decoded_rows_equal"] and out["kept_rows_equal"] and out["matrix_equal"]
                     and blur_err <= 1e-12 and out["tile_pixels_differing"] == 0)
    return out


def measured_ceilings(ctx, pipe, achieved_gbs):
    """What this device sustains on plain streams, by the library's probe kernels and the same HIP-event timers as the
    pipeline's kernels: read-only, copy 1:1, and read 3 : write 1 (the framing kernel's mix).  Runs after every
    result has been read back: it streams over the tile and kept-row buffers and leaves them overwritten."""
    import ctypes as C
    src, dst = pipe.tiles.view(-1), pipe.k_rss.view(torch.uint8).view(-1)
    n = min(src.numel(), dst.numel(), 4 << 30) // 4096 * 4096
    if n < (64 << 20):
        return None
    sp, dp = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr())
    ctx.timing_reset()
    ctx.timing(True)
    for _ in range(6):
        ctx.check(ctx.lib.slam_copy_probe_dev(ctx.h, sp, sp, n))            # same pointer: read-only stream
        ctx.check(ctx.lib.slam_copy_probe_dev(ctx.h, dp, sp, n))
        ctx.check(ctx.lib.slam_read_probe_dev(ctx.h, dp, sp, n, 20))
    ctx.timing(False)
    t = ctx.timing_read()
    gbs = lambda name, nbytes: round(nbytes / (t[name][1] / t[name][0] * 1e-3) / 1e9, 1)      # noqa: E731
    mix = gbs("mix3_0", n + n // 3)
    return dict(read_GBps=gbs("read_probe", n), copy_GBps=gbs("copy_probe", 2 * n), read3_write1_GBps=mix,
                probe_bytes=n, frac_of_read3_write1=round(achieved_gbs / mix, 4))


def pcie_fed(pipe, text, stride, tl, dev, frames=64, batches=6):
    """After the clock, never part of `value`: the same pipeline fed over PCIe.  `frames` frames of text sit in pinned
    host memory and are copied batch by batch into two device buffers on a copy stream while the batch before is
    processed on a compute stream; also the plain host-to-device copy rate of the same buffer."""
    F_all = pipe.bt.n_frames
    frames = min(frames, F_all, text.numel() // stride // 2)
    if frames < 1:
        return None
    n = frames * stride
    host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    host.copy_(text[:n])
    bufs = [text[:n], text[n:2 * n]]
    pipe.set_uniform_layout(frames, stride, tl)
    copy_s, comp_s = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ready = [torch.cuda.Event() for _ in range(2)]
    freed = [torch.cuda.Event() for _ in range(2)]

    def run(k):
        for i in range(k):
            j = i & 1
            with torch.cuda.stream(copy_s):
                if i >= 2:
                    copy_s.wait_event(freed[j])
                bufs[j].copy_(host, non_blocking=True)
                ready[j].record(copy_s)
            with torch.cuda.stream(comp_s):
                comp_s.wait_event(ready[j])
                pipe.run(text, text_base_offset=j * n)
                freed[j].record(comp_s)
        torch.cuda.synchronize(dev)

    torch.cuda.synchronize(dev)
    run(3)
    t0 = time.perf_counter()
    run(batches)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(4):
        bufs[0].copy_(host, non_blocking=True)
    torch.cuda.synchronize(dev)
    link = 4 * n / (time.perf_counter() - t1) / 1e9
    pipe.set_uniform_layout(F_all, stride, tl)
    del host
    return dict(frames_per_s=round(frames * batches / dt, 1), h2d_GBps=round(link, 1), frames_per_batch=frames, batches=batches,
                note="text in pinned host memory, copied batch by batch on a copy stream while the batch before computes; "
                     "measured after the timed region, never part of `value`")


# the kernels one of which is the dominant one (decode, or decode + correction in one): timed inside the timed region
DOMINANT_KERNELS = ("frame_onepass", "frame_fused")

# The three builds of the fused correction kernel share one job: frames (or workgroups) the lean build leaves go to
# the mid or the full one, so their times add up against one set of algorithmic bytes.
KERNEL_GROUPS = {"rows_finish": ("rows_finish", "rows_finish_mid", "rows_finish_rest")}


def kernel_accounting(kern, nbytes, valid, kept, tile_bytes, frames_per_launch_of, B, N, upscale):
    """Per-kernel table and the roofline entry of the dominant kernel.  `kern`: name -> (launches, total ms) from the
    library's HIP-event timers.  Algorithmic bytes per frame (each kernel's own inputs and outputs once, re-reads not
    counted); kernels that split one job (KERNEL_GROUPS) are priced together."""
    alg = {
        "frame_onepass": 3.0 * nbytes + 11.0 * valid,    # text in, records out
        "frame_fused": 3.0 * nbytes + 11.0 * valid + 10.0 * kept,      # text in, records and kept rows out
        "tile_groups": 0.5 * valid,                      # 12 bits per 4 entries
        "rows_finish": 10.0 * valid + 10.0 * kept,       # flag, ue, rss, clk in; kept rows out
        "pivot": 6.0 * kept,
        "raster": float(tile_bytes),
    }
    merged = {}
    for name, (launches, ms) in kern.items():
        group = next((g for g, members in KERNEL_GROUPS.items() if name in members), name)
        e = merged.setdefault(group, dict(launches=0, ms=0.0, parts={}))
        e["ms"] += ms
        e["launches"] = max(e["launches"], launches)      # members are launched once per batch each
        e["parts"][name] = (launches, ms)
    tot_ms = sum(e["ms"] for e in merged.values())
    table = {}
    for name, e in sorted(merged.items(), key=lambda kv: -kv[1]["ms"]):
        per = e["ms"] / max(e["launches"], 1)
        ent = dict(launches=e["launches"], avg_ms=round(per, 5), share=round(e["ms"] / tot_ms, 4))
        if len(e["parts"]) > 1:
            ent["parts_avg_ms"] = {k: round(ms / max(l, 1), 5) for k, (l, ms) in e["parts"].items()}
        if name in alg:
            gbps = alg[name] * frames_per_launch_of(e["launches"]) / (per * 1e-3) / 1e9
            if gbps <= HBM_PEAK_GBS:
                ent["alg_GBps"] = round(gbps, 1)
            else:                  # a share of the job ran elsewhere or the launch did next to nothing: no figure
                ent["alg_GBps"] = None
                ent["alg_note"] = "launch too short for its nominal bytes (work done by another kernel)"
        table[name] = ent
    dom = max((n for n in merged if n in alg), key=lambda n: merged[n]["ms"])
    per = merged[dom]["ms"] / merged[dom]["launches"]
    fpl = frames_per_launch_of(merged[dom]["launches"])
    ach = alg[dom] * fpl / (per * 1e-3) / 1e9
    traffic, note, pipe_traffic = None, None, None
    import glob
    tag = f"{B}x{N}" + (f"_up{upscale}" if upscale > 1 else "")
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True)   # newest first; shape checked below
    for pmc in cands:
        pj = json.load(open(pmc))                 # HBM bytes per launch from committed rocprofv3 PMC passes of this shape
        parts = [k for k in KERNEL_GROUPS.get(dom, (dom,)) if k in pj["kernels"]]
        if pj.get("shape") == [B, N] and pj.get("upscale", 1) == upscale and parts:
            traffic = int(sum(pj["kernels"][k]["fetch_bytes_per_frame"] + pj["kernels"][k]["write_bytes_per_frame"]
                              for k in parts) * fpl)
            note = os.path.relpath(pmc, ROOT)
            pipe_traffic = pj.get("pipeline_bytes_per_frame")      # every kernel of the batch, per frame
            break
    if traffic is None:
        note = f"no committed PMC pass for shape {tag}"
    roof = dict(bound="hbm", kernel=dom, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=note,
                alg_bytes_per_launch=int(alg[dom] * fpl), frames_per_launch=fpl, avg_launch_ms=round(per, 5),
                pipeline_traffic_per_frame=pipe_traffic)
    return table, roof, ach


class Job:
    """Distributed plumbing shared by every measurement of one process."""

    def __init__(self, a):
        self.a = a
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        self.local = local % max(1, torch.cuda.device_count())          # ranks share GPUs only in a --backend gloo rehearsal
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        self.ranks_seen, self.comm_version, self.comm_name = 1, None, None
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if a.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
                self.comm_name = "RCCL over xGMI"
                try:
                    self.comm_version = "rccl/nccl " + ".".join(str(x) for x in torch.cuda.nccl.version())
                except Exception as e:               # noqa: BLE001
                    self.comm_version = f"nccl (version unavailable: {e})"
            else:
                dist.init_process_group("gloo")
                self.comm_name = "gloo, staged through host memory: a rehearsal, not RCCL"
                self.comm_version = "gloo (host-staged rehearsal)"
            ones = torch.ones(1, dtype=torch.int64, device=self.dev if a.backend == "nccl" else "cpu")
            dist.all_reduce(ones)                      # every rank the process group really holds adds one
            self.ranks_seen = int(ones.item())

    def gather_to_root(self, src, out_list):
        """dist.gather of a device tensor to rank 0 (RCCL; through host memory under gloo)."""
        if self.a.backend == "nccl":
            self.dist.gather(src, out_list if self.rank == 0 else None, dst=0)
            return
        host = src.cpu()
        tmp = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(host, tmp, dst=0)
        if self.rank == 0:
            for o, t in zip(out_list, tmp):
                o.copy_(t)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def reduce_scalar(self, v, op):
        if self.world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=self.dev if self.a.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=op)
        return float(t.item())


def measure(job, B, N, F, K, W, n_distinct_want, tile, upscale, gmode, timing, extra_modes=()):
    """W warm-up steps, then K timed steps of F frames of shape B x N on every rank (barrier + synchronize on both
    sides, max over ranks).  Returns the result dict plus the live objects the follow-up checks need."""
    from slamhip import _lib, synth
    from slamhip.pipeline import FramePipeline, default_mapping, expand_compact, synth_frames, viridis_lut
    a, dev, rank, world = job.a, job.dev, job.rank, job.world
    mapping, lut = default_mapping(), viridis_lut()
    tl = synth.text_len(B, N)
    stride = ((tl + 255) // 256) * 256
    n_distinct = max(F, min(n_distinct_want, (K + W) * F))
    n_distinct = (n_distinct // F) * F
    modes = [gmode] + [m for m in extra_modes if m != gmode]
    tiles_mode = world > 1 and "tiles" in modes
    # fit the batch to the memory that is free: text ring + per frame about 2.5 x its text (row arrays, kept rows,
    # library workspace; 89 MB measured for a 34.6 MB frame) + its tile slots + the wide grids of an up-scaled table
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    n_axis = 64 * upscale
    wide = (n_axis * n_axis * (8 + 4 + 4 + 8 + 8 + 8 + 8) if n_axis > 64 else 0)
    per_frame = 2.5 * stride + tile * tile * 4 * (2 if tiles_mode else 1) + wide + (1 << 20)
    need = n_distinct * stride + F * per_frame
    if need > 0.92 * free_b:
        scale = 0.92 * free_b / need
        F = max(8, int(F * scale) // 8 * 8)
        n_distinct = min(n_distinct, 2 * F) // F * F
        print(f"bench: {free_b / 2**30:.0f} GiB free, batch reduced to {F} frames/step ({n_distinct} resident)", file=sys.stderr)
    if world > 1:                                            # every rank runs the same batch size
        Fm = int(job.reduce_scalar(float(F), job.dist.ReduceOp.MIN))
        if Fm != F:
            F = Fm
            n_distinct = min(n_distinct, 2 * F) // F * F
    # frames of this rank: ids rank*n_distinct .. so every rank works on different data
    text = synth_frames(n_distinct, B, N, first_frame=rank * n_distinct, device=job.local)
    any_gather = any(m != "none" for m in modes)
    pipe = FramePipeline(F, tl, device=job.local, tile_w=tile, tile_h=tile, mapping=mapping, lut=lut, upscale=upscale,
                         n_tile_slots=2 * F if tiles_mode else F, n_compact_slots=2 if any_gather else 1, fused=a.fused if a.fused is None else bool(a.fused))
    pipe.set_uniform_layout(F, stride, tl)
    ctx = pipe.ctx
    torch.cuda.synchronize(dev)
    comm_stream = torch.cuda.Stream(dev) if any_gather else None

    def run_mode(mode, with_timing):
        gather = mode != "none"
        recv = None
        if gather and rank == 0:
            if mode == "tiles":
                recv = [torch.empty((F, tile, tile, 4), dtype=torch.uint8, device=dev) for _ in range(world)]
            else:
                recv = [torch.empty_like(pipe.compact[0]) for _ in range(world)]
        pending = []

        def step(i):
            batch = (i * F) % n_distinct
            slot = (i % 2) * F if mode == "tiles" else 0
            cslot = i % 2 if gather else 0
            pipe.run(text, text_base_offset=batch * stride, tile_slot=slot, compact_slot=cslot)
            if gather:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                with torch.cuda.stream(comm_stream):
                    comm_stream.wait_event(ev)
                    src = pipe.tiles[slot:slot + F] if mode == "tiles" else pipe.compact[cslot]
                    job.gather_to_root(src, recv)
                    done = torch.cuda.Event()
                    done.record(comm_stream)
                # the slot is reused two steps later: compute must wait for this gather
                pending.append(done)
                if len(pending) >= 2:
                    torch.cuda.current_stream(dev).wait_event(pending.pop(0))

        for i in range(W):
            step(i)
        torch.cuda.synchronize(dev)
        if with_timing:
            # "dominant": only the kernel the roofline entry prices is bracketed by events (two per step); "all": every
            # launch (a few dozen per step, which costs the pipeline about 4 % of its rate)
            ctx.timing_select(DOMINANT_KERNELS if with_timing == "dominant" else None)
            ctx.timing_reset()
            ctx.timing(True)
        # one event per step boundary on the compute stream (recording costs nothing to speak of): the spread of the
        # K step times tells a change of a few per cent from the noise of a box
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        job.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        marks[0].record(torch.cuda.current_stream(dev))
        for i in range(K):
            step(W + i)
            marks[i + 1].record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        job.barrier()
        t1 = time.perf_counter()
        step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(K)]
        kern = {}
        if with_timing:
            ctx.timing(False)
            kern = ctx.timing_read()
        dt = t1 - t0
        verified = None
        if mode == "compact":
            # after the clock: the last batch's full tiles of every rank travel once, and rank 0 repaints the compact
            # results it received for that batch; both must be the same bytes
            full = [torch.empty((F, tile, tile, 4), dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
            job.gather_to_root(pipe.tiles[:F], full)
            if rank == 0:
                verified = all(bool(torch.equal(expand_compact(recv[r], F, tile, tile, cells_per_frame=pipe.cells_per_frame),
                                                full[r])) for r in range(world))
                del full
        if world > 1:
            dt = job.reduce_scalar(dt, job.dist.ReduceOp.MAX)
        del recv
        return dt, kern, verified, step_ms

    dt, kern, verified, step_ms = run_mode(gmode, "dominant" if timing else None)
    gather_rates = {gmode: dict(frames_per_s=round(K * F * world / dt, 2), ms_per_step=round(dt / K * 1e3, 4),
                                gather_verified=verified)}
    for m in modes[1:]:                                      # the other gather mode(s), after the headline's clock
        dt2, _k, ver2, _s = run_mode(m, False)
        gather_rates[m] = dict(frames_per_s=round(K * F * world / dt2, 2), ms_per_step=round(dt2 / K * 1e3, 4),
                               gather_verified=ver2)

    # per-frame counts of the last batch for the byte accounting
    st = pipe.stats_host().astype(np.int64)
    nbytes = float(st[:, _lib.STAT_NBYTES].mean())
    valid = float(st[:, _lib.STAT_VALID].mean())
    kept = float(st[:, _lib.STAT_KEPT].mean())
    assert int(st[:, _lib.STAT_NOT_STRICT].sum()) == 0
    kappa = kept / max(valid, 1.0)
    tile_bytes = tile * tile * 4
    # algorithmic bytes per frame: read the text (3 per data byte), write decoded records (11 each),
    # write kept rows (10 each), write the tile
    alg_frame = 3.0 * nbytes + 11.0 * valid + 10.0 * kept + tile_bytes
    fps_rank = K * F / dt
    table, roof, ach = ({}, None, 0.0)
    all_pass_ms = None
    if kern:
        # the other kernels' times: the same K steps once more after the clock, every launch bracketed by events; the
        # dominant kernel keeps the time it had inside the timed region
        dt_all, kern_all, _v, _s = run_mode("none", "all")
        all_pass_ms = round(dt_all / K * 1e3, 4)
        kern = {**kern_all, **kern}
    if kern:
        table, roof, ach = kernel_accounting(kern, nbytes, valid, kept, tile_bytes, lambda launches: F * K / max(launches, 1),
                                             B, N, upscale)
    served = int(((st[:, _lib.STAT_LAYOUT] & _lib.LAYOUT_KEPT_TILES) != 0).sum())
    why = st[:, _lib.STAT_LAYOUT] >> _lib.LAYOUT_FUSED_WHY_SHIFT              # frames the fused kernel handed on, by reason bits
    why = {int(w): int((why == w).sum()) for w in np.unique(why) if w}
    res = dict(value=fps_rank * world, ms_per_step=dt / K * 1e3, F=F, n_distinct=n_distinct, tl=tl, kappa=kappa,
               fused=pipe.ctx.get_fused(), fused_frames=served, fused_gave_up=why,
               step_ms=dict(min=round(min(step_ms), 4), median=round(float(np.median(step_ms)), 4), max=round(max(step_ms), 4),
                            slowest_step=int(np.argmax(step_ms)), all=[round(x, 3) for x in step_ms] if len(step_ms) <= 40 else None,
                            note="this rank's compute stream, one HIP event per step boundary inside the timed region"),
               alg_frame=alg_frame, fps_rank=fps_rank, kernels=table, roofline=roof, achieved=ach, all_pass_ms=all_pass_ms,
               gather_verified=verified, gather_rates=gather_rates if world > 1 else None)
    return res, pipe, text


def short_config(job, name, B, N, F, K, W, tile, upscale):
    """One of the other shapes, briefly: whole-pipeline rate and HBM fraction, the two largest kernels."""
    res, pipe, text = measure(job, B, N, F, K, W, F, tile, upscale, "none", True)
    kern = res["kernels"]
    top = sorted(kern.items(), key=lambda kv: -kv[1]["share"])[:3]
    out = dict(workload=f"synthetic {B}x{N} hex frames ({B * N} records, {res['tl']} text bytes each)" +
                        (f", beam table up-scaled x{upscale} ({64 * upscale} entries, full grid; parity unpinned: the "
                         f"reference admits 64 beams)" if upscale > 1 else ""),
               value=round(res["value"], 2), unit="frames/s", frames_per_step=res["F"], steps=K, warmup=W,
               ms_per_step=round(res["ms_per_step"], 4), kappa_kept_ratio=round(res["kappa"], 4),
               fused_frames=res["fused_frames"] if res["fused"] else None,
               alg_bytes_per_frame=int(res["alg_frame"]),
               pipeline_alg_GBps=round(res["fps_rank"] * res["alg_frame"] / 1e9, 1),
               pipeline_hbm_frac=round(res["fps_rank"] * res["alg_frame"] / 1e9 / HBM_PEAK_GBS, 4),
               dominant_kernel=dict(name=res["roofline"]["kernel"], avg_launch_ms=res["roofline"]["avg_launch_ms"],
                                    alg_GBps=res["roofline"]["achieved"], frac=res["roofline"]["frac"],
                                    alg_bytes_per_launch=res["roofline"]["alg_bytes_per_launch"],
                                    traffic=res["roofline"]["traffic"], traffic_source=res["roofline"]["traffic_source"]),
               pipeline_traffic_per_frame=res["roofline"]["pipeline_traffic_per_frame"],
               top_kernels={k: dict(avg_ms=v["avg_ms"], share=v["share"], alg_GBps=v.get("alg_GBps")) for k, v in top})
    del pipe, text
    torch.cuda.empty_cache()
    return out


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and a.gpus == 1:
        a.gpus = world                      # started under torchrun without --gpus: the launcher's world size counts
    if world != a.gpus and "WORLD_SIZE" in os.environ:
        print(f"bench: --gpus {a.gpus} but the launcher started {world} rank(s); no line printed", file=sys.stderr)
        sys.exit(3)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly with --gpus N: run the N ranks as a child job (nothing here has touched the GPU yet)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        sys.exit(subprocess.run(cmd).returncode)
    cpu_line = None
    if rank == 0 and a.cpu_frames > 0:      # rank 0 of any world size, before anything here initialises the GPU
        cpu_line = cpu_baseline(a.beams, a.samples, a.cpu_frames, a.tile)
    job = Job(a)
    if job.ranks_seen != a.gpus:            # a rank that never joined: a line over fewer GPUs than asked for would be wrong
        print(f"bench: the process group holds {job.ranks_seen} rank(s), --gpus asked for {a.gpus}; no line printed", file=sys.stderr)
        sys.exit(3)
    B, N, K, W = a.beams, a.samples, a.steps, a.warmup
    gmode = "none" if (world == 1 or a.no_gather) else a.gather
    extra = ()
    if world > 1 and gmode != "none":       # the other way of collecting results, timed after the headline
        extra = ("tiles",) if gmode == "compact" else ("compact",)
    res, pipe, text = measure(job, B, N, a.frames_per_step, K, W, a.distinct_frames, a.tile, a.upscale, gmode,
                              not a.no_kernel_timing, extra)
    F, roof = res["F"], res["roofline"]

    if cpu_line is not None:
        dump = cpu_line.pop("_dump")
        if os.path.exists(dump):
            if a.upscale == 1:
                cpu_line["parity_check"] = parity_check(pipe, text, dump)
            os.remove(dump)
            os.rmdir(os.path.dirname(dump))
    if roof is not None and rank == 0:
        roof["measured_ceilings"] = measured_ceilings(pipe.ctx, pipe, res["achieved"])

    fed = None
    if world == 1 and not a.no_pcie_fed and rank == 0:
        try:
            stride = ((res["tl"] + 255) // 256) * 256
            fed = pcie_fed(pipe, text, stride, res["tl"], job.dev)
        except Exception as e:                   # noqa: BLE001  (the headline line must still be printed)
            fed = dict(error=f"{type(e).__name__}: {e}")

    other = None
    if world == 1 and not a.no_other_configs and (B, N, a.upscale) == (256, 4096, 1) and not a.no_kernel_timing:
        # short passes over the other two synthetic shapes the benchmark is specified for, after the headline region;
        # the headline's buffers are released first
        del pipe, text
        torch.cuda.empty_cache()
        other = {}
        for name, (b2, n2, f2, up2) in {"256x16384": (256, 16384, 200, 1), "1024x8192_up4": (1024, 8192, 100, 4),
                                        "1024x8192": (1024, 8192, 100, 1)}.items():
            try:
                other[name] = short_config(job, name, b2, n2, f2, 5, 1, a.tile, up2)
            except Exception as e:               # noqa: BLE001  (the headline line must still be printed)
                other[name] = dict(error=f"{type(e).__name__}: {e}")

    records = B * N
    out = {
        "metric": "frames/s (256-beam x 4096-sample decode+correct+raster)",
        "value": round(res["value"], 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(res["ms_per_step"], 4), "step_ms": res["step_ms"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"synthetic {B}x{N} hex frames ({records} records, {res['tl']} text bytes each), "
                               f"{F} frames/step, {res['n_distinct']} distinct frames resident in HBM per GPU, "
                               f"tile {a.tile}x{a.tile} RGBA" + (f", beam table up-scaled x{a.upscale}" if a.upscale > 1 else ""),
                   "frames_per_step": F, "beams": B, "samples": N,
                   "decode_correct": ("one fused kernel" if res["fused"] else "separate kernels") +
                                     (f" ({res['fused_frames']} of {F} frames of the last batch served by it" +
                                      (f"; handed on, by reason bits: {res['fused_gave_up']}" if res["fused_gave_up"] else "") + ")" if res["fused"] else ""),
                   "parallelism": f"frames sharded over {world} GPU(s)" +
                                  {"none": "", "tiles": f", 4 MiB tiles gathered to rank 0 every step ({job.comm_name})",
                                   "compact": ", tiles gathered to rank 0 every step in compact form (cell colours + "
                                              f"pixel maps, repainted bit-exactly by slam_raster_dev; {job.comm_name})"}[gmode],
                   "gather": gmode, "gather_verified": res["gather_verified"], "gather_rates": res["gather_rates"],
                   "ranks_seen": job.ranks_seen, "comm": job.comm_version},
        "kappa_kept_ratio": round(res["kappa"], 4),
        "alg_bytes_per_frame": int(res["alg_frame"]),
        "pipeline_alg_GBps": round(res["fps_rank"] * res["alg_frame"] / 1e9, 1),
        "pipeline_hbm_frac": round(res["fps_rank"] * res["alg_frame"] / 1e9 / HBM_PEAK_GBS, 4),
        "roofline": roof,
        "pcie_fed": fed,
        "kernels": res["kernels"],
        "kernels_note": (f"{'/'.join(DOMINANT_KERNELS)}: HIP events inside the timed region; the other kernels: the same {K} steps "
                         f"once more after the clock with every launch bracketed by events ({res['all_pass_ms']} ms per step that way)"
                         if res["all_pass_ms"] is not None else None),
        "other_configs": other,
    }
    if rank == 0:
        out["cpu_baseline"] = cpu_line
        print(json.dumps(out))
    if world > 1:
        job.dist.destroy_process_group()


if __name__ == "__main__":
    main()
