"""Batch driver: capture logs in, one data tile (PNG) and one summary line per log out.

    python -m slamhip.run_logs [--out-dir DIR] [--tile 1024] [--backend nccl|gloo] LOG_OR_DIR ...

Single process: every log goes through decode -> correct -> pivot -> blur -> raster in one batched device call.
Under torch.distributed (started with torchrun, one process per GPU): the logs are balanced over the ranks by
size, each rank processes its own, the tiles are gathered to rank 0 (RCCL), and rank 0 writes the files.  This is
the chain of the reference's three scripts (serial_hex_to_excel_v3 -> bs_beam_correction -> excel_heatmap_v3 with
its CLI defaults: linear scale, viridis, blur sigma 1.0) without the workbooks in between.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
from typing import List


def collect(paths: List[str]) -> List[str]:
    out = []
    for p in paths:
        if os.path.isdir(p):
            out += sorted(glob.glob(os.path.join(p, "*.txt")))
        else:
            out.append(p)
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("logs", nargs="+", help="capture .txt files, or directories of them")
    ap.add_argument("--out-dir", default=None, help="where the PNG tiles go (default: heatmap_outputs next to each log)")
    ap.add_argument("--tile", type=int, default=1024)
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl")
    a = ap.parse_args(argv)
    paths = collect(a.logs)
    missing = [p for p in paths if not os.path.exists(p)]
    if missing or not paths:
        print(f"输入文件不存在: {missing or a.logs}", file=sys.stderr)
        return 1

    import torch
    from . import _lib, batch, png
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    timing = {}
    tiles, res = batch.process_files_sharded(paths, device=local, tile=a.tile, gather_via_host=a.backend == "gloo",
                                             timing=timing)
    # every rank reports its own logs' counters; rank 0 also writes the gathered tiles
    from . import dist as D
    mine = D.balance_files([os.path.getsize(p) for p in paths], world)[rank]
    for i, r in zip(mine, res):
        st = r["stats"]
        line = json.dumps(dict(log=paths[i], rank=rank, valid=int(st[_lib.STAT_VALID]), discarded=int(st[_lib.STAT_DISCARDED]),
                               groups=int(st[_lib.STAT_GROUPS]), baselines=int(st[_lib.STAT_BASELINES]),
                               kept=int(st[_lib.STAT_KEPT]), grid=[len(r["aod"]), len(r["aoa"])]), ensure_ascii=False)
        sys.stdout.write(line + "\n")          # one write per line: the ranks share the launcher's pipe
        sys.stdout.flush()
    # throughput of this rank's share: logs and text bytes per second through the device chain (file reading and the
    # gather are reported beside it, not inside it)
    proc = max(timing.get("process_s", 0.0), 1e-9)
    sys.stdout.write(json.dumps(dict(summary=True, rank=rank, world=world, logs=timing.get("logs", 0),
                                     text_MB=round(timing.get("text_bytes", 0) / 1e6, 3),
                                     logs_per_s=round(timing.get("logs", 0) / proc, 2),
                                     text_MB_per_s=round(timing.get("text_bytes", 0) / 1e6 / proc, 1),
                                     read_s=round(timing.get("read_s", 0.0), 4), process_s=round(proc, 4),
                                     gather_s=round(timing.get("gather_s", 0.0), 4))) + "\n")
    sys.stdout.flush()
    if rank == 0:
        for p, t in zip(paths, tiles):
            d = a.out_dir or os.path.join(os.path.dirname(os.path.abspath(p)), "heatmap_outputs")
            os.makedirs(d, exist_ok=True)
            png.write_rgba(os.path.join(d, os.path.splitext(os.path.basename(p))[0] + "_tile.png"), t)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
