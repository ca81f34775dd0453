"""Multi-GPU layout of the P-frame path: one process per GPU.

Inference: GOPs / sequences are independent (tools/predict.py:51 resets the reference list per
GOP), frames inside a GOP are strictly serial -> GOPs are dealt round-robin to ranks and NO
collective touches the data path; only per-frame scalars (bpp, PSNR, ...) are gathered at the end.
"""
from __future__ import annotations

import torch.distributed as dist


def shard_gops(n_gops: int, world: int, rank: int) -> list:
    """indices of the GOPs rank `rank` codes (round-robin, so long and short sequences mix)"""
    return list(range(rank, n_gops, world))


def gather_frame_stats(stats: list) -> list:
    """all-gather python dicts of per-frame scalars and return them sorted by (gop, frame)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sorted(stats, key=lambda s: (s["gop"], s["frame"]))
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, stats)
    flat = [s for part in out for s in part]
    return sorted(flat, key=lambda s: (s["gop"], s["frame"]))
