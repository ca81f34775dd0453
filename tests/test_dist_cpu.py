"""CPU, world_size 2 over gloo: the GOP-sharded inference path has no data-path collective —
ranks own disjoint GOPs and only per-frame scalars are gathered."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tdvc_amd.parallel import gather_frame_stats, shard_gops


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_gops(7, world, rank)
    stats = [{"gop": g, "frame": f, "bpp": 0.1 * g + 0.01 * f, "psnr": 30.0 + g} for g in mine for f in range(1, 3)]
    allstats = gather_frame_stats(stats)
    if rank == 0:
        q.put((mine, allstats))
    dist.barrier()
    dist.destroy_process_group()


def test_gop_sharding_and_stat_gather():
    assert shard_gops(7, 2, 0) == [0, 2, 4, 6] and shard_gops(7, 2, 1) == [1, 3, 5]
    assert sorted(sum((shard_gops(13, 8, r) for r in range(8)), [])) == list(range(13))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    mine, allstats = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert mine == [0, 2, 4, 6]
    assert len(allstats) == 14 and [s["gop"] for s in allstats] == sorted(s["gop"] for s in allstats)
    assert {s["gop"] for s in allstats} == set(range(7))
