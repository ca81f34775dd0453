#!/usr/bin/env python
"""Counterpart of the reference's tools/train.py loop on the MI355X path.

The reference script wraps the model in nn.DataParallel, needs the Vimeo-90k tree plus cv2 / natsort / albumentations,
and a tensorboard writer; this driver keeps what defines the optimisation — the sample rule of
`main/dataloader/dataset.py:211-247` (input frame t of a septuplet, references [im1, t-3 .. t-1] padded by repetition,
plus the long-range sample im7 <- [im1, im1, im3, im5]), `rd_loss = train_lambda * MSE + bpp_res + bpp_mv`, the
optimizer / clip / aux-optimizer order (`tools/train.py:136-152`), checkpoint naming `{iter}_lambda{λ}.pth` and
`latest.pth` (`:199-203`) — on synthetic septuplets (`synth.make_gop`), one process per GPU with the batch sharded
by rank and the gradients averaged over RCCL (`train.GradBuckets`).

  python -m tdvc_amd.tools.train --iters 20 --batch 4 --size 256
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m tdvc_amd.tools.train --batch 4
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch

from ..model import VideoCompressor
from ..synth import fill_parameters, make_gop
from ..train import TrainStep


def septuplet_samples(frames: torch.Tensor):
    """all (input, refs) pairs of one septuplet, dataset.py:211-247.  frames: (7,3,H,W) -> list of ((3,H,W), (4,3,H,W))"""
    out = []
    for start in range(1, 7):                      # input = im{start+1} (1-based), refs from the ORIGINAL frames
        idx = [1] + list(range(max(start + 1 - 3, 1), start + 1))
        idx += [idx[-1]] * (4 - len(idx))
        out.append((frames[start], frames[[i - 1 for i in idx]]))
    out.append((frames[6], frames[[0, 0, 2, 4]]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4, help="samples per rank")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--train-lambda", type=float, default=2048.0)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--loss-scale", type=float, default=128.0)
    ap.add_argument("--pretrain", default="")
    ap.add_argument("--save-dir", default="")
    ap.add_argument("--save-every", type=int, default=10000)
    ap.add_argument("--latest-every", type=int, default=2000, help="iterations between latest.pth saves (tools/train.py:168,196: every 2000)")
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--vimeo", default="", help="Vimeo-septuplet root (<dir>/<clip>/im1..7.png): tdvc_amd.data.DataSet with the "
                                                "reference's sample rule and augmentation (train.py:78-80); default: synthetic septuplets")
    ap.add_argument("--num-workers", type=int, default=4)
    a = ap.parse_args()

    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    torch.manual_seed(a.seed + rank)

    net = VideoCompressor()
    if a.pretrain:
        net.load_state_dict(torch.load(a.pretrain, map_location="cpu"))
    else:
        fill_parameters(net)
    net = net.to(dev)
    step = TrainStep(net, train_lambda=a.train_lambda, lr=a.lr, loss_scale=a.loss_scale)

    pool, cursor = [], 0
    t0 = time.time()
    loader = None
    if a.vimeo:                                         # train.py:78-80: shuffled DataLoader over the septuplet samples
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler

        from ..data import DataSet
        ds = DataSet(a.vimeo, resize_size=a.size, seed=a.seed + rank)
        sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=a.seed) if world > 1 else None

        def _reseed(worker_id):                         # every loader worker draws its own augmentations
            import numpy as np
            from torch.utils.data import get_worker_info
            get_worker_info().dataset.rng = np.random.default_rng(a.seed + 7919 * rank + 104729 * (worker_id + 1))
        dl = DataLoader(ds, batch_size=a.batch, shuffle=sampler is None, sampler=sampler, num_workers=a.num_workers, drop_last=True,
                        worker_init_fn=_reseed)

        def _cycle():
            epoch = 0
            while True:
                if sampler is not None:
                    sampler.set_epoch(epoch)
                yield from dl
                epoch += 1
        loader = _cycle()
    for it in range(a.iters):
        if loader is not None:
            x, refs = next(loader)
            x, refs = x.to(dev), refs.to(dev)
        else:
            while len(pool) < a.batch:                  # every rank walks its own septuplets (seed = base + global index)
                pool += septuplet_samples(make_gop(a.seed + (cursor * world + rank), 7, a.size, a.size))
                cursor += 1
            batch, pool = pool[:a.batch], pool[a.batch:]
            x = torch.stack([b[0] for b in batch]).to(dev)
            refs = torch.stack([b[1] for b in batch]).to(dev)
        log = step(x, refs)
        if rank == 0:
            psnr = 10.0 * torch.log10(torch.tensor(1.0 / max(log["mse"], 1e-12))).item()
            print(json.dumps({"iter": it + 1, "rd_loss": round(log["rd_loss"], 4), "psnr": round(psnr, 3),
                              "bpp": round(log["bpp_res"] + log["bpp_mv"], 4), "aux": round(log["aux_loss"], 2),
                              "grad_norm": round(log["grad_norm"], 3), "s_per_iter": round((time.time() - t0) / (it + 1), 3)}), flush=True)
            if a.save_dir and ((it + 1) % a.latest_every == 0 or (it + 1) % a.save_every == 0 or it + 1 == a.iters):
                os.makedirs(a.save_dir, exist_ok=True)
                torch.save(net.state_dict(), os.path.join(a.save_dir, "latest.pth"))
                if (it + 1) % a.save_every == 0:
                    torch.save(net.state_dict(), os.path.join(a.save_dir, f"{it + 1}_lambda{int(a.train_lambda)}.pth"))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
