#!/usr/bin/env python
"""BASELINE.json configs[4]: the BPP / PSNR / MS-SSIM curve over lambda in {256, 512, 1024, 2048}, GOP-sharded over the
GPUs of one node.

The reference evaluates ONE checkpoint per `tools/predict.py` run and recovers lambda from the checkpoint's file name
(`..._lambda{L}.pth`, tools/predict.py:131; saved as `{iter}_lambda{L}.pth`, tools/train.py:201-203).  This driver does
the whole sweep in one launch: the work items are (lambda, GOP) pairs dealt round-robin to the ranks — GOPs are
independent (predict.py:51 resets the reference list), so there is no collective on the data path; a rank reloads the
state-dict only when its next item has another lambda.  Per-frame scalars are gathered once at the end and rank 0
prints one table (and optionally writes the reference's result files, predict.py:102-108, one per lambda).

  python -m torch.distributed.run --nproc-per-node 8 -m tdvc_amd.tools.rd_sweep --ckpt-dir ckpts --gops 8
  python -m tdvc_amd.tools.rd_sweep --gops 2 --height 256 --width 256          # synthetic filler weights for every lambda

Without checkpoints every lambda codes with the same closed-form filler weights (no checkpoint ships with the
reference): the table then checks the plumbing, not rate-distortion.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import re
import time

import torch

from ..model import VideoCompressor
from ..parallel import gather_frame_stats
from ..synth import fill_parameters, make_gop
from .predict import code_gop

LAMBDAS = (256, 512, 1024, 2048)


def lambda_from_name(path: str) -> int:
    """`tools/predict.py:131`: int(pretrain.split('_lambda')[-1].split('.')[0])"""
    m = re.search(r"_lambda(\d+)\.pth$", os.path.basename(path))
    if not m:
        raise ValueError(f"no _lambda<L>.pth suffix in {path!r} (tools/train.py:201-203 names checkpoints '<iter>_lambda<L>.pth')")
    return int(m.group(1))


def find_checkpoints(ckpt_dir: str | None, lambdas) -> dict:
    """lambda -> newest checkpoint of that lambda in `ckpt_dir` (highest iteration number), or None"""
    out = {int(l): None for l in lambdas}
    if not ckpt_dir:
        return out
    best = {}
    for f in glob.glob(os.path.join(ckpt_dir, "*_lambda*.pth")):
        lam = lambda_from_name(f)
        it = re.match(r"(\d+)_lambda", os.path.basename(f))
        key = int(it.group(1)) if it else -1
        if lam in out and (lam not in best or key > best[lam][0]):
            best[lam] = (key, f)
    for lam, (_, f) in best.items():
        out[lam] = f
    return out


def work_items(lambdas, n_gops: int, world: int, rank: int) -> list:
    """(lambda, gop) pairs of rank `rank`: the flat list [(l0, g0), (l0, g1), ..., (l1, g0), ...] dealt round-robin, so that
    every rank is busy for any lambda count and consecutive items of a rank mostly share a lambda (few reloads)"""
    flat = [(int(l), g) for l in lambdas for g in range(n_gops)]
    return flat[rank::world]


def assemble_table(allstats: list, lambdas) -> list:
    """per-lambda means over all coded frames, in the order of `lambdas` (predict.py:98-100)"""
    rows = []
    for lam in lambdas:
        fr = [s for s in allstats if s["lambda"] == int(lam)]
        n = max(1, len(fr))
        ms = [s["msssim"] for s in fr if s["msssim"] == s["msssim"]]
        rows.append({"lambda": int(lam), "frames": len(fr), "bpp": sum(s["bpp"] for s in fr) / n, "psnr": sum(s["psnr"] for s in fr) / n,
                     "msssim": sum(ms) / len(ms) if ms else float("nan")})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lambdas", type=int, nargs="+", default=list(LAMBDAS))
    ap.add_argument("--ckpt-dir", default=None, help="directory of '<iter>_lambda<L>.pth' state-dicts; missing lambdas use the synthetic filler")
    ap.add_argument("--gops", type=int, default=8, help="synthetic UVG-shape GOPs per lambda (seeds 2000 + gop, SURVEY 8d)")
    ap.add_argument("--gop-size", type=int, default=7)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--fp32-coders", action="store_true", help="enable_amp = False: both coders as fp32 islands")
    ap.add_argument("--out-dir", default=None, help="write '<L>.txt' result files in the layout of tools/predict.py:102-108")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(dev)
    ckpts = find_checkpoints(a.ckpt_dir, a.lambdas)
    net, loaded = VideoCompressor(), None
    t0 = time.time()
    stats = []
    with torch.no_grad():
        for lam, g in work_items(a.lambdas, a.gops, world, rank):
            if loaded != lam:
                if ckpts[lam]:
                    net.load_state_dict(torch.load(ckpts[lam], map_location="cpu"), strict=True)      # predict.py:150
                else:
                    fill_parameters(net)
                    net.clear_packed()
                net = net.to(dev).eval()
                loaded = lam
            frames = make_gop(2000 + g, a.gop_size, a.height, a.width).to(dev)
            for s in code_gop(net, frames, not a.fp32_coders):
                s.update(gop=g, **{"lambda": lam})
                stats.append(s)
    torch.cuda.synchronize()
    # gather_frame_stats sorts by (gop, frame); the lambda column keeps the rows apart
    allstats = gather_frame_stats(stats)
    if rank == 0:
        rows = assemble_table(allstats, a.lambdas)
        res = {"curve": rows, "gops_per_lambda": a.gops, "frames": len(allstats), "seconds": round(time.time() - t0, 2), "n_gpus": world,
               "size": [a.height, a.width], "checkpoints": {str(k): v for k, v in ckpts.items()},
               "coders": "fp32 islands" if a.fp32_coders else "fp16-in / fp32-accumulate"}
        print(json.dumps(res))
        print("lambda      bpp       psnr     ms-ssim   frames")
        for r in rows:
            print(f"{r['lambda']:6d}  {r['bpp']:8.5f}  {r['psnr']:8.4f}  {r['msssim']:9.6f}  {r['frames']:6d}")
        if a.out_dir:
            os.makedirs(a.out_dir, exist_ok=True)
            for r in rows:
                with open(os.path.join(a.out_dir, f"{r['lambda']}.txt"), "w") as f:
                    f.write("bpp : %.6f\n\npsnr : %.6f\n\nmsssim : %.6f\n" % (r["bpp"], r["psnr"], r["msssim"]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
