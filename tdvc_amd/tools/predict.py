#!/usr/bin/env python
"""Counterpart of the reference's tools/predict.py GOP loop on the MI355X path.

The reference script needs CUDA at import, cv2/natsort datasets under /dataset/... and a
checkpoint that does not ship; this driver keeps its behaviour — yaml keys
`model, pretrain, val_dataset, class, enable_amp`, the reference-list rule (:55-62), pad/crop to 64
(:51-53,69-70), PSNR / bpp averaging over all frames (:87-108) — on synthetic GOPs (or a checkpoint
given by --pretrain), one process per GPU with GOPs sharded round-robin across ranks.

  python -m tdvc_amd.tools.predict --gops 2 --height 1080 --width 1920
  python -m tdvc_amd.tools.predict --gops 1 --height 128 --width 192 --bitstream-dir /tmp/tdvc_bits     # real bitstreams
  python -m torch.distributed.run --nproc-per-node 8 -m tdvc_amd.tools.predict --gops 16
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch
import yaml

from .. import metrics
from ..codec_utils import crop, pad, psnr
from ..model import VideoCompressor
from ..parallel import gather_frame_stats, shard_gops
from ..synth import fill_parameters, make_gop, ref_list


def _msssim(rc: torch.Tensor, xc: torch.Tensor) -> float:
    """predict.py:92-94: ms_ssim(recon.float(), input, data_range=1.0); the 5-level pyramid needs >= 176 pixels a side"""
    if min(rc.shape[-2:]) < 176:
        return float("nan")
    return float(metrics.ms_ssim(rc.float(), xc.float(), data_range=1.0))


def code_gop(net, frames: torch.Tensor, enable_amp: bool = True, bitstream_dir: str | None = None, tag: str = ""):
    """frames: (T,3,h,w) on the GPU, frame 0 = I-frame reconstruction. Returns per-P-frame stats.
    With `bitstream_dir` every frame is really coded (`VideoCompressor.encode`), written as a container file in the
    record layout of tools/utils/encoder.py:61-68, read back and decoded; the decoder's frame must equal the encoder's."""
    h, w = frames.shape[-2:]
    refs = [pad(frames[0:1], 64)]
    stats = []
    for t in range(1, frames.shape[0]):
        x = pad(frames[t:t + 1], 64)
        if bitstream_dir:
            from .. import bitstream
            rl = ref_list(refs)
            torch.cuda.synchronize()
            t_e = time.time()
            enc = net.encode(x, rl)
            torch.cuda.synchronize()
            t_e = time.time() - t_e
            flat = [s[0] for s in enc["strings"]]
            # first shape word: the reference writes the batch index (always 0); the y records of a wavefront-ordered
            # stream carry 1 there so that a reader cannot mistake them for compressai's raster order
            wf = int(net.stream_order == "wavefront")
            shp = [(wf if i % 2 == 0 else 0, 128, *enc["shapes"][i // 2]) for i in range(4)]
            path = os.path.join(bitstream_dir, f"{tag}frame{t:03d}.bin")
            with open(path, "wb") as f:
                nbytes = bitstream.write_records(f, flat, shp)
            with open(path, "rb") as f:
                strings, shapes = bitstream.read_records(f, 4)
            if shapes[0][0] != wf or shapes[2][0] != wf:
                raise RuntimeError(f"{path}: stream order flag {shapes[0][0]} does not match the decoder's ({net.stream_order})")
            t_d = time.time()
            recon = net.decode([[s] for s in strings], [shapes[0][2:], shapes[2][2:]], rl)
            torch.cuda.synchronize()
            t_d = time.time() - t_d
            assert torch.equal(recon, enc["recon"]), "decoder / encoder reconstruction mismatch"
            refs.append(recon)
            rc, xc = crop(recon, (h, w)), crop(x, (h, w))
            bpp = 8.0 * nbytes / (x.shape[-2] * x.shape[-1])
            stats.append({"frame": t, "psnr": psnr(rc, xc), "msssim": _msssim(rc, xc), "bpp": bpp, "bpp_mv": 8.0 * (len(flat[0]) + len(flat[1])) / (x.shape[-2] * x.shape[-1]),
                          "bpp_res": 8.0 * (len(flat[2]) + len(flat[3])) / (x.shape[-2] * x.shape[-1]), "bytes": nbytes,
                          "encode_s": t_e, "decode_s": t_d})
            continue
        recon, bpp_res, bpp_mv = net(x, ref_list(refs), enable_amp)
        refs.append(recon)                                   # padded reconstruction re-enters the list (:68)
        rc, xc = crop(recon, (h, w)), crop(x, (h, w))
        stats.append({"frame": t, "psnr": psnr(rc, xc), "msssim": _msssim(rc, xc), "bpp": float(bpp_res + bpp_mv),
                      "bpp_mv": float(bpp_mv), "bpp_res": float(bpp_res)})
    return stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default=None, help="yaml with the reference's predict keys")
    ap.add_argument("--pretrain", default=None, help="state-dict checkpoint (reference key names); default: synthetic filler")
    ap.add_argument("--gops", type=int, default=2)
    ap.add_argument("--gop-size", type=int, default=7)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--out", default=None)
    ap.add_argument("--bitstream-dir", default=None, help="really encode / write / read / decode every frame into this directory")
    ap.add_argument("--dataset-root", default=None, help="ori_img/ + compress_img_bpg/ tree (tdvc_amd.data); default: synthetic GOPs")
    ap.add_argument("--val-dataset", default="UVG", choices=("UVG", "MCL-JCV", "HEVC"), help="with --dataset-root (predict.py:154-166)")
    ap.add_argument("--cls", default="B", help="HEVC class A..E")
    ap.add_argument("--train-lambda", type=int, default=2048, help="selects the BPG QP of the I-frames (dataset.py:25-36)")
    ap.add_argument("--stream-order", default="raster", choices=("raster", "wavefront"),
                    help="with --bitstream-dir: y-symbol order (raster = the reference's; wavefront = diagonal-parallel decoding)")
    ap.add_argument("--coder-fp32", action="store_true", help="both coders as fp32 islands (the reference's precision), also with enable_amp: True")
    a = ap.parse_args()
    opt = {"model": "pnet", "pretrain": a.pretrain, "val_dataset": "synthetic", "class": "-", "enable_amp": True}
    if a.cfg:
        opt.update(yaml.safe_load(open(a.cfg)))
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(dev)
    net = VideoCompressor()
    if opt["pretrain"] and os.path.exists(str(opt["pretrain"])):
        net.load_state_dict(torch.load(opt["pretrain"], map_location="cpu"), strict=True)
    else:
        fill_parameters(net)
    net = net.to(dev).eval()
    # `coder_fp32: true` (an extra yaml key, or --coder-fp32): both coders as the reference's fp32 islands (pnet.py:33,57) whatever
    # `enable_amp` says -- the reference-faithful precision for a reference checkpoint + cfg/predict.yaml (`enable_amp: True`), which
    # otherwise selects this build's fp16-in / fp32-accumulate coders
    net.coder_fp32 = bool(opt.get("coder_fp32", False)) or a.coder_fp32
    net.stream_order = a.stream_order
    t0 = time.time()
    stats = []
    dataset = None
    if a.dataset_root:                                   # predict.py:154-166: GOP 12 (UVG, MCL-JCV) / 10 (HEVC), testfull
        from .. import data
        opt.update(val_dataset=a.val_dataset, **{"class": a.cls if a.val_dataset == "HEVC" else "-"})
        dataset = (data.HEVCDataSet(a.dataset_root, a.train_lambda, 10, a.cls, testfull=True, isTrain=False) if a.val_dataset == "HEVC" else
                   data.UVGDataSet(a.dataset_root, a.train_lambda, 12, testfull=True, isTrain=False))
    with torch.no_grad():
        for g in shard_gops(len(dataset) if dataset is not None else a.gops, world, rank):
            if dataset is not None:
                inp, ref_image, ref_bpp, ref_psnr, ref_msssim = dataset[g][:5]
                frames = torch.cat([torch.from_numpy(ref_image)[None], torch.from_numpy(inp)]).to(dev)
                # the BPG I-frame enters the averages with its own bpp / PSNR / MS-SSIM (predict.py:46-49)
                stats.append({"gop": g, "frame": 0, "psnr": float(ref_psnr), "msssim": float(ref_msssim), "bpp": float(ref_bpp),
                              "bpp_mv": 0.0, "bpp_res": 0.0})
            else:
                frames = make_gop(2000 + g, a.gop_size, a.height, a.width).to(dev)
            if a.bitstream_dir:
                os.makedirs(a.bitstream_dir, exist_ok=True)
            for s in code_gop(net, frames, bool(opt["enable_amp"]), a.bitstream_dir, f"gop{g:03d}_"):
                s["gop"] = g
                stats.append(s)
    torch.cuda.synchronize()
    allstats = gather_frame_stats(stats)
    if rank == 0:
        n = max(1, len(allstats))
        res = {"frames": len(allstats), "bpp": sum(s["bpp"] for s in allstats) / n,
               "psnr": sum(s["psnr"] for s in allstats) / n, "msssim": sum(s["msssim"] for s in allstats) / n, "seconds": time.time() - t0, "cfg": opt}
        timed = [s for s in allstats if "encode_s" in s]
        if timed:
            res.update(stream_order=a.stream_order, encode_s_per_frame=sum(s["encode_s"] for s in timed) / len(timed),
                       decode_s_per_frame=sum(s["decode_s"] for s in timed) / len(timed),
                       encode_s=[round(s["encode_s"], 4) for s in timed], decode_s=[round(s["decode_s"], 4) for s in timed])
        print(json.dumps(res))
        if a.out:
            with open(a.out, "w") as f:
                f.write("bpp : %.6f\n\npsnr : %.6f\n\nmsssim : %.6f\n" % (res["bpp"], res["psnr"], res["msssim"]))       # predict.py:102
                f.write("cfg :\n" + json.dumps(opt, indent=4) + "\ncost_time :\n" + str(res["seconds"]) + "\n")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
