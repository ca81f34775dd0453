"""Kernel-level timing of representative hot-path layers at 1080p (HIP events on the launch stream).
Usage: python tools/bench_conv.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops  # noqa: E402

LAYERS = [
    # name, cin, cout, k, stride, H, W
    ("3x3 64->64 s1 @1088x1920", 64, 64, 3, 1, 1088, 1920),
    ("3x3 128->64 s1 @1088x1920", 128, 64, 3, 1, 1088, 1920),
    ("1x1 256->64 s1 @1088x1920", 256, 64, 1, 1, 1088, 1920),
    ("3x3 8->64 s1 @1088x1920", 8, 64, 3, 1, 1088, 1920),
    ("3x3 64->216 s1 @1088x1920", 64, 216, 3, 1, 1088, 1920),
    ("3x3 64->128 s2 @1088x1920", 64, 128, 3, 2, 1088, 1920),
    ("3x3 128->128 s1 @544x960", 128, 128, 3, 1, 544, 960),
    ("3x3 128->512 s1 @272x480", 128, 512, 3, 1, 272, 480),
    ("7x7 32->64 s1 @1088x1920", 32, 64, 7, 1, 1088, 1920),
    ("7x7 8->32 s1 @1088x1920", 8, 32, 7, 1, 1088, 1920),
    ("7x7 64->32 s1 @1088x1920", 64, 32, 7, 1, 1088, 1920),
    ("7x7 32->16 s1 @1088x1920", 32, 16, 7, 1, 1088, 1920),
    ("3x3 128->128 s1 @68x120", 128, 128, 3, 1, 68, 120),
    ("3x3 128->128 s1 @34x60", 128, 128, 3, 1, 34, 60),
    ("3x3 128->128 s1 @17x30", 128, 128, 3, 1, 17, 30),
    ("3x3 192->768 s1 @34x60", 192, 768, 3, 1, 34, 60),
    ("1x1 512->426 s1 @68x120", 512, 426, 1, 1, 68, 120),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="", help="substring filter on the layer names")
    a = ap.parse_args()
    if os.environ.get("V11_EXPERIMENT"):
        import ctypes
        ex = ops.L.lib().tdvc_debug_set_v11_experiment
        ex.argtypes = [ctypes.c_int]
        ex(int(os.environ["V11_EXPERIMENT"]))
    for name, cin, cout, k, s, H, W in LAYERS:
        if a.only and a.only not in name:
            continue
        zero = os.environ.get("TDVC_BENCH_ZERO") == "1"      # zero operands: separates power-limited clocks from stalls
        x = ops.FM((torch.zeros if zero else torch.randn)(1, H, W, cin, device="cuda").half())
        w = (torch.zeros if zero else torch.randn)(cout, cin, k, k) * 0.05
        pc = ops.pack_conv(w, torch.zeros(cout), stride=s, pad=k // 2)
        y = ops.conv(x, pc, act=ops.ACT_RELU)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.conv(x, pc, out=y, act=ops.ACT_RELU)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        Ho, Wo = y.H, y.W
        fl = 2.0 * Ho * Wo * cout * cin * k * k
        by = 2.0 * (H * W * cin + Ho * Wo * cout)
        kern = ops.L.lib().tdvc_last_conv_kernel().decode()
        print(f"{name:32s} {kern:16s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TFLOP/s  {by/ms/1e6:8.1f} GB/s (algorithmic)  ck={pc.ck}", flush=True)
    if a.only:
        return
    # fused DCN at 1080p
    from tdvc_amd.model.modules import DCN
    m = DCN(64, 64, 3, 1, 1, deformable_groups=8).cuda()
    torch.nn.init.normal_(m.weight, std=0.05)
    torch.nn.init.normal_(m.conv_offset_mask.weight, std=0.02)
    H, W = 1088, 1920
    x = ops.FM(torch.randn(1, H, W, 64, device="cuda").half())
    yy = ops.FM(torch.randn(1, H, W, 64, device="cuda").half())
    out = ops.FM.empty(1, H, W, 64)
    om = ops.conv(yy, ops.pack_conv(m.conv_offset_mask.weight, m.conv_offset_mask.bias, stride=1, pad=1))
    pc = ops.pack_conv(m.weight, m.bias, stride=1, pad=1, ck=64)
    ops.dcn_fused(x, om, pc, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        ops.dcn_fused(x, om, pc, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    by = H * W * (64 + 216 + 64) * 2.0
    print(f"{'fused DCN 64->64 g8 @1088x1920':32s} {ms*1e3:9.1f} us  {2.0*H*W*64*576/ms/1e9:8.1f} TFLOP/s  {by/ms/1e6:8.1f} GB/s (algorithmic)", flush=True)


if __name__ == "__main__":
    main()
