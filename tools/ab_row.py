"""A/B of conv_row (weights in registers, row streaming) against conv_mfma_v11 (Cin = 128) / conv_mfma_v10 (Cin = 64) on layer shapes of a 1080p frame:
the two kernels interleaved in one process, random operands, outputs in rotation over fresh buffers (so that the 256 MB
Infinity Cache does not absorb the writes), HIP events on the launch stream, median of `--rounds` loops.
Usage: python tools/ab_row.py [--iters 20] [--rounds 3]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib  # noqa: E402
if os.environ.get("TDVC_LIB"):          # another build of the library (tools/build_row_variants.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["TDVC_LIB"])
from tdvc_amd import ops  # noqa: E402

SHAPES = [
    # name, cin, cout, H, W, shuffle, residual, launches per frame
    ("128->128 @544x960", 128, 128, 544, 960, False, False, 6),
    ("128->128 @544x960 +res", 128, 128, 544, 960, False, True, 6),
    ("128->256 @544x960", 128, 256, 544, 960, False, False, 2),
    ("128->512 shuffle @272x480", 128, 512, 272, 480, True, False, 4),
    ("128->128 @272x480 +res", 128, 128, 272, 480, False, True, 12),
    ("128->128 @136x240 +res", 128, 128, 136, 240, False, True, 12),
    ("128->512 shuffle @136x240", 128, 512, 136, 240, True, False, 4),
    ("64->64 @1088x1920", 64, 64, 1088, 1920, False, False, 4),
    ("64->64 @1088x1920 +res", 64, 64, 1088, 1920, False, True, 4),
    ("64->64 @544x960", 64, 64, 544, 960, False, False, 4),
    ("64->64 @272x480", 64, 64, 272, 480, False, False, 2),
    ("128->64 @1088x1920", 128, 64, 1088, 1920, False, False, 4),
    ("128->64 @544x960", 128, 64, 544, 960, False, False, 1),
]


def enable(name, on):
    fn = getattr(ops.L.lib(), "tdvc_debug_enable_" + name)
    fn.argtypes, fn.restype = [ctypes.c_int], None
    fn((15 if name == "conv_row" else 1) if on else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default="", help="substring filter on the shape names")
    ap.add_argument("--no-other", action="store_true", help="time conv_row alone")
    ap.add_argument("--zero", action="store_true", help="zero operands: the clock the chip holds without data toggling (separates power-limited time from stalls)")
    a = ap.parse_args()
    tot = {"conv_row": 0.0, "other": 0.0}
    for name, cin, cout, H, W, shuf, with_res, per_frame in SHAPES:
        if a.only and a.only not in name:
            continue
        other = "conv_mfma_v11" if cin == 128 else "conv_mfma_v10"
        rnd = torch.zeros if a.zero else torch.randn
        x = ops.FM(rnd(1, H, W, cin, device="cuda").half())
        w = rnd(cout, cin, 3, 3) * 0.03
        pc = ops.pack_conv(w, torch.randn(cout) * 0.1, stride=1, pad=1, shuffle=shuf)
        oc, oh, ow = (cout // 4, 2 * H, 2 * W) if shuf else (cout, H, W)
        nbuf = max(2, int(600e6 // (oc * oh * ow * 2)))
        ys = [ops.FM.empty(1, oh, ow, oc) for _ in range(nbuf)]
        res = ops.FM(torch.randn(1, oh, ow, oc, device="cuda").half()) if with_res else None
        kw = dict(act=ops.ACT_LRELU, slope=0.01, res=res)
        out = {}
        for rnd in range(a.rounds):
            for kern, row_on in ((("conv_row", True),) if a.no_other else (("conv_row", True), (other, False))):
                enable("conv_row", row_on)
                ops.conv(x, pc, out=ys[0], **kw)
                got = ops.L.lib().tdvc_last_conv_kernel().decode()
                assert got == kern, (got, kern)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(a.iters):
                    ops.conv(x, pc, out=ys[i % nbuf], **kw)
                e1.record()
                torch.cuda.synchronize()
                out.setdefault(kern, []).append(e0.elapsed_time(e1) / a.iters * 1e3)
        enable("conv_row", True)
        fl = 2.0 * H * W * cout * cin * 9
        med = {k: sorted(v)[len(v) // 2] for k, v in out.items()}
        if a.no_other:
            tot["conv_row"] += med["conv_row"] * per_frame
            print(f"{name:30s} conv_row {med['conv_row']:7.1f} us = {fl / med['conv_row'] / 1e6:7.1f} TFLOP/s", flush=True)
            continue
        tot["conv_row"] += med["conv_row"] * per_frame
        tot["other"] += med[other] * per_frame
        print(f"{name:30s} conv_row {med['conv_row']:7.1f} us = {fl / med['conv_row'] / 1e6:7.1f} TFLOP/s | {other} {med[other]:7.1f} us = "
              f"{fl / med[other] / 1e6:7.1f} TFLOP/s | x{med[other] / med['conv_row']:.3f}", flush=True)
    # the 3x3 stride-2 convs with 64 input channels: conv_row's space-to-depth geometry against conv_mfma_v3's
    for name, cout, H, W, per_frame in (("64->128 stride 2 @1088x1920", 128, 1088, 1920, 2), ("64->128 stride 2 @544x960", 128, 544, 960, 0)):
        if a.only and a.only not in name:
            continue
        x = ops.FM(torch.randn(1, H, W, 64, device="cuda").half())
        pc = ops.pack_conv(torch.randn(cout, 64, 3, 3) * 0.04, torch.randn(cout) * 0.1, stride=2, pad=1)
        ys = [ops.FM.empty(1, H // 2, W // 2, cout) for _ in range(4)]
        out = {}
        for rnd_ in range(a.rounds):
            for kern, row_on in (("conv_row(s2d)", True), ("conv_mfma_v3(s2d)", False)):
                enable("conv_row", row_on)
                ops.conv(x, pc, out=ys[0], act=ops.ACT_LRELU, slope=0.1)
                assert ops.L.lib().tdvc_last_conv_kernel().decode() == kern
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(a.iters):
                    ops.conv(x, pc, out=ys[i % 4], act=ops.ACT_LRELU, slope=0.1)
                e1.record()
                torch.cuda.synchronize()
                out.setdefault(kern, []).append(e0.elapsed_time(e1) / a.iters * 1e3)
        enable("conv_row", True)
        fl = 2.0 * (H // 2) * (W // 2) * cout * 64 * 9
        med = {k: sorted(v)[len(v) // 2] for k, v in out.items()}
        tot["conv_row"] += med["conv_row(s2d)"] * per_frame
        tot["other"] += med["conv_mfma_v3(s2d)"] * per_frame
        print(f"{name:30s} conv_row {med['conv_row(s2d)']:7.1f} us = {fl / med['conv_row(s2d)'] / 1e6:7.1f} TFLOP/s | conv_mfma_v3(s2d) {med['conv_mfma_v3(s2d)']:7.1f} us = "
              f"{fl / med['conv_mfma_v3(s2d)'] / 1e6:7.1f} TFLOP/s | x{med['conv_mfma_v3(s2d)'] / med['conv_row(s2d)']:.3f}", flush=True)
    print(f"per frame (launch counts of a 1080p P-frame): conv_row {tot['conv_row'] / 1e3:.3f} ms, conv_mfma_v11 / v10 {tot['other'] / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
