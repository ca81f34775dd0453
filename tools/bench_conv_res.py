"""3x3 64->64 @1088x1920 with 0 / 1 / 2 fp16 residuals (fresh output buffers in rotation, so the writes are not
absorbed by the Infinity Cache): conv_mfma_v10 vs conv_mfma_v7 (in-process A/B through the debug switch)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib, ops  # noqa: E402

H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 1088), 1920
NBUF = int(sys.argv[2]) if len(sys.argv) > 2 else 4          # output buffers in rotation (1: the same buffer every launch)
lib = _lib.lib()
sw = lib.tdvc_debug_enable_conv_v10
sw.argtypes, sw.restype = [ctypes.c_int], None
x = ops.FM(torch.randn(1, H, W, 64, device="cuda").half())
rs = [ops.FM(torch.randn(1, H, W, 64, device="cuda").half()) for _ in range(2)]
ys = [ops.FM.empty(1, H, W, 64) for _ in range(NBUF)]
pc = ops.pack_conv(torch.randn(64, 64, 3, 3) * 0.05, torch.zeros(64), stride=1, pad=1)
for nres in (0, 1, 2):
    kw = dict(act=ops.ACT_RELU)
    if nres > 0:
        kw["res"] = rs[0]
    if nres > 1:
        kw["res2"] = rs[1]
    for v10 in (1, 0, 1):
        sw(v10)
        ops.conv(x, pc, out=ys[0], **kw)
        name = lib.tdvc_last_conv_kernel().decode()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(24):
            ops.conv(x, pc, out=ys[i % NBUF], **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 24
        by = 2.0 * H * W * 64 * (2 + nres)
        print(f"H={H} nbuf={NBUF} nres={nres} {name:14s} {ms * 1e3:7.1f} us ({ms * 1e3 * 1088 / H:6.1f} per 1088 rows)  {2.0 * H * W * 64 * 576 / ms / 1e9:7.1f} TFLOP/s  {by / ms / 1e6:7.1f} GB/s algorithmic", flush=True)
sw(1)
