"""A/B of conv_mfma_v11 variants on one layer, interleaved rounds in one process: python3 tools/ab_v11.py [cin cout H W]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib, ops  # noqa: E402

cin, cout, H, W = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (128, 128, 544, 960)
lib = _lib.lib()
early = lib.tdvc_debug_set_v11_dma_early
early.argtypes = [ctypes.c_int]
x = ops.FM(torch.randn(1, H, W, cin, device="cuda").half())
pc = ops.pack_conv(torch.randn(cout, cin, 3, 3) * 0.03, torch.zeros(cout), stride=1, pad=1)
y = ops.conv(x, pc, act=ops.ACT_RELU)
assert lib.tdvc_last_conv_kernel() == b"conv_mfma_v11"
ref = y.t.clone()


def loop(n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.conv(x, pc, out=y, act=ops.ACT_RELU)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = {0: [], 1: []}
for rnd in range(8):
    for v in (0, 1):
        early(v)
        loop(5)
        res[v].append(loop())
        assert torch.equal(y.t, ref), f"variant {v} changed the result"
early(0)
for v in (0, 1):
    r = sorted(res[v])
    print(f"dma_early={v}: median {r[len(r) // 2]:.1f} us  min {r[0]:.1f}  max {r[-1]:.1f}  ({2.0 * H * W * cin * cout * 9 / r[len(r) // 2] / 1e6:.0f} TFLOP/s)")
