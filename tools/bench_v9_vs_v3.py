"""small-map layers on conv_mfma_v9 (split-K, operands from L2) vs the tiled kernels: where is the crossover?"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib, ops  # noqa: E402

lib = _lib.lib()
lim = lib.tdvc_debug_set_conv_v9_work_limit
lim.argtypes = [ctypes.c_long]
lim.restype = None
SHAPES = [(4, 128, 128, 3, 32, 32), (4, 128, 512, 3, 32, 32), (4, 512, 128, 3, 32, 32), (4, 128, 128, 3, 16, 16), (4, 128, 512, 3, 16, 16),
          (4, 192, 256, 3, 16, 16), (4, 192, 768, 3, 8, 8), (1, 128, 128, 3, 68, 120), (1, 128, 256, 3, 68, 120), (1, 128, 512, 3, 68, 120),
          (1, 192, 768, 3, 34, 60), (1, 128, 512, 3, 34, 60), (4, 128, 256, 3, 32, 32), (4, 256, 128, 3, 32, 32), (1, 128, 192, 3, 68, 120)]


def t(x, pc, y):
    for _ in range(3):
        ops.conv(x, pc, out=y, act=ops.ACT_RELU)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv(x, pc, out=y, act=ops.ACT_RELU)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3, lib.tdvc_last_conv_kernel().decode()


for N, cin, cout, k, H, W in SHAPES:
    x = ops.FM(torch.randn(N, H, W, cin, device="cuda").half())
    pc = ops.pack_conv(torch.randn(cout, cin, k, k) * 0.05, torch.zeros(cout), stride=1, pad=k // 2)
    y = ops.conv(x, pc, act=ops.ACT_RELU)
    lim(1 << 40)
    a, ka = t(x, pc, y)
    lim(0)
    b, kb = t(x, pc, y)
    lim(1 << 20)
    c, kc = t(x, pc, y)
    print(f"{N}x{H}x{W} {k}x{k} {cin}->{cout}: px*cout {N * H * W * cout / 2**20:5.2f} M   {ka} {a:6.1f} us   {kb} {b:6.1f} us   default -> {kc}")
