"""conv_wgrad + reduce time against the cap on the partial-sum workspace (more workers = more workgroups per CU, more bytes to
reduce) for the training step's most frequent weight-gradient shapes."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib, ops  # noqa: E402

lib = _lib.lib()
setcap = lib.tdvc_debug_set_wgrad_partial_cap_mb
setcap.argtypes, setcap.restype = [ctypes.c_int], None
setw = lib.tdvc_debug_set_wgrad_max_workers
setw.argtypes, setw.restype = [ctypes.c_int], None
shapes = [(64, 64, 3, 4, 256, 256), (128, 128, 3, 4, 128, 128), (64, 32, 7, 4, 256, 256), (128, 128, 3, 4, 64, 64), (64, 64, 3, 4, 128, 128)]
for cin, cout, k, N, H, W in shapes:
    x = ops.FM(torch.randn(N, H, W, cin, device="cuda").half())
    g = ops.FM(torch.randn(N, H, W, cout, device="cuda").half())
    w = torch.randn(cout, cin, k, k).cuda() * 0.05
    pc = ops.pack_conv(w, torch.zeros(cout).cuda(), stride=1, pad=k // 2)
    dw, db = torch.zeros_like(w), torch.zeros(cout, device="cuda")
    row = []
    for cap, mw in ((4, 256), (8, 256), (16, 256), (32, 256), (32, 512), (64, 512)):
        setcap(cap)
        setw(mw)
        for _ in range(3):
            ops.conv_wgrad(pc, g, x, dw.view(-1), db=db)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv_wgrad(pc, g, x, dw.view(-1), db=db)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        row.append(f"cap {cap} MB / {mw} workers: {best:6.1f} us")
    print(f"{k}x{k} {cin}->{cout} @{N}x{H}x{W}: " + "; ".join(row), flush=True)
setcap(16)
setw(256)
