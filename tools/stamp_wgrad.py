"""diagnostic: cycles per phase of conv_wgrad_kernel, per wave, summed over a workgroup's pixel blocks.
Usage: stamp_wgrad.py cin cout k N H W"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib, ops  # noqa: E402

cin, cout, k, N, H, W = [int(v) for v in sys.argv[1:7]]
lib = _lib.lib()
cap = 4096
buf = torch.zeros(cap * 4 * 8, dtype=torch.int64, device="cuda")
x = ops.FM(torch.randn(N, H, W, cin, device="cuda").half())
g = ops.FM(torch.randn(N, H, W, cout, device="cuda").half())
w = torch.randn(cout, cin, k, k).cuda() * 0.05
pc = ops.pack_conv(w, torch.zeros(cout).cuda(), stride=1, pad=k // 2)
dw = torch.zeros_like(w)
db = torch.zeros(cout, device="cuda")
for _ in range(3):
    ops.conv_wgrad(pc, g, x, dw.view(-1), db=db)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.conv_wgrad(pc, g, x, dw.view(-1), db=db)
e1.record()
torch.cuda.synchronize()
print(f"conv_wgrad + reduce: {e0.elapsed_time(e1) * 100:.1f} us per call")
fn = lib.tdvc_debug_set_stamp_buffer_wgrad
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
fn(buf.data_ptr(), cap)
ops.conv_wgrad(pc, g, x, dw.view(-1), db=db)
torch.cuda.synchronize()
fn(None, 0)
s = buf.cpu().numpy().reshape(cap, 4, 8).astype(np.float64)
s = s[s[:, 0, :6].sum(axis=1) > 0]
names = ["barrier1", "lds store", "barrier2", "prefetch issue", "contraction", "epilogue"]
print("workgroups", len(s), "cycles per workgroup (median over workgroups), per wave:")
for wv in range(4):
    print("  wave", wv, {n: int(np.median(s[:, wv, i])) for i, n in enumerate(names)}, "total", int(np.median(s[:, wv, :6].sum(axis=1))))
