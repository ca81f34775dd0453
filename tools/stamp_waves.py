"""diagnostic: per-wave phase stamps of conv_mfma_v7 (stage 3 of every workgroup), relative to the
workgroup's earliest stamp.  Usage: stamp_waves.py cin cout H W"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib, ops  # noqa: E402

cin, cout, H, W = [int(v) for v in sys.argv[1:5]]
lib = _lib.lib()
nb = 8192
buf = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
x = ops.FM(torch.randn(1, H, W, cin, device="cuda").half())
pc = ops.pack_conv(torch.randn(cout, cin, 3, 3) * 0.05, torch.zeros(cout), stride=1, pad=1)
y = ops.conv(x, pc, act=ops.ACT_RELU)
torch.cuda.synchronize()
fn = getattr(lib, "tdvc_debug_set_stamp_buffer_" + os.environ.get("STAMP_KERNEL", "v7"))
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
fn(buf.data_ptr(), nb)
ops.conv(x, pc, out=y, act=ops.ACT_RELU)
torch.cuda.synchronize()
fn(None, 0)
s = buf.cpu().numpy().reshape(nb // 8, 8, 8).astype(np.float64)
s = s[s[:, 0, 0] > 0]
rel = s[:, :, :6] - s[:, :, :1].min(axis=1, keepdims=True)
print("blocks", len(s))
for w in range(8):
    print("wave", w, "stamps of the instrumented stage, relative to the workgroup's first:", np.median(rel[:, w], axis=0).astype(int).tolist())
