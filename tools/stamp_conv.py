"""diagnostic: per-phase cycle stamps of conv_mfma_v2 (wave 0 of each workgroup)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib, ops  # noqa: E402

cin, cout, k, H, W = [int(v) for v in sys.argv[1:6]]
lib = _lib.lib()
nb = 8192
buf = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
x = ops.FM(torch.randn(1, H, W, cin, device="cuda").half())
pc = ops.pack_conv(torch.randn(cout, cin, k, k) * 0.05, torch.zeros(cout), stride=1, pad=k // 2)
y = ops.conv(x, pc, act=ops.ACT_RELU)
torch.cuda.synchronize()
for fn in (lib.tdvc_debug_set_stamp_buffer, lib.tdvc_debug_set_stamp_buffer_v3):
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn(buf.data_ptr(), nb)
ops.conv(x, pc, out=y, act=ops.ACT_RELU)
torch.cuda.synchronize()
lib.tdvc_debug_set_stamp_buffer(None, 0)
lib.tdvc_debug_set_stamp_buffer_v3(None, 0)
s = buf.cpu().numpy().reshape(nb, 8)
s = s[s[:, 0] > 0]
n = int((s[0] > 0).sum())
d = np.diff(s[:, :n].astype(np.float64), axis=1)
print("blocks", len(s), "stamps", n)
print("phase medians (cycles):", np.median(d, axis=0).astype(int).tolist())
print("phase means   (cycles):", d.mean(axis=0).astype(int).tolist())
print("block total median", int(np.median(s[:, n - 1] - s[:, 0])), "kernel span", int(s[:, n - 1].max() - s[:, 0].min()))
