"""diagnostic: per-wave phase stamps of conv_mfma_v10 (the instrumented stage = second stage of the second tile of every
workgroup): top-of-stage wait, barrier, the six matrix groups, tile barrier, epilogue pieces.  Usage: stamp_v10.py H W [nres]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib, ops  # noqa: E402

H, W = int(sys.argv[1]), int(sys.argv[2])
nres = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = _lib.lib()
nb = 4096
buf = torch.zeros(nb * 24, dtype=torch.int64, device="cuda")
x = ops.FM(torch.randn(1, H, W, 64, device="cuda").half())
pc = ops.pack_conv(torch.randn(64, 64, 3, 3) * 0.05, torch.zeros(64), stride=1, pad=1)
kw = dict(act=ops.ACT_RELU)
if nres > 0:
    kw["res"] = ops.FM(torch.randn(1, H, W, 64, device="cuda").half())
if nres > 1:
    kw["res2"] = ops.FM(torch.randn(1, H, W, 64, device="cuda").half())
y = ops.conv(x, pc, **kw)
torch.cuda.synchronize()
fn = lib.tdvc_debug_set_stamp_buffer_v10
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
fn(buf.data_ptr(), nb)
ops.conv(x, pc, out=y, **kw)
torch.cuda.synchronize()
fn(None, 0)
s = buf.cpu().numpy().reshape(nb // 4, 4, 24).astype(np.float64)
s = s[s[:, 0, 0] > 0]
names = ["top", "vmcnt", "barrier"] + [f"g{g}" for g in range(6)] + ["mat_end", "tile_barrier", "epilogue"]
order = [0, 1, 2, 8, 9, 10, 11, 12, 13, 3, 4, 5]
print("workgroups", len(s))
for w in range(4):
    t = s[:, w][:, order]
    d = np.diff(t, axis=1)
    print(f"wave {w}: phase durations (median cycles): " + " ".join(f"{names[i + 1]}={int(np.median(d[:, i]))}" for i in range(d.shape[1])),
          "| total", int(np.median(t[:, -1] - t[:, 0])))
