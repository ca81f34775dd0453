"""diagnostic: per-wave phase stamps of conv_mfma_v11 (last stage of every workgroup's second tile + the stage before it).
Usage: stamp_v11.py cin cout H W [nres]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib, ops  # noqa: E402

cin, cout, H, W = [int(v) for v in sys.argv[1:5]]
nres = int(sys.argv[5]) if len(sys.argv) > 5 else 0
lib = _lib.lib()
nb = 4096
buf = torch.zeros(nb * 8 * 16, dtype=torch.int64, device="cuda")
x = ops.FM(torch.randn(1, H, W, cin, device="cuda").half())
pc = ops.pack_conv(torch.randn(cout, cin, 3, 3) * 0.03, torch.zeros(cout), stride=1, pad=1)
kw = dict(act=ops.ACT_LRELU, slope=0.01)
if nres > 0:
    kw["res"] = ops.FM(torch.randn(1, H, W, cout, device="cuda").half())
y = ops.conv(x, pc, **kw)
assert lib.tdvc_last_conv_kernel() == b"conv_mfma_v11", lib.tdvc_last_conv_kernel()
torch.cuda.synchronize()
fn = lib.tdvc_debug_set_stamp_buffer_v11
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
ex = lib.tdvc_debug_set_v11_experiment
ex.argtypes = [ctypes.c_int]
ex(int(os.environ.get("V11_EXPERIMENT", "0")))
fn(buf.data_ptr(), nb)
ops.conv(x, pc, out=y, **kw)
torch.cuda.synchronize()
fn(None, 0)
ex(0)
s = buf.cpu().numpy().reshape(nb, 8, 16).astype(np.float64)
s = s[s[:, 0, 0] > 0]
print("workgroups", len(s))
for w in range(8):
    t = s[:, w]
    prev = np.median(np.diff(t[:, 8:12], axis=1), axis=0).astype(int)
    last = np.median(np.diff(t[:, 0:6], axis=1), axis=0).astype(int)
    print(f"wave {w}: middle stage vmcnt={prev[0]} barrier={prev[1]} matrix={prev[2]} | last stage vmcnt={last[0]} barrier={last[1]} matrix={last[2]} "
          f"tile_barrier={last[3]} epilogue={last[4]} | middle-stage start -> last-stage start {int(np.median(t[:, 0] - t[:, 8]))}")
