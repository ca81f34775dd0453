"""per-step cycle breakdown of conv_pair (steady-state steps): python tools/stamp_pair.py [H W]"""
import os
import ctypes, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops

H, W = (int(v) for v in (sys.argv[1:3] + ["1088", "1920"][len(sys.argv) - 1:]))
lib = ops.L.lib()
fn = lib.tdvc_debug_set_stamp_buffer_pair
fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], None
ex = lib.tdvc_debug_set_pair_experiment
ex.argtypes, ex.restype = [ctypes.c_int], None
w1, w2 = torch.randn(64, 64, 3, 3, device="cuda") * 0.05, torch.randn(64, 64, 3, 3, device="cuda") * 0.05
b1 = torch.randn(64, device="cuda") * 0.1
x = ops.FM(torch.randn(1, H, W, 64, device="cuda").half())
y = ops.FM.empty(1, H, W, 64)
pp = ops.pack_conv_pair(w1, b1, w2, b1)
for _ in range(5):
    ops.conv_pair(x, pp, out=y)
torch.cuda.synchronize()
buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
for e in (0, 1, 2, 4, 7):
    ex(e)
    fn(buf.data_ptr(), 256)
    for _ in range(5):
        ops.conv_pair(x, pp, out=y)
    torch.cuda.synchronize()
    fn(None, 0)
    r = buf.view(256, 8, 8).double().cpu()
    n = r[:, :, 4].clamp(min=1)
    per = lambda j: (r[:, :, j] / n)
    tot = per(0) + per(1) + per(2)
    print(f"experiment {e}: steps/wave {float(n.mean()):.0f}; cycles per step: total {float(tot.mean()):.0f}; per wave (0-3 conv1, 4-7 conv2) "
          f"busy {[round(float(per(0)[:, w].mean())) for w in range(8)]} vmcnt wait {[round(float(per(1)[:, w].mean())) for w in range(8)]} "
          f"barrier {[round(float(per(2)[:, w].mean())) for w in range(8)]}", flush=True)
ex(0)
