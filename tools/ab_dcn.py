"""A/B of the two tdvc_dcn_fused kernels (LDS window vs L1 gather) on a 1080p map.
python tools/ab_dcn.py [iters] [sigma] [shift_dy shift_dx]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib, ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
shift = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (0.0, 0.0)
H, W = 1088, 1920
fn = _lib.lib().tdvc_debug_enable_dcn_lds
fn.argtypes = [ctypes.c_int]
fn.restype = None
torch.manual_seed(0)
x = ops.FM(torch.randn(1, H, W, 64, device="cuda").half())
om = torch.randn(1, H, W, 216, device="cuda")
om[..., :144] *= sigma
om[..., 0:144:2] += shift[0]
om[..., 1:144:2] += shift[1]
om = ops.FM(om.half())
pc = ops.pack_conv(torch.randn(64, 64, 3, 3) * 0.05, torch.zeros(64), stride=1, pad=1, ck=64)
mode = int(os.environ.get("TDVC_AB_MODE", "1"))      # 2 / 3: timing-only builds of the LDS kernel (no matrix phase / no window staging)
res = {}
for on in (mode, 0, mode, 0):
    fn(on)
    y = ops.FM.empty(1, H, W, 64, device="cuda")
    for _ in range(3):
        ops.dcn_fused(x, om, pc, y, planar=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.dcn_fused(x, om, pc, y, planar=False)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    res.setdefault(on, []).append((us, y.t.clone()))
fn(1)
same = torch.equal(res[mode][0][1], res[0][0][1]) if mode else None
print(f"dcn_fused 1080p sigma {sigma} shift {shift}: lds(mode {mode}) {[round(r[0], 1) for r in res[mode]]} us, gather {[round(r[0], 1) for r in res[0]]} us, bit-equal {same}")
