"""run the fused DCN forward repeatedly on a 1080p map with bench-like offsets (|.| ~ 1, groups uncorrelated): for
rocprofv3 --pmc.  python tools/one_dcn.py iters [sigma]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 1.3
H, W, G = 1088, 1920, 8
torch.manual_seed(0)
x = ops.FM(torch.randn(1, H, W, 64, device="cuda").half())
om = torch.randn(1, H, W, 216, device="cuda")
om[..., :144] *= sigma
om = ops.FM(om.half())
pc = ops.pack_conv(torch.randn(64, 64, 3, 3) * 0.05, torch.zeros(64), stride=1, pad=1)
y = ops.FM.empty(1, H, W, 64, device="cuda")
for _ in range(2):
    ops.dcn_fused(x, om, pc, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.dcn_fused(x, om, pc, y)
e1.record()
torch.cuda.synchronize()
print(f"dcn_fused 1080p sigma {sigma}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us")
