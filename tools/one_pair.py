"""run ONE fused conv pair repeatedly (for rocprofv3 --pmc): python tools/one_pair.py H W iters"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops  # noqa: E402

H, W, iters = [int(v) for v in sys.argv[1:4]]
x = ops.FM(torch.randn(1, H, W, 64, device="cuda").half())
pp = ops.pack_conv_pair(torch.randn(64, 64, 3, 3, device="cuda") * 0.05, torch.zeros(64, device="cuda"),
                        torch.randn(64, 64, 3, 3, device="cuda") * 0.05, torch.zeros(64, device="cuda"))
y = ops.conv_pair(x, pp)
for _ in range(iters):
    ops.conv_pair(x, pp, out=y)
torch.cuda.synchronize()
