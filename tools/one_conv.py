"""run ONE layer repeatedly (for rocprofv3 --pmc): python tools/one_conv.py cin cout k stride H W iters"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops  # noqa: E402

cin, cout, k, s, H, W, iters = [int(v) for v in sys.argv[1:8]]
x = ops.FM(torch.randn(1, H, W, cin, device="cuda").half())
pc = ops.pack_conv(torch.randn(cout, cin, k, k) * 0.05, torch.zeros(cout), stride=s, pad=k // 2)
y = ops.conv(x, pc, act=ops.ACT_RELU)
for _ in range(iters):
    ops.conv(x, pc, out=y, act=ops.ACT_RELU)
torch.cuda.synchronize()
