"""fused DCN forward at 1088x1920 (8 groups, offsets ~N(0, sigma px)): NHWC gather vs the group-planar gather (incl. its
re-layout pass), HIP events on the launch stream.  Usage: python tools/bench_dcn.py [sigma ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops  # noqa: E402

H, W = 1088, 1920
x = ops.FM(torch.randn(1, H, W, 64, device="cuda").half())
y = ops.FM.empty(1, H, W, 64)
pc = ops.pack_conv(torch.randn(64, 64, 3, 3) * 0.05, torch.zeros(64), stride=1, pad=1, ck=64)
for sigma in [float(v) for v in sys.argv[1:]] or [0.5, 1.5, 4.0]:
    om = ops.FM((torch.randn(1, H, W, 216, device="cuda") * sigma).half())
    for planar in (False, True, False, True):
        ops.dcn_fused(x, om, pc, y, groups=8, act=ops.ACT_LRELU, slope=0.1, planar=planar)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.dcn_fused(x, om, pc, y, groups=8, act=ops.ACT_LRELU, slope=0.1, planar=planar)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"sigma {sigma:4.1f} px  planar={int(planar)}  {ms * 1e3:8.1f} us  {688.0 * H * W / ms / 1e6:7.1f} GB/s algorithmic", flush=True)
