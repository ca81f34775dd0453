"""How much of a training step is HOST time?  Median TrainStep time at the training size (4 x 256 x 256) and at sizes whose GPU work is
4x / 64x smaller: the launch count is the same (~2.4 k kernels from one Python thread), so what remains at 1 x 64 x 64 is the host's
floor.  python tools/train_host_bound.py"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd.model import VideoCompressor  # noqa: E402
from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402
from tdvc_amd.train import TrainStep  # noqa: E402

for B, H, W in ((4, 256, 256), (1, 256, 256), (1, 64, 64)):
    torch.manual_seed(0)
    m = VideoCompressor(); fill_parameters(m); m = m.cuda().train()
    xs, rs = [], []
    for i in range(B):
        g = make_gop(1000 + i, 7, H, W).cuda()
        xs.append(g[3:4]); rs.append(ref_list([g[0:1], g[1:2], g[2:3]]))
    x, refs = torch.cat(xs), torch.cat(rs)
    step = TrainStep(m, loss_scale=128.0)
    for _ in range(6):
        step(x, refs)
    torch.cuda.synchronize()
    ts, hs = [], []
    for _ in range(25):
        t0 = time.perf_counter()
        step(x, refs)
        t1 = time.perf_counter()                 # the step's own float() reads synchronise: t1 - t0 is already the whole step
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        hs.append((t1 - t0) * 1e3)
    print(f"{B} x {H} x {W}: step median {statistics.median(ts):.2f} ms (min {min(ts):.2f})", flush=True)
    del step, m
    torch.cuda.empty_cache()
