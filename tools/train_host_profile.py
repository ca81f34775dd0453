"""Where does the HOST time of a training step go?  tools/train_host_bound.py shows the step takes the same ~34.5 ms at 4 x 256 x 256 and at
1 x 64 x 64: it is bound by the ~2.4 k launches one Python thread issues, not by the GPU.  cProfile of 10 steps at 1 x 64 x 64 (GPU work
negligible), top functions by own time and by cumulative time.  python tools/train_host_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd.model import VideoCompressor  # noqa: E402
from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402
from tdvc_amd.train import TrainStep  # noqa: E402

torch.manual_seed(0)
m = VideoCompressor(); fill_parameters(m); m = m.cuda().train()
g = make_gop(1000, 7, 64, 64).cuda()
x, refs = g[3:4], ref_list([g[0:1], g[1:2], g[2:3]])
step = TrainStep(m, loss_scale=128.0)
for _ in range(6):
    step(x, refs)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step(x, refs)
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumtime"):
    print(f"===== top 45 by {key} (10 steps)")
    pstats.Stats(pr).strip_dirs().sort_stats(key).print_stats(45)
