"""Back-to-back training steps (no host read of the log, the way bench.py --mode train times them) against isolated ones
(tools/train_host_bound.py synchronises after every step): per-step HOST time of 30 consecutive steps, the wall time of the
whole run, and the same with the garbage collector off.  python tools/train_back_to_back.py [lagged|exact]"""
import gc
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd.model import VideoCompressor  # noqa: E402
from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402
from tdvc_amd.train import TrainStep  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "exact"
torch.manual_seed(int(os.environ.get("SEED", "0")))
m = VideoCompressor(); fill_parameters(m); m = m.cuda().train()
xs, rs = [], []
for i in range(4):
    g = make_gop(1000 + i, 7, 256, 256).cuda()
    xs.append(g[3:4]); rs.append(ref_list([g[0:1], g[1:2], g[2:3]]))
x, refs = torch.cat(xs), torch.cat(rs)
step = TrainStep(m, train_lambda=2048.0, lr=1e-4, loss_scale=128.0, scale_update=mode)
for _ in range(6):
    step(x, refs)
torch.cuda.synchronize()
for label, use_gc in (("gc on", True), ("gc off", False), ("gc on", True)):
    gc.collect()
    (gc.enable if use_gc else gc.disable)()
    torch.cuda.synchronize()
    hs, t0 = [], time.perf_counter()
    for _ in range(30):
        a = time.perf_counter()
        step(x, refs)
        hs.append((time.perf_counter() - a) * 1e3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    gc.enable()
    print(f"{mode}, {label}: {1e3 * (t2 - t0) / 30:.2f} ms/step wall; host per step median {statistics.median(hs):.2f} max {max(hs):.2f} "
          f"sum {sum(hs):.1f} ms; final drain {1e3 * (t2 - t1):.2f} ms; slowest five {sorted(round(h, 1) for h in hs)[-5:]}", flush=True)
