"""diagnostic: per-launch table of the conv kernels of one 1080p P-frame (HIP events on the launch stream)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tdvc_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
runner = bench.GopRunner(model, bench.make_inputs(2000, dev))
for _ in range(2):
    runner.step()
torch.cuda.synchronize()
ops.PROFILE = []
runner.step()
torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
agg = {}
for r in prof:
    k = (r["kernel"], r["shape"])
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += r["e0"].elapsed_time(r["e1"])
    a[2] += r["flops_real"]
for (k, sh), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms:7.3f} ms  n={n:3d}  {ms / n * 1e3:7.1f} us/launch  {fl / ms / 1e9:7.1f} TF  {k:18s} {sh}")
print("total", sum(v[1] for v in agg.values()))
