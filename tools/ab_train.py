"""A/B inside ONE process (box-to-box and run-to-run noise is larger than the effects measured here): median step time
of TrainStep with an attribute toggled.  python tools/ab_train.py attr value_a value_b
or with two statements (names in scope: step, lib, torch):  python tools/ab_train.py exec "stmt_a" "stmt_b" """
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd.model import VideoCompressor  # noqa: E402
from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402
from tdvc_amd.train import TrainStep  # noqa: E402

attr = sys.argv[1]
va, vb = (sys.argv[2], sys.argv[3]) if attr == "exec" else (eval(sys.argv[2]), eval(sys.argv[3]))
from tdvc_amd import _lib  # noqa: E402
lib = _lib.lib()
torch.manual_seed(0)
m = VideoCompressor(); fill_parameters(m); m = m.cuda().train()
xs, rs = [], []
for i in range(4):
    g = make_gop(1000 + i, 7, 256, 256).cuda()
    xs.append(g[3:4]); rs.append(ref_list([g[0:1], g[1:2], g[2:3]]))
x, refs = torch.cat(xs), torch.cat(rs)
step = TrainStep(m, loss_scale=128.0)
for _ in range(5):
    step(x, refs)
res = {repr(va): [], repr(vb): []}
for rnd in range(int(os.environ.get("AB_ROUNDS", "4"))):
    for v in (va, vb):
        if attr == "exec":
            exec(v)
        else:
            setattr(step, attr, v)
        for _ in range(3):
            step(x, refs)
        torch.cuda.synchronize()
        if os.environ.get("AB_BACK_TO_BACK", "1") == "1":          # blocks of 10 steps, no host read in between (how bench.py times them)
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(10):
                    step(x, refs)
                torch.cuda.synchronize()
                res[repr(v)].append((time.perf_counter() - t0) * 1e2)
            continue
        for _ in range(15):
            t0 = time.perf_counter()
            step(x, refs)
            torch.cuda.synchronize()
            res[repr(v)].append((time.perf_counter() - t0) * 1e3)
for k, v in res.items():
    print(f"{attr} = {k}: median {statistics.median(v):.2f} ms  min {min(v):.2f}  p90 {sorted(v)[int(0.9 * len(v))]:.2f}  (n={len(v)})")
