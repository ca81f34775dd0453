"""Are two identical TrainStep runs bitwise identical?  Runs the tests' operating-point training recipe twice from the same
seed for a few steps and lists the parameters whose first-step gradients / final values differ.
python tools/determinism_probe.py [steps] [side_stream 0|1] [pool 0|1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops  # noqa: E402
from tdvc_amd.model import VideoCompressor  # noqa: E402
from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402
from tdvc_amd.train import TrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
side = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
use_pool = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True


def batches(n):
    pool, cursor, out = [], 0, []
    while len(out) < n:
        while len(pool) < 4:
            gop = make_gop(5000 + cursor, 7, 256, 256)
            cursor += 1
            for t in range(1, 7):
                pool.append((gop[t:t + 1], ref_list([gop[k:k + 1] for k in range(0, t)][-4:] if t > 3 else [gop[k:k + 1] for k in range(0, t)])))
        b, pool = pool[:4], pool[4:]
        out.append((torch.cat([q[0] for q in b]).cuda(), torch.cat([q[1] for q in b]).cuda()))
    return out


def run(data):
    torch.manual_seed(1111)
    ops.DETERMINISTIC = True
    net = VideoCompressor()
    fill_parameters(net)
    net = net.cuda().train()
    step = TrainStep(net, train_lambda=256.0, lr=2e-4, loss_scale=128.0, side_stream=side)
    step.use_pool = use_pool
    first = None
    for it, (x, refs) in enumerate(data):
        log = step(x, refs)
        if it == 0:
            torch.cuda.synchronize()
            first = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()
    return first, {k: v.detach().clone() for k, v in net.state_dict().items()}, log


data = batches(steps)
ga, sa, la = run(data)
gb, sb, lb = run(data)
bad_g = [n for n in ga if not torch.equal(ga[n], gb[n])]
bad_s = [n for n in sa if not torch.equal(sa[n], sb[n])]
print(f"side_stream={side} pool={use_pool} steps={steps}: first-step gradients differ in {len(bad_g)}/{len(ga)} tensors; "
      f"final state differs in {len(bad_s)}/{len(sa)} tensors; logs equal: {la == lb}")
for n in bad_g[:60]:
    d = (ga[n].float() - gb[n].float()).abs()
    print(f"   grad {n}: {int((d > 0).sum())}/{d.numel()} elements, max |d| {float(d.max()):.3e} (max |g| {float(ga[n].float().abs().max()):.3e})")
if not bad_g:
    for n in bad_s[:30]:
        print("   state", n)
