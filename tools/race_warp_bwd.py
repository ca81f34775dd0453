"""Is the tail of a SPyNet level's backward (zero mirror -> accumulate -> spynet_level_input_backward) reproducible while
another stream keeps the GPU busy with weight-gradient launches?  (tools/determinism_probe.py 'mirrors' names the mirror
of the level's up-sampled flow as the first tensor that differs between two identical training runs.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops  # noqa: E402

N, H, W = 4, 256, 256
g = torch.Generator(device="cuda").manual_seed(5)
supp = ops.FM(torch.rand(N, H, W, 4, device="cuda", generator=g))
flow_up = ops.FM(torch.randn(N, H, W, 2, device="cuda", generator=g) * 2.0)
dcat8 = ops.FM((torch.randn(N, H, W, 8, device="cuda", generator=g) * 1e-3).half())
dflow = ops.FM(torch.randn(N, H, W, 2, device="cuda", generator=g) * 1e-3)
# load for the other stream: a 7x7 weight gradient at the same size
x = ops.FM(torch.randn(N, H, W, 32, device="cuda", generator=g).half())
gy = ops.FM(torch.randn(N, H, W, 64, device="cuda", generator=g).half())
pc = ops.pack_conv(torch.randn(64, 32, 7, 7) * 0.02, torch.zeros(64), stride=1, pad=3)
dw = torch.zeros(64 * 32 * 49, device="cuda")
db = torch.zeros(64, device="cuda")
side = torch.cuda.Stream()


def tail(stage, sync=False):
    dup = ops.FM(torch.zeros_like(flow_up.t))
    dlo = ops.FM(torch.zeros(N, H // 2, W // 2, 2, device="cuda"))
    if sync:
        torch.cuda.current_stream().synchronize()
    if stage >= 1:
        ops.scale_act_res(dup, dup, res=dflow, res_sign=1.0)
    if sync:
        torch.cuda.current_stream().synchronize()
    if stage >= 2:
        ops.spynet_level_input_backward(supp, flow_up, dcat8, dup, dlo if stage >= 3 else None)
    return dup.t, dlo.t


def bits(t):
    return int(t.view(torch.int32).sum(dtype=torch.int64))


def load_side():
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        for _ in range(3):
            ops.conv_wgrad(pc, gy, x, dw, scale=1.0, db=db)


for stage, sync in ((0, False), (1, False), (2, False), (3, False), (3, True)):
    out = tail(stage, sync)
    torch.cuda.synchronize()
    ref = (bits(out[0]), bits(out[1]))
    for load in (False, True):
        bad = [0, 0]
        for it in range(30):
            if load:
                load_side()
            out = tail(stage, sync)
            torch.cuda.synchronize()
            bad[0] += bits(out[0]) != ref[0]
            bad[1] += bits(out[1]) != ref[1]
        print(f"stage {stage} (0 zeros, 1 +accumulate, 2 +warp backward, 3 +upsample backward) sync-between {sync}, side-stream load {load}: "
              f"dflow_up differs in {bad[0]}/30 runs, dflow_lo in {bad[1]}/30")
