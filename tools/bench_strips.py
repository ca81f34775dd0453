"""Does running a conv chain strip-by-strip keep intermediates in the 256 MB Infinity Cache?
chain of K 3x3 64->64 convs at 1088x1920: whole-frame layer by layer vs S horizontal strips."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops  # noqa: E402

H, W, C, K = 1088, 1920, 64, 6
bufs = [ops.FM(torch.randn(1, H, W, C, device="cuda").half()) for _ in range(3)]
pcs = [ops.pack_conv(torch.randn(C, C, 3, 3) * 0.04, torch.zeros(C), stride=1, pad=1) for _ in range(K)]


def rows(fm, a, b):
    v = ops.FM(fm.t[:, a:b])
    return v


def chain_full():
    x = bufs[0]
    for i in range(K):
        y = bufs[1 + (i & 1)]
        ops.conv(x, pcs[i], out=y, act=ops.ACT_RELU)
        x = y


def chain_strips(S):
    step = (H + S - 1) // S
    for s in range(S):
        r0, r1 = s * step, min(H, (s + 1) * step)
        x = bufs[0]
        a_prev, b_prev = max(0, r0 - K), min(H, r1 + K)
        for i in range(K):
            a, b = max(0, r0 - (K - 1 - i)), min(H, r1 + (K - 1 - i))
            y = bufs[1 + (i & 1)]
            # input view rows [a_prev, b_prev) ; output rows [a, b): the conv pads zeros at the VIEW edge,
            # so compute the whole input view's rows and keep only [a, b)
            xin = rows(x, a_prev, b_prev)
            yout = rows(y, a_prev, b_prev)
            ops.conv(xin, pcs[i], out=yout, act=ops.ACT_RELU)
            x = y
            a_prev, b_prev = a, b


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t = timeit(chain_full)
print(f"full frame   : {t:7.3f} ms for {K} convs = {t/K*1e3:7.1f} us/conv", flush=True)
for S in (2, 4, 8, 16):
    t = timeit(lambda: chain_strips(S))
    print(f"{S:2d} strips    : {t:7.3f} ms for {K} convs = {t/K*1e3:7.1f} us/conv", flush=True)
