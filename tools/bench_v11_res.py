"""3x3 128->128 @544x960 (conv_mfma_v11) with 0 / 1 fp16 residuals, outputs in rotation.  TDVC_LIB=path runs another build of
the library (A/B of two builds: one process each, run back to back on the same box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib  # noqa: E402

if os.environ.get("TDVC_LIB"):
    _lib.LIB_PATH = os.environ["TDVC_LIB"]
from tdvc_amd import ops  # noqa: E402

H, W, C = 544, 960, 128
x = ops.FM(torch.randn(1, H, W, C, device="cuda").half())
r = ops.FM(torch.randn(1, H, W, C, device="cuda").half())
ys = [ops.FM.empty(1, H, W, C) for _ in range(4)]
pc = ops.pack_conv(torch.randn(C, C, 3, 3) * 0.03, torch.zeros(C), stride=1, pad=1)
for rep in range(2):
    for nres in (0, 1):
        kw = dict(act=ops.ACT_LRELU, slope=0.1) if nres == 0 else dict(res=r)
        for _ in range(5):
            ops.conv(x, pc, out=ys[0], **kw)
        name = _lib.lib().tdvc_last_conv_kernel().decode()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40):
            ops.conv(x, pc, out=ys[i % 4], **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        print(f"{os.environ.get('TDVC_LIB', 'current build')}: nres={nres} {name} {us:7.1f} us  {2.0 * H * W * C * C * 9 / us / 1e6:7.1f} TFLOP/s", flush=True)
