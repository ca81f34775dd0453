"""A/B inside ONE process: median 1080p P-frame time with two statements toggled (names in scope: net, lib, torch).
python tools/ab_infer.py "stmt_a" "stmt_b" """
import ctypes
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib  # noqa: E402
from tdvc_amd.codec_utils import pad  # noqa: E402
from tdvc_amd.model import VideoCompressor  # noqa: E402
from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402

lib = _lib.lib()
lib.tdvc_debug_set_conv_v9_work_limit.argtypes = [ctypes.c_long]
sa, sb = sys.argv[1], sys.argv[2]
net = VideoCompressor(); fill_parameters(net); net = net.cuda().eval()
g = make_gop(1234, 7, 1080, 1920).cuda()
frames = [pad(g[i:i + 1], 64) for i in range(7)]


def gop():
    refs = [frames[0]]
    ts = []
    for t in range(1, 7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        recon, _, _ = net(frames[t], ref_list(refs), True)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        refs.append(recon)
    return ts


with torch.no_grad():
    gop()
    res = {sa: [], sb: []}
    for rnd in range(4):
        for s in (sa, sb):
            exec(s)
            gop()
            res[s] += gop() + gop()
for k, v in res.items():
    print(f"{k}: median {statistics.median(v):.3f} ms  min {min(v):.3f}  p90 {sorted(v)[int(0.9 * len(v))]:.3f}  (n={len(v)})")
