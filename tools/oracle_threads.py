"""How fast is the CPU oracle (the checker of the GPU tests) on this host as a function of torch's intra-op threads?  One P-frame at 512x768
and one at 1088x1920 (fp32 path), filler weights.  python tools/oracle_threads.py [threads ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.tdvc_ref import VideoCompressor as Ref  # noqa: E402
from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402

ths = [int(v) for v in sys.argv[1:]] or [8, 16, 32, 64]
print("default torch threads", torch.get_num_threads(), "cpus", os.cpu_count(), flush=True)
net = Ref().eval()
fill_parameters(net)
for (H, W) in ((512, 768), (1088, 1920)):
    g = make_gop(1234, 2, H, W)
    x, refs = g[1:2], ref_list([g[0:1]])
    for th in ths:
        if (H, W) == (1088, 1920) and th not in (16, 32):
            continue
        torch.set_num_threads(th)
        with torch.no_grad():
            t0 = time.perf_counter()
            net(x, refs, False)
            dt = time.perf_counter() - t0
        print(f"{H}x{W} threads {th}: {dt:.1f} s", flush=True)
