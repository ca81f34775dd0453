"""repeat the conv_pair cases most exposed to synchronisation holes (short segments, ragged strips, second residual) and
compare every launch with the first: python tools/stress_pair.py [reps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(3)
w = [torch.randn(64, 64, 3, 3, device="cuda") * 0.05 for _ in range(2)]
b = [torch.randn(64, device="cuda") * 0.1 for _ in range(2)]
pp = ops.pack_conv_pair(w[0], b[0], w[1], b[1])
bad = 0
import ctypes
geom = int(os.environ.get("PAIR_GEOMETRY", "0"))          # 0: the per-width choice, 2 / 4: 30- / 62-column strips (conv_pair.hip, PairGeo<NCB>)
gfn = ops.L.lib().tdvc_debug_set_pair_geometry
gfn.argtypes, gfn.restype = [ctypes.c_int], None
gfn(geom)
load = os.environ.get("PAIR_SIDE_LOAD") == "1"            # a second stream keeps launching convs: other wave timings, other CU occupancy
if load:
    side = torch.cuda.Stream()
    lx = ops.FM(torch.randn(1, 272, 480, 128, device="cuda").half())
    lpc = ops.pack_conv(torch.randn(128, 128, 3, 3) * 0.03, torch.randn(128) * 0.1, stride=1, pad=1)
for (N, H, W, r2) in ((1, 33, 250, True), (1, 100, 131, False), (2, 64, 160, True), (5, 16, 1920, False), (1, 272, 480, True), (4, 48, 192, True),
                      (1, 1088, 1920, False), (3, 544, 960, True)):
    x = ops.FM(torch.randn(N, H, W, 64, device="cuda").half())
    res2 = ops.FM(torch.randn(N, H, W, 64, device="cuda").half()) if r2 else None
    y0 = ops.conv_pair(x, pp, res2=res2, act1=ops.ACT_LRELU, slope1=0.1)
    ref = y0.t.clone()
    n_bad = 0
    for it in range(reps if H * W * N < 1000000 else max(20, reps // 10)):
        if load and it % 4 == 0:
            with torch.cuda.stream(side):
                ops.conv(lx, lpc, act=ops.ACT_LRELU, slope=0.1)
        y = ops.conv_pair(x, pp, res2=res2, act1=ops.ACT_LRELU, slope1=0.1)
        if not torch.equal(y.t, ref):
            n_bad += 1
    print(f"{N}x{H}x{W} res2={r2} geometry {geom}: {n_bad} launches differ from the first", flush=True)
    bad += n_bad
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
