"""tdvc_conv_pair against the same pair as two tdvc_conv2d launches: python tools/bench_pair.py [H W N reps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import _lib
if os.environ.get("TDVC_LIB"):          # another build of the library (tools/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["TDVC_LIB"])
from tdvc_amd import ops

H, W, N, reps = (int(v) for v in (sys.argv[1:5] + ["1088", "1920", "1", "20"][len(sys.argv) - 1:]))
torch.manual_seed(0)
w1, w2 = torch.randn(64, 64, 3, 3, device="cuda") * 0.05, torch.randn(64, 64, 3, 3, device="cuda") * 0.05
b1, b2 = torch.randn(64, device="cuda") * 0.1, torch.randn(64, device="cuda") * 0.1
x = ops.FM(torch.randn(N, H, W, 64, device="cuda").half())
y = ops.FM.empty(N, H, W, 64)
t = ops.FM.empty(N, H, W, 64)
pp = ops.pack_conv_pair(w1, b1, w2, b2)
pc1, pc2 = ops.pack_conv(w1, b1, stride=1, pad=1), ops.pack_conv(w2, b2, stride=1, pad=1)


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def two():
    ops.conv(x, pc1, out=t, act=ops.ACT_RELU)
    ops.conv(t, pc2, out=y, res=x)


fl = 2 * 2.0 * N * H * W * 64 * 64 * 9
import ctypes, os
if os.environ.get("PAIR_EXPERIMENTS"):
    fn = ops.L.lib().tdvc_debug_set_pair_experiment
    fn.argtypes, fn.restype = [ctypes.c_int], None
    for e in [int(v) for v in os.environ["PAIR_EXPERIMENTS"].split(",")]:
        fn(e)
        print(f"experiment {e}: {timed(lambda: ops.conv_pair(x, pp, out=y)):.1f} us", flush=True)
    fn(0)
pair = lambda: ops.conv_pair(x, pp, out=y)
if os.environ.get("PAIR_GEOMETRIES", "1") == "1":          # 30- against 62-column strips (conv_pair.hip, PairGeo<NCB>), interleaved
    gfn = ops.L.lib().tdvc_debug_set_pair_geometry
    gfn.argtypes, gfn.restype = [ctypes.c_int], None
    res = {2: [], 4: []}
    timed(pair)
    for _ in range(4):
        for g_ in (2, 4):
            gfn(g_)
            res[g_].append(timed(pair))
    gfn(0)
    for g_, v in res.items():
        m = sorted(v)[len(v) // 2]
        print(f"{N}x{H}x{W}: {16 * g_ - 2}-column strips {m:.1f} us (runs {[round(u, 1) for u in v]}) = {fl / m / 1e6:.0f} TFLOP/s ({fl / m / 1e6 / 2500:.3f} of peak)", flush=True)
timed(pair); timed(two)                       # the first timed loop of a process runs ~20 % slow (clocks): not reported
A, B = [], []
for _ in range(4):
    A.append(timed(pair))
    B.append(timed(two))
a, b = sorted(A)[len(A) // 2], sorted(B)[len(B) // 2]
print(f"{N}x{H}x{W}: conv_pair {a:.1f} us (runs {[round(v, 1) for v in A]}) = {fl / a / 1e6:.0f} TFLOP/s ({fl / a / 1e6 / 2500:.3f} of peak); "
      f"two launches {b:.1f} us (runs {[round(v, 1) for v in B]}) = {fl / b / 1e6:.0f} TFLOP/s", flush=True)
