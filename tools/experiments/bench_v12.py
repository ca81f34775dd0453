"""conv_mfma_v12 against conv_mfma_v11 on 3x3 128->128: python tools/bench_v12.py [H W N nres reps]"""
import ctypes, sys
import torch
sys.path.insert(0, "/root/repo")
from tdvc_amd import ops

H, W, N, nres, reps = (int(v) for v in (sys.argv[1:6] + ["544", "960", "1", "0", "20"][len(sys.argv) - 1:]))
fn = ops.L.lib().tdvc_debug_enable_conv_v12
fn.argtypes, fn.restype = [ctypes.c_int], None
torch.manual_seed(0)
pc = ops.pack_conv(torch.randn(128, 128, 3, 3) * 0.03, torch.randn(128) * 0.1, stride=1, pad=1)
x = ops.FM(torch.randn(N, H, W, 128, device="cuda").half())
y = ops.FM.empty(N, H, W, 128)
rs = [ops.FM(torch.randn(N, H, W, 128, device="cuda").half()) for _ in range(nres)]
kw = dict(act=ops.ACT_LRELU, slope=0.01)
if nres > 0:
    kw["res"] = rs[0]
if nres > 1:
    kw["res2"] = rs[1]


def timed():
    for _ in range(3):
        ops.conv(x, pc, out=y, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ops.conv(x, pc, out=y, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fl = 2.0 * N * H * W * 128 * 128 * 9
fn(1); timed(); fn(0); timed()
A, B = [], []
for _ in range(4):
    fn(1); A.append(timed()); k1 = ops.L.lib().tdvc_last_conv_kernel().decode()
    fn(0); B.append(timed()); k0 = ops.L.lib().tdvc_last_conv_kernel().decode()
fn(1)
a, b = sorted(A)[2], sorted(B)[2]
print(f"{N}x{H}x{W} nres={nres}: {k1} {a:.1f} us (runs {[round(v, 1) for v in A]}) = {fl / a / 1e6:.0f} TFLOP/s ({fl / a / 1e6 / 2500:.3f}); "
      f"{k0} {b:.1f} us (runs {[round(v, 1) for v in B]}) = {fl / b / 1e6:.0f} TFLOP/s", flush=True)
