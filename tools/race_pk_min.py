"""Minimal probes of the packed-FP32 finding: tiny elementwise kernels (tools/experiments/pk_min.hip) repeated while a second
stream runs MFMA weight-gradient launches; every run is compared with the unloaded result."""
import ctypes
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops  # noqa: E402

exp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments")
so = os.path.join(exp, "libpkmin.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", so,
                           os.path.join(exp, "pk_min.hip")])
lib = ctypes.CDLL(so)
lib.pk_min_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
n = 4 * 256 * 256
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn(n, 2, device="cuda", generator=g)
b = torch.randn(n, 2, device="cuda", generator=g)
table = torch.randn(1 << 24, device="cuda", generator=g)          # 64 MB: the gathers of variants 4-6 miss the caches
N, H, W = 4, 256, 256
x = ops.FM(torch.randn(N, H, W, 32, device="cuda", generator=g).half())
gy = ops.FM(torch.randn(N, H, W, 64, device="cuda", generator=g).half())
pc = ops.pack_conv(torch.randn(64, 32, 7, 7) * 0.02, torch.zeros(64), stride=1, pad=3)
dw, db = torch.zeros(64 * 32 * 49, device="cuda"), torch.zeros(64, device="cuda")
side = torch.cuda.Stream()
LOAD = os.environ.get("PK_LOAD", "wgrad")
big1 = torch.randn(64 << 20, device="cuda", generator=g)
big2 = torch.empty_like(big1)


def run(v):
    c = torch.zeros_like(a)
    for _ in range(8):                                     # a few launches per run: more waves exposed
        assert lib.pk_min_launch(v, a.data_ptr(), b.data_ptr(), c.data_ptr(), n, table.data_ptr(), (1 << 24) - 1, torch.cuda.current_stream().cuda_stream) == 0
    return c


print(f"side-stream load: {LOAD}")
for v in ((10, 11) if LOAD != "wgrad" else (0, 4, 7, 9, 10, 11, 12, 8)):
    ref = run(v)
    torch.cuda.synchronize()
    bad, lanes = 0, torch.zeros(64, dtype=torch.long, device="cuda")
    for it in range(30):
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            if LOAD == "wgrad":                            # MFMA kernels
                for _ in range(3):
                    ops.conv_wgrad(pc, gy, x, dw, scale=1.0, db=db)
            elif LOAD == "copy":                           # memory-bound, no matrix instructions
                for _ in range(6):
                    big2.copy_(big1)
            elif LOAD == "valu":                           # vector-ALU bound, no matrix instructions
                for _ in range(3):
                    torch.erfinv(torch.tanh(big1), out=big2)
        c = run(v)
        torch.cuda.synchronize()
        d = (c != ref).any(dim=1)
        if bool(d.any()):
            bad += 1
            lanes += torch.bincount(d.nonzero().flatten() % 64, minlength=64)
    q = [int(lanes[16 * k:16 * k + 16].sum()) for k in range(4)]
    print(f"pk_min variant {v}: {bad}/30 loaded runs differ from the unloaded result; wrong elements by quarter wave {q}")
