"""helpers shared by the GPU parity tests"""
import numpy as np
import torch


def rnd16(t):
    return t.half().float()


def to_fm(x_nchw, ops, Cpad=None, dtype=torch.float16):
    return ops.from_nchw(x_nchw.float().cuda().contiguous(), Cpad=Cpad, dtype=dtype)


def fm_to_cpu(fm, C=None):
    return fm.to_nchw(C).cpu()


def err_stats(got, ref):
    d = (got.double() - ref.double()).abs()
    return float(d.max()), float(d.mean()), float(ref.double().abs().max())


def assert_close(got, ref, rtol, atol, what, report=None):
    mx, mean, scale = err_stats(got, ref)
    bad = (got.double() - ref.double()).abs() > (atol + rtol * ref.double().abs())
    msg = f"{what}: max|d|={mx:.3e} mean|d|={mean:.3e} ref_absmax={scale:.3e} nbad={int(bad.sum())}/{bad.numel()}"
    if report:
        report(msg)
    assert not bad.any(), msg


def randn(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale
