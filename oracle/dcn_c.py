"""ctypes front-end of oracle/dcn_ref.c (CPU ORACLE — test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdcn_ref.so")
_lib = None


class Geo(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("C", "H", "W", "Ho", "Wo", "kh", "kw", "ph", "pw", "sh", "sw", "dh", "dw", "G")]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE])
        _lib = C.CDLL(_SO)
    return _lib


def _geo(x, kh, kw, sh, sw, ph, pw, dh, dw, G):
    _, Cc, H, W = x.shape
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    return Geo(Cc, H, W, Ho, Wo, kh, kw, ph, pw, sh, sw, dh, dw, G)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def forward(x, w, b, off, msk, kh, kw, sh, sw, ph, pw, dh, dw, G):
    g = _geo(x, kh, kw, sh, sw, ph, pw, dh, dw, G)
    xs, ws, bs, os_, ms = [np.ascontiguousarray(t.detach().numpy(), dtype=np.float32) for t in (x, w, b, off, msk)]
    out = np.zeros((x.shape[0], w.shape[0], g.Ho, g.Wo), dtype=np.float32)
    lib().dcn_ref_forward(_p(xs), _p(ws), _p(bs), _p(os_), _p(ms), _p(out), x.shape[0], w.shape[0], C.byref(g))
    return torch.from_numpy(out)


def backward(x, w, b, off, msk, gout, kh, kw, sh, sw, ph, pw, dh, dw, G):
    g = _geo(x, kh, kw, sh, sw, ph, pw, dh, dw, G)
    xs, ws, os_, ms, gs = [np.ascontiguousarray(t.detach().numpy(), dtype=np.float32) for t in (x, w, off, msk, gout)]
    gx, goff, gm, gw, gb = (np.zeros(t.shape, dtype=np.float32) for t in (x, off, msk, w, b))
    lib().dcn_ref_backward(_p(xs), _p(ws), _p(os_), _p(ms), _p(gs), _p(gx), _p(goff), _p(gm), _p(gw), _p(gb),
                           x.shape[0], w.shape[0], C.byref(g))
    return [torch.from_numpy(a) for a in (gx, goff, gm, gw, gb)]


class DCNv2Function(torch.autograd.Function):
    """autograd wrapper in the shape of `_DCNv2` (dcn_v2_amp.py:23-119) for gradcheck"""

    @staticmethod
    def forward(ctx, x, off, msk, w, b, stride, padding, dilation, G):
        ctx.cfg = (w.shape[2], w.shape[3], stride, stride, padding, padding, dilation, dilation, G)
        ctx.save_for_backward(x, off, msk, w, b)
        return forward(x.float(), w.float(), b.float(), off.float(), msk.float(), *ctx.cfg).to(x.dtype)

    @staticmethod
    def backward(ctx, gout):
        x, off, msk, w, b = ctx.saved_tensors
        gx, goff, gm, gw, gb = backward(x.float(), w.float(), b.float(), off.float(), msk.float(), gout.float().contiguous(), *ctx.cfg)
        return gx.to(x.dtype), goff.to(x.dtype), gm.to(x.dtype), gw.to(x.dtype), gb.to(x.dtype), None, None, None, None
