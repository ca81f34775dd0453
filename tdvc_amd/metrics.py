"""PSNR and MS-SSIM of the evaluation loop (`tools/predict.py:87-100`), on the GPU.

`ms_ssim` / `ssim` keep the signatures of `main/model/ms_ssim_torch.py:87-191` (float32 NCHW inputs, `data_range`,
`size_average`, `weights`, optional 1-D `win`); every level is one HIP pass over X and Y (`csrc/metrics.hip`) instead of
ten depthwise convolutions over five full-size products.  No CPU / eager fallback: the HIP library must load."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib as L

_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def gauss_1d(size: int = 11, sigma: float = 1.5) -> list[float]:
    """ms_ssim_torch.py:5-18 (float32 arithmetic: exp, normalise by the sum)"""
    c = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    return (g / g.sum()).tolist()


def _check(X, Y, win_size):
    if X.dim() != 4:
        raise ValueError("Input images must 4-d tensor.")
    if X.dtype != Y.dtype or X.device != Y.device:
        raise ValueError("Input images must have the same dtype.")
    if X.shape != Y.shape:
        raise ValueError("Input images must have the same dimensions.")
    if win_size % 2 != 1:
        raise ValueError("Window size must be odd.")
    if not X.is_cuda:
        raise RuntimeError("tdvc_amd.metrics runs on the GPU only (no CPU fallback)")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _level(X, Y, taps, data_range):
    """-> (ssim, cs) per image, already through the reference's (v + 1) / 2 (ms_ssim_torch.py:79-81)"""
    lib = L.lib()
    N, Cc, H, W = X.shape
    n = lib.tdvc_ssim_level_work_floats(N, Cc, H, W, len(taps))
    if n <= 0:
        raise ValueError(f"ms_ssim: image {H}x{W} smaller than the {len(taps)}-tap window (or window > 15 taps)")
    work = torch.empty(n, dtype=torch.float32, device=X.device)
    out = torch.empty(2, N, dtype=torch.float32, device=X.device)
    win = (C.c_float * len(taps))(*taps)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    L.check(lib.tdvc_ssim_level(X.data_ptr(), Y.data_ptr(), N, Cc, H, W, win, len(taps), c1, c2, out[0].data_ptr(), out[1].data_ptr(),
                                work.data_ptr(), n, _stream()), "ssim_level")
    return (out[0] + 1) / 2, (out[1] + 1) / 2


def _pool(X):
    N, Cc, H, W = X.shape
    Ho, Wo = (H + 2 * (H % 2) - 2) // 2 + 1, (W + 2 * (W % 2) - 2) // 2 + 1
    out = torch.empty(N, Cc, Ho, Wo, dtype=torch.float32, device=X.device)
    L.check(L.lib().tdvc_avgpool2_pad_f32(X.data_ptr(), N * Cc, H, W, out.data_ptr(), _stream()), "avgpool2_pad_f32")
    return out


def _prep(X, Y, win_size, win_sigma, win):
    taps = gauss_1d(win_size, win_sigma) if win is None else [float(v) for v in torch.as_tensor(win).reshape(-1, torch.as_tensor(win).shape[-1])[0]]
    return X.float().contiguous(), Y.float().contiguous(), taps


def ssim(X, Y, win_size=11, win_sigma=1.5, win=None, data_range=255, size_average=True, full=False):
    _check(X, Y, win_size)
    X, Y, taps = _prep(X, Y, win_size, win_sigma, win)
    s, cs = _level(X, Y, taps, float(data_range))
    if size_average:
        s, cs = s.mean(), cs.mean()
    return (s, cs) if full else s


def ms_ssim(X, Y, win_size=11, win_sigma=1.5, win=None, data_range=255, size_average=True, full=False, weights=None):
    """ms_ssim_torch.py:132-191: cs of the first levels and ssim of the last, weighted product"""
    _check(X, Y, win_size)
    X, Y, taps = _prep(X, Y, win_size, win_sigma, win)
    w = torch.tensor(_WEIGHTS if weights is None else [float(v) for v in weights], dtype=torch.float32, device=X.device)
    mcs = []
    for lv in range(w.numel()):
        s, cs = _level(X, Y, taps, float(data_range))
        mcs.append(cs)
        if lv + 1 < w.numel():
            X, Y = _pool(X), _pool(Y)
    mcs = torch.stack(mcs, dim=0)
    val = torch.prod((mcs[:-1] ** w[:-1].unsqueeze(1)) * (s ** w[-1]), dim=0)
    return val.mean() if size_average else val


def psnr(recon: torch.Tensor, target: torch.Tensor) -> float:
    """tools/predict.py:87-88: 10 log10(1 / MSE) on [0, 1] images"""
    mse = float(torch.mean((recon.float() - target.float()) ** 2))
    return 10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf")
