"""Caller-side helpers of the P-frame path (the parts of tools/predict.py / main/utils/utils.py a
driver needs): centred zero pad / crop to a multiple of 64 and the per-frame metrics."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def pad(x: torch.Tensor, p: int = 64) -> torch.Tensor:
    """`main/utils/utils.py:59-72`"""
    h, w = x.shape[-2:]
    H, W = (h + p - 1) // p * p, (w + p - 1) // p * p
    left, top = (W - w) // 2, (H - h) // 2
    return F.pad(x, (left, W - w - left, top, H - h - top), mode="constant", value=0)


def crop(x: torch.Tensor, size) -> torch.Tensor:
    """`main/utils/utils.py:75-87`"""
    H, W = x.shape[-2:]
    h, w = size
    left, top = (W - w) // 2, (H - h) // 2
    return x[..., top:top + h, left:left + w]


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """`tools/predict.py:87-88`: 10*log10(1/MSE) on [0,1] images"""
    mse = float(torch.mean((a.float() - b.float()) ** 2))
    return 10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf")
