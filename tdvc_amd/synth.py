"""Deterministic synthetic inputs and weights.

No checkpoint ships with the reference (`main/pretrained/.gitkeep`) and there is no
network, so both the CPU oracle and the MI355X path regenerate *identical* weights
from a closed-form, RNG-version-independent filler keyed by state-dict name, and
identical 7-frame GOPs from a seeded numpy generator.

Nothing here touches the GPU; it is shared by the product, the oracle and the tests.
"""
from __future__ import annotations

import math
import zlib

import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x: np.ndarray) -> np.ndarray:
    """murmur3 finaliser on uint64 lanes holding 32-bit values (vectorised, exact)."""
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x


def hash_uniform(n: int, salt: int) -> np.ndarray:
    """n doubles in [0,1), a pure function of (index, salt)."""
    idx = np.arange(n, dtype=np.uint64)
    h = _mix32(idx * np.uint64(0x9E3779B1) + np.uint64(salt & 0xFFFFFFFF))
    h = _mix32(h ^ np.uint64((salt * 0x632BE5AB) & 0xFFFFFFFF))
    return h.astype(np.float64) / 4294967296.0


def _salt(name: str) -> int:
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def _uniform_like(t: torch.Tensor, name: str, amp: float) -> torch.Tensor:
    u = hash_uniform(t.numel(), _salt(name))
    v = (u * 2.0 - 1.0) * amp
    return torch.from_numpy(v.astype(np.float32)).view(t.shape)


# gain applied to He-style uniform conv init; keeps activations O(1) over ~60 layers
_CONV_GAIN = 1.0


def fill_parameters(module: torch.nn.Module) -> None:
    """Overwrite every parameter of `module` in place with the closed-form filler.

    Rules (keyed on the state-dict name / shape):
      * conv / linear weights (ndim >= 2, not GDN gamma, not entropy-bottleneck
        matrices): uniform(-a, a), a = gain * sqrt(3 / fan_in)
      * biases: uniform(-0.05, 0.05)
      * GDN beta/gamma, entropy-bottleneck matrices/factors/quantiles: kept at the
        CompressAI default initialisation (already closed-form), biases of the
        entropy bottleneck: hash-uniform(-0.5, 0.5) (their default is torch RNG)
      * DCN `conv_offset_mask`: small non-zero weights so offsets are ~±1.5 px
        (the reference zero-initialises them, `dcn_v2_amp.py:213-215`, which would
        make the deformable path trivial)
      * last SPyNet conv of each level (16->2): scaled by 0.05 so the synthetic flow
        stays within a few pixels
      * last conv of each analysis transform (`g_a.7`): x4 so latents span several bins;
        `loopfilter.featdown`: weight x0.25, bias +0.5 so the output is not clamped away;
        `entropy_parameters.4.bias[:N]` (+1.5): predicted Gaussian scales of order 1, so the
        synthetic rate is a few bits per latent instead of sitting on the 1e-9 likelihood floor
    """
    with torch.no_grad():
        for name, p in module.named_parameters():
            leaf = name.rsplit(".", 1)[-1]
            if leaf in ("beta", "gamma") and ("gdn" in name):
                continue
            if leaf.startswith("_matrix") or leaf.startswith("_factor") or leaf == "quantiles":
                continue
            if leaf.startswith("_bias"):
                p.copy_(_uniform_like(p, name, 0.5))
                continue
            if p.ndim >= 2:
                fan_in = p[0].numel()
                amp = _CONV_GAIN * math.sqrt(3.0 / fan_in)
                if "conv_offset_mask" in name:
                    amp *= 0.5
                if ".spynet." in name and p.shape[0] == 2:
                    amp *= 0.05
                if ".g_a.7." in name:
                    amp *= 4.0          # spread the latents over several quantiser bins
                if name.endswith("loopfilter.featdown.weight"):
                    amp *= 0.25
                p.copy_(_uniform_like(p, name, amp))
            else:
                p.copy_(_uniform_like(p, name, 0.05))
                if name.endswith("loopfilter.featdown.bias"):
                    p.add_(0.5)         # keep the synthetic reconstruction inside (0, 1)
                if name.endswith("entropy_parameters.4.bias"):
                    p[: p.numel() // 2].add_(1.5)   # predicted scales ~1.5: a sane (not 1e-9-floor) rate model


def _box_blur(a: np.ndarray, k: int) -> np.ndarray:
    """k x k box filter over the last two axes, 'valid'-free (edge padded), float64."""
    r = k // 2
    ap = np.pad(a, ((0, 0), (r, r), (r, r)), mode="edge")
    c = np.cumsum(ap, axis=1)
    c = np.concatenate([np.zeros_like(c[:, :1]), c], axis=1)
    a1 = (c[:, k:] - c[:, :-k]) / k
    c = np.cumsum(a1, axis=2)
    c = np.concatenate([np.zeros_like(c[:, :, :1]), c], axis=2)
    return (c[:, :, k:] - c[:, :, :-k]) / k


def make_gop(seed: int, T: int, H: int, W: int) -> torch.Tensor:
    """(T, 3, H, W) float32 in [0,1], uint8-quantised.

    base texture = uniform noise (3, H+64, W+64) box-blurred 9x9 twice, min-max
    normalised; frame t = crop shifted by (dy, dx) = (t, 2t) px + N(0, 0.01^2) noise.
    Frame 0 doubles as the I-frame reconstruction.
    """
    rng = np.random.default_rng(seed)
    base = rng.random((3, H + 64, W + 64))
    base = _box_blur(_box_blur(base, 9), 9)
    base = (base - base.min()) / (base.max() - base.min())
    frames = np.empty((T, 3, H, W), dtype=np.float32)
    for t in range(T):
        dy, dx = t, 2 * t
        crop = base[:, 16 + dy:16 + dy + H, 16 + dx:16 + dx + W]
        noisy = crop + rng.standard_normal(crop.shape) * 0.01
        q = np.clip(np.rint(np.clip(noisy, 0.0, 1.0) * 255.0), 0, 255)
        frames[t] = (q / 255.0).astype(np.float32)
    return torch.from_numpy(frames)


def ref_list(refs: list) -> torch.Tensor:
    """Reference-list rule of `tools/predict.py:55-62`: [I,I,I,I], [I,I,x1,x1],
    then [I, x(t-3), x(t-2), x(t-1)].  `refs[0]` is the (padded) I-frame,
    the rest are padded reconstructions.  Returns (B, 4, 3, H, W)."""
    if len(refs) == 1:
        sel = [refs[0], refs[-1], refs[-1], refs[-1]]
    elif len(refs) == 2:
        sel = [refs[0], refs[-2], refs[-1], refs[-1]]
    else:
        sel = [refs[0], refs[-3], refs[-2], refs[-1]]
    return torch.stack(sel, dim=1)


def split_optim_params(net):
    """`configure_optimizers` partition of `main/utils/utils.py:90-113`: (main, aux) parameter
    names; aux = names ending `.quantiles` (Adam lr vs Adam 10*lr)."""
    main = sorted(n for n, p in net.named_parameters() if not n.endswith(".quantiles") and p.requires_grad)
    aux = sorted(n for n, p in net.named_parameters() if n.endswith(".quantiles") and p.requires_grad)
    assert not (set(main) & set(aux))
    return main, aux
