"""`_ext.dcn_v2_backward` host wrapper (src/dcn_v2.h:48-92, src/cuda/dcn_v2_cuda.cu:97-216)."""
from __future__ import annotations

import torch

from . import _lib as L
from .ops import _stream


def dcn_v2_backward(input, weight, bias, offset, mask, grad_output, kh, kw, sh, sw, ph, pw, dh, dw, deformable_group):
    ts = (input, weight, bias, offset, mask, grad_output)
    for t in ts:
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32):
            raise RuntimeError("dcn_v2_backward: fp32 CUDA/HIP tensors only")
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    input, weight, bias, offset, mask, grad_output = [t.contiguous() for t in ts]
    B, C, H, W = input.shape
    Cout = weight.shape[0]
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    gi, go, gm = torch.empty_like(input), torch.empty_like(offset), torch.empty_like(mask)
    gw, gb = torch.empty_like(weight), torch.empty_like(bias)
    cols = torch.empty(C * kh * kw * Ho * Wo, dtype=torch.float32, device=input.device)
    with torch.cuda.device(input.device):
        L.check(L.lib().tdvc_dcn_v2_backward_f32(
            input.data_ptr(), weight.data_ptr(), bias.data_ptr(), offset.data_ptr(), mask.data_ptr(),
            grad_output.data_ptr(), gi.data_ptr(), go.data_ptr(), gm.data_ptr(), gw.data_ptr(), gb.data_ptr(),
            cols.data_ptr(), B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, deformable_group, _stream()),
            "dcn_v2_backward")
    return [gi, go, gm, gw, gb]


class _DCNv2(torch.autograd.Function):
    """autograd twin of the reference's `_DCNv2` (main/utils/dcnv2/dcn_v2_amp.py:23-119): fp32 compute,
    optional fp16 rounding of the output / gradients (`use_amp`, which the reference leaves True)."""

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups, use_amp=True):
        from .ops import dcn_v2_forward
        if use_amp:
            input, offset, mask, weight, bias = (t.float() for t in (input, offset, mask, weight, bias))
        kh, kw = weight.shape[2:4]
        ctx.cfg = (kh, kw, stride, stride, padding, padding, dilation, dilation, deformable_groups)
        ctx.use_amp = use_amp
        out = dcn_v2_forward(input, weight, bias, offset, mask, *ctx.cfg)
        ctx.save_for_backward(input, offset, mask, weight, bias)
        return out.half() if use_amp else out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask, weight, bias = ctx.saved_tensors
        go = grad_output.float().contiguous() if ctx.use_amp else grad_output.contiguous()
        gi, goff, gm, gw, gb = dcn_v2_backward(input, weight, bias, offset, mask, go, *ctx.cfg)
        if ctx.use_amp:
            gi, goff, gm, gw, gb = (t.half() for t in (gi, goff, gm, gw, gb))
        return gi, goff, gm, gw, gb, None, None, None, None, None


def dcn_v2_conv(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups, use_amp=True):
    return _DCNv2.apply(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups, use_amp)
