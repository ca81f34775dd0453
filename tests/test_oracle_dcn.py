"""CPU: the C restatement of the reference's DCN CPU kernels, pinned by the reference's own tests."""
import torch

from oracle import dcn_c
from oracle.tdvc_ref.blocks import dcn_v2_forward_ref


def test_check_zero_offset_kat():
    """main/utils/dcnv2/testcpu.py:34-69"""
    N, C, H, W, k = 2, 2, 4, 4, 3
    w = torch.zeros(C, C, k, k)
    for c in range(C):
        w[c, c, 1, 1] = 1.0
    x = torch.randn(N, C, H, W, generator=torch.Generator().manual_seed(0))
    out = dcn_c.forward(x, w, torch.zeros(C), torch.zeros(N, 2 * k * k, H, W), torch.sigmoid(torch.zeros(N, k * k, H, W)),
                        k, k, 1, 1, 1, 1, 1, 1, 1)
    assert float((x - 2 * out).abs().max()) < 1e-10


def test_c_forward_equals_torch_restatement():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 16, 11, 13, generator=g)
    w = torch.randn(24, 16, 3, 3, generator=g) * 0.2
    b = torch.randn(24, generator=g)
    off = torch.randn(2, 2 * 4 * 9, 6, 7, generator=g) * 2
    m = torch.sigmoid(torch.randn(2, 4 * 9, 6, 7, generator=g))
    a = dcn_c.forward(x, w, b, off, m, 3, 3, 2, 2, 1, 1, 1, 1, 4)
    r = dcn_v2_forward_ref(x, w, b, off, m, 3, 3, 2, 2, 1, 1, 1, 1, 4)
    assert torch.allclose(a, r, atol=1e-5, rtol=1e-5)


def test_gradcheck_reference_settings():
    """check_gradient_dconv (testcpu.py:71-99): N=2, C=2, 4x4, offsets ~ N(0, 2^2), eps=1e-3, atol=1e-4,
    rtol=1e-2.  (float64 inputs are cast to fp32 inside, like `input.float()` in dcn_v2_amp.py:37-42, so
    the finite differences see fp32 resolution: eps is the reference's 1e-3.)"""
    g = torch.Generator().manual_seed(2)
    N, C, H, W, outC, k = 2, 2, 4, 4, 2, 3
    x = (torch.rand(N, C, H, W, generator=g) * 0.01).double().requires_grad_()
    off = (torch.randn(N, 2 * k * k, H, W, generator=g) * 2).double()
    # keep sampling points away from integer coordinates: the bilinear kernel is not differentiable there
    frac = off - torch.floor(off)
    off = (torch.floor(off) + frac.clamp(0.15, 0.85)).requires_grad_()
    msk = torch.sigmoid(torch.rand(N, k * k, H, W, generator=g)).double().requires_grad_()
    w = torch.randn(outC, C, k, k, generator=g).double().requires_grad_()
    b = torch.rand(outC, generator=g).double().requires_grad_()
    assert torch.autograd.gradcheck(dcn_c.DCNv2Function.apply, (x, off, msk, w, b, 1, 1, 1, 1),
                                    eps=1e-3, atol=1e-4, rtol=1e-2, nondet_tol=1e-6)
