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


def affine_dcn_case(N=1, G=2, cpg=3, Cout=5, H=14, W=17, k=3, seed=7, osc=1.2):
    """A case whose answer is known in closed form, FRACTIONAL offsets included: on an image that is affine in (h, w) per channel,
    x[c,h,w] = A_c h + B_c w + C_c, bilinear interpolation is exact, so the sample of channel c at tap (i, j) of pixel (h, w)
    is A_c (h - 1 + i + dh) + B_c (w - 1 + j + dw) + C_c wherever the four corners lie inside the image, and
    out[co] = bias[co] + sum_{c,t} W[co,c,t] m[g(c),t] sample(c,t)   (dcn_v2_im2col_cuda.cu:125-195, dcn_v2_cuda.cu:69-92).
    Returns the operands, the closed-form output and the mask of pixels all of whose samples are interior."""
    g = torch.Generator().manual_seed(seed)
    C, K = G * cpg, k * k
    A, B, C0 = torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g)
    hh = torch.arange(H, dtype=torch.float64).view(1, 1, H, 1)
    ww = torch.arange(W, dtype=torch.float64).view(1, 1, 1, W)
    x = (A.double().view(1, C, 1, 1) * hh + B.double().view(1, C, 1, 1) * ww + C0.double().view(1, C, 1, 1)).expand(N, C, H, W).contiguous()
    w = torch.randn(Cout, C, k, k, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    off = (torch.rand(N, 2 * G * K, H, W, generator=g) * 2 - 1) * osc            # fractional, |.| <= osc
    m = torch.sigmoid(torch.randn(N, G * K, H, W, generator=g))
    out = b.double().view(1, Cout, 1, 1).expand(N, Cout, H, W).clone()
    ok = torch.ones(N, H, W, dtype=torch.bool)
    for gi in range(G):
        for t in range(K):
            i, j = t // k, t % k
            ph = hh.view(1, H, 1) - 1 + i + off[:, gi * 2 * K + 2 * t].double()     # (N, H, W)
            pw = ww.view(1, 1, W) - 1 + j + off[:, gi * 2 * K + 2 * t + 1].double()
            ok &= (ph >= 0) & (ph <= H - 1) & (pw >= 0) & (pw <= W - 1)
            for cc in range(cpg):
                c = gi * cpg + cc
                samp = (A[c].double() * ph + B[c].double() * pw + C0[c].double()) * m[:, gi * K + t].double()
                out += w[:, c, i, j].double().view(1, Cout, 1, 1) * samp.unsqueeze(1)
    return x.float(), w, b, off, m, out, ok


def test_fractional_offsets_closed_form_on_affine_image():
    """Pins the FRACTIONAL-offset sampling (the zero-offset KAT and the integer-shift property do not): the C restatement and the
    torch restatement against the closed form on an affine image"""
    x, w, b, off, m, want, ok = affine_dcn_case()
    assert int(ok.sum()) > 40                                  # enough interior pixels
    sel = ok.unsqueeze(1).expand_as(want)
    for name, fn in (("C restatement", dcn_c.forward), ("torch restatement", dcn_v2_forward_ref)):
        got = fn(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 2).double()
        err = float((got - want)[sel].abs().max())
        assert err < 2e-4 * max(1.0, float(want[sel].abs().max())), f"{name}: max |out - closed form| = {err:.3e}"
