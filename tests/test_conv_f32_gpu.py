"""The fp32 conv form (conv_f32.hip, the reference's fp32 islands: pnet.py:33,57) vs torch's fp32 conv on the same
fp32 operands: only the summation order differs, so the gate is 1e-5 of the result's scale."""
import pytest
import torch
import torch.nn.functional as F

from util import assert_close, fm_to_cpu, randn, to_fm

pytestmark = pytest.mark.gpu

RT, AT = 2e-5, 2e-5


def _ops():
    from tdvc_amd import ops
    return ops


CASES = [
    # name, N, cin, cout, k, stride, pad, H, W      (the layer shapes of Cheng2020Anchor / MVCoder / ResCoder)
    ("3x3_s2_64_128", 1, 64, 128, 3, 2, 1, 32, 48),
    ("3x3_s2_64_128_large", 1, 64, 128, 3, 2, 1, 192, 256),
    ("3x3_128_128", 2, 128, 128, 3, 1, 1, 17, 30),
    ("3x3_s2_128_128_odd", 1, 128, 128, 3, 2, 1, 18, 30),
    ("1x1_s2_64_128", 1, 64, 128, 1, 2, 0, 32, 64),
    ("1x1_s2_128_128", 1, 128, 128, 1, 2, 0, 16, 24),
    ("3x3_128_192", 1, 128, 192, 3, 1, 1, 17, 30),
    ("3x3_192_256", 1, 192, 256, 3, 1, 1, 8, 12),
    ("1x1_512_426", 1, 512, 426, 1, 1, 0, 8, 15),
    ("1x1_426_341", 1, 426, 341, 1, 1, 0, 8, 15),
    ("1x1_341_256", 1, 341, 256, 1, 1, 0, 8, 15),
    ("1x1_1536_256_ctx", 1, 1536, 256, 1, 1, 0, 1, 7),
    ("3x3_128_128_1x1", 1, 128, 128, 3, 1, 1, 1, 1),
    ("3x3_128_128_2x2", 2, 128, 128, 3, 1, 1, 2, 2),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_f32_plain(case, report):
    ops = _ops()
    name, N, cin, cout, k, s, p, H, W = case
    x = randn(N, cin, H, W, seed=1)
    w = randn(cout, cin, k, k, seed=2) * (1.0 / (cin * k * k) ** 0.5)
    b = randn(cout, seed=3) * 0.1
    ref = F.leaky_relu(F.conv2d(x, w, b, stride=s, padding=p), 0.01)
    pc = ops.pack_conv(w, b, stride=s, pad=p)
    y = ops.conv(to_fm(x, ops, Cpad=ops.pad8(cin), dtype=torch.float32), pc, act=ops.ACT_LRELU, slope=0.01)
    assert y.f32 and ops.L.lib().tdvc_last_conv_kernel() == b"conv_f32"
    assert_close(fm_to_cpu(y, cout), ref, RT, AT, f"conv_f32 {name}", report)
    if y.C > cout:      # padded channels must be exactly zero (they feed the next layer)
        assert float(fm_to_cpu(y)[:, cout:].abs().max()) == 0.0


def test_conv_f32_masked5x5_and_1x1_form(report):
    """the type-A masked 5x5 context conv and its tap-gathered 1x1 form (compress / decompress)"""
    ops = _ops()
    from tdvc_amd.model.coder import MaskedConv2d
    M = 128
    cp = MaskedConv2d(M, 2 * M, kernel_size=5, padding=2, stride=1)
    with torch.no_grad():
        cp.weight.copy_(randn(2 * M, M, 5, 5, seed=4) * 0.02)
        cp.bias.copy_(randn(2 * M, seed=5) * 0.1)
    x = torch.round(randn(1, M, 6, 9, seed=6) * 3.0)
    ref = F.conv2d(x, cp.weight.detach() * cp.mask, cp.bias.detach(), padding=2)
    pc = ops.pack_conv(cp.weight.detach(), cp.bias.detach(), stride=1, pad=2, taps=cp.live_taps())
    y = ops.conv(to_fm(x, ops, dtype=torch.float32), pc)
    assert_close(fm_to_cpu(y), ref, RT, AT, "conv_f32 masked 5x5", report)


def test_conv_f32_gdn_shuffle_residual(report):
    """GDN / inverse GDN (x * rsqrt(beta + gamma . x^2)), PixelShuffle store, fp32 and fp16 residuals, fp16 output"""
    ops = _ops()
    C, H, W = 128, 12, 20
    x = randn(1, C, H, W, seed=7)
    gamma = (randn(C, C, seed=8).abs() * 0.02 + 0.1 * torch.eye(C)).reshape(C, C, 1, 1)
    beta = randn(C, seed=9).abs() + 0.5
    r = randn(1, C, H, W, seed=10)
    pc = ops.pack_conv(gamma, beta, stride=1, pad=0)
    xf, rf = to_fm(x, ops, dtype=torch.float32), to_fm(r, ops, dtype=torch.float32)
    for inverse in (False, True):
        norm = F.conv2d(x * x, gamma, beta)
        ref = x * (torch.sqrt(norm) if inverse else torch.rsqrt(norm)) + r
        y = ops.conv(xf, pc, square=True, gdn=ops.GDN_INV if inverse else ops.GDN_FWD, aux=xf, res=rf)
        assert_close(fm_to_cpu(y), ref, RT, AT, f"conv_f32 {'i' if inverse else ''}GDN + fp32 residual", report)
    # sub-pixel conv: 128 -> 4 x 64, PixelShuffle(2), fp16 residual, fp16 output (the last layer of g_s + prediction)
    w = randn(256, C, 3, 3, seed=11) * 0.03
    b = randn(256, seed=12) * 0.1
    r16 = randn(1, 64, 2 * H, 2 * W, seed=13).half().float()
    ref = F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2) + r16
    y = ops.conv(xf, ops.pack_conv(w, b, stride=1, pad=1, shuffle=True), res=to_fm(r16, ops), out_dtype=torch.float16)
    assert not y.f32
    assert_close(fm_to_cpu(y), ref, 2e-3, 2e-3, "conv_f32 sub-pixel + fp16 residual -> fp16", report)
    y32 = ops.conv(xf, ops.pack_conv(w, b, stride=1, pad=1, shuffle=True))
    assert_close(fm_to_cpu(y32), F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2), RT, AT, "conv_f32 sub-pixel fp32", report)
