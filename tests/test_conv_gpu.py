"""MFMA conv kernel vs torch fp32 conv on fp16-rounded operands (so the only differences are
fp32 summation order and the final fp16 rounding of the output)."""
import pytest
import torch
import torch.nn.functional as F

from util import assert_close, fm_to_cpu, randn, rnd16, to_fm

pytestmark = pytest.mark.gpu

RT, AT = 2e-3, 2e-3


def _ops():
    from tdvc_amd import ops
    return ops


@pytest.fixture(autouse=True, params=["v9", "tiled"])
def small_map_kernel(request):
    """every case of this module runs twice: with the small-map split-K kernel (conv_mfma_v9, the product dispatch for
    <= 8192 output pixels) and with it switched off, so the tiled kernels keep their small-shape coverage"""
    import ctypes

    from tdvc_amd import _lib
    fn = _lib.lib().tdvc_debug_enable_conv_v9
    fn.argtypes = [ctypes.c_int]
    fn.restype = None
    fn(1 if request.param == "v9" else 0)
    yield request.param
    fn(1)


def _enable(name, on):
    import ctypes

    from tdvc_amd import _lib
    fn = getattr(_lib.lib(), "tdvc_debug_enable_" + name)
    fn.argtypes, fn.restype = [ctypes.c_int], None
    fn(int(on) if not isinstance(on, bool) else ((15 if name == "conv_row" else 1) if on else 0))     # conv_row: one bit per geometry (Cin 128, Cin 64, Cin 128 -> 64, stride 2)


CASES = [
    # name, N, cin, cout, k, stride, pad, H, W
    ("3x3_64_64", 1, 64, 64, 3, 1, 1, 24, 40),
    ("3x3_64_64_big", 2, 64, 64, 3, 1, 1, 70, 100),
    ("3x3_s2_64_128", 1, 64, 128, 3, 2, 1, 32, 48),
    ("3x3_s2_odd", 1, 128, 128, 3, 2, 1, 18, 30),
    ("3x3_s2_64_128_large", 1, 64, 128, 3, 2, 1, 192, 256),     # > 8192 output pixels: the space-to-depth form (v3)
    ("1x1_128_64", 1, 128, 64, 1, 1, 0, 20, 36),
    ("1x1_s2_64_128", 1, 64, 128, 1, 2, 0, 32, 64),
    ("1x1_256_64", 1, 256, 64, 1, 1, 0, 16, 40),
    ("3x3_3_64", 1, 3, 64, 3, 1, 1, 33, 47),
    ("3x3_3_64_c8", 3, 3, 64, 3, 1, 1, 70, 150),                 # >= 8192 output pixels: the first-layer kernel (conv_c8), ragged row segments
    ("7x7_8_32", 1, 8, 32, 7, 1, 3, 34, 60),
    ("7x7_32_64", 1, 32, 64, 7, 1, 3, 17, 30),
    ("7x7_64_32", 1, 64, 32, 7, 1, 3, 17, 30),
    ("7x7_32_16", 1, 32, 16, 7, 1, 3, 9, 15),
    ("3x3_128_192", 1, 128, 192, 3, 1, 1, 17, 30),
    ("1x1_512_426", 1, 512, 426, 1, 1, 0, 8, 15),
    ("1x1_426_341", 1, 426, 341, 1, 1, 0, 8, 15),
    ("3x3_64_216", 1, 64, 216, 3, 1, 1, 16, 32),
    ("3x3_192_256", 1, 192, 256, 3, 1, 1, 8, 12),
    ("3x3_128_128_4x4", 2, 128, 128, 3, 1, 1, 4, 4),            # hyper-analysis sizes at a 64x64 crop
    ("3x3_128_128_2x2", 1, 128, 128, 3, 1, 1, 2, 2),
    ("3x3_128_128_1x1", 1, 128, 128, 3, 1, 1, 1, 1),
    ("3x3_s2_128_128_4x4", 1, 128, 128, 3, 2, 1, 4, 4),
    ("5x5_128_256_3x5", 1, 128, 256, 5, 1, 2, 3, 5),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_plain(case, report):
    ops = _ops()
    name, N, cin, cout, k, s, p, H, W = case
    x = rnd16(randn(N, cin, H, W, seed=1))
    w = rnd16(randn(cout, cin, k, k, seed=2) * (1.0 / (cin * k * k) ** 0.5))
    b = randn(cout, seed=3) * 0.1
    ref = F.leaky_relu(F.conv2d(x, w, b, stride=s, padding=p), 0.1)
    pc = ops.pack_conv(w, b, stride=s, pad=p)
    y = ops.conv(to_fm(x, ops), pc, act=ops.ACT_LRELU, slope=0.1)
    got = fm_to_cpu(y, cout)
    assert_close(got, ref, RT, AT, f"conv {name}", report)
    if y.C > cout:      # padded channels must be exactly zero (they feed the next layer)
        assert float(fm_to_cpu(y)[:, cout:].abs().max()) == 0.0


def test_conv_residuals_and_slices(report):
    ops = _ops()
    x = rnd16(randn(2, 128, 20, 36, seed=4))
    w = rnd16(randn(64, 64, 3, 3, seed=5) * 0.05)
    b = randn(64, seed=6) * 0.1
    r1 = rnd16(randn(2, 64, 20, 36, seed=7))
    r2 = rnd16(randn(2, 64, 20, 36, seed=8))
    ref = F.relu(F.conv2d(x[:, 64:], w, b, padding=1)) + r1 + r2
    xf = to_fm(x, ops)
    out = ops.FM.zeros(2, 20, 36, 192)
    ops.conv(xf.ch(64, 64), ops.pack_conv(w, b, stride=1, pad=1), out=out.ch(128, 64), act=ops.ACT_RELU,
             res=to_fm(r1, ops), res2=to_fm(r2, ops))
    full = fm_to_cpu(out)
    assert_close(full[:, 128:], ref, RT, AT, "conv slice-in/slice-out + 2 residuals", report)
    assert float(full[:, :128].abs().max()) == 0.0


def test_conv_cin_perm(report):
    ops = _ops()
    x = rnd16(randn(1, 128, 16, 32, seed=9))
    w = rnd16(randn(64, 128, 3, 3, seed=10) * 0.03)
    ref = F.conv2d(torch.cat([x[:, 64:], x[:, :64]], 1), w, None, padding=1)
    perm = list(range(64, 128)) + list(range(0, 64))
    y = ops.conv(to_fm(x, ops), ops.pack_conv(w, None, stride=1, pad=1, cin_perm=perm))
    assert_close(fm_to_cpu(y), ref, RT, AT, "conv cin_perm", report)


def test_conv_frames_as_slices(report):
    """Conv3d(1,3,3) over a (B,H,W,T*64) buffer == per-frame 2-D conv"""
    ops = _ops()
    B, T, H, W = 2, 4, 16, 32
    x = rnd16(randn(B, T * 64, H, W, seed=11))
    w = rnd16(randn(64, 64, 3, 3, seed=12) * 0.05)
    b = randn(64, seed=13) * 0.1
    src, dst = to_fm(x, ops), ops.FM.zeros(B, H, W, T * 64)
    pc = ops.pack_conv(w, b, stride=1, pad=1)
    for bi in range(B):
        ops.conv(src.as_slices(bi, T, 64), pc, out=dst.as_slices(bi, T, 64))
    got = fm_to_cpu(dst)
    ref = torch.cat([F.conv2d(x[:, t * 64:(t + 1) * 64], w, b, padding=1) for t in range(T)], 1)
    assert_close(got, ref, RT, AT, "conv over frame slices", report)


def test_conv_masked5x5(report):
    ops = _ops()
    x = rnd16(randn(1, 128, 17, 30, seed=14))
    w = rnd16(randn(256, 128, 5, 5, seed=15) * 0.02)
    b = randn(256, seed=16) * 0.1
    mask = torch.ones_like(w)
    mask[:, :, 2, 2:] = 0
    mask[:, :, 3:] = 0
    ref = F.conv2d(x, w * mask, b, padding=2)
    taps = [(dy, dx) for dy in range(5) for dx in range(5) if dy < 2 or (dy == 2 and dx < 2)]
    y = ops.conv(to_fm(x, ops), ops.pack_conv(w, b, stride=1, pad=2, taps=taps))
    assert_close(fm_to_cpu(y), ref, RT, AT, "masked 5x5 context conv", report)


@pytest.mark.parametrize("cin,cout", [(128, 128), (192, 192), (128, 64)])
def test_conv_pixel_shuffle(cin, cout, report):
    ops = _ops()
    x = rnd16(randn(1, cin, 9, 15, seed=17))
    w = rnd16(randn(cout * 4, cin, 3, 3, seed=18) * 0.03)
    b = randn(cout * 4, seed=19) * 0.1
    r = rnd16(randn(1, cout, 18, 30, seed=20))
    ref = F.leaky_relu(F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2), 0.01) + r
    y = ops.conv(to_fm(x, ops), ops.pack_conv(w, b, stride=1, pad=1, shuffle=True), act=ops.ACT_LRELU, slope=0.01,
                 res=to_fm(r, ops))
    assert_close(fm_to_cpu(y), ref, RT, AT, f"subpel conv {cin}->{cout}", report)


@pytest.mark.parametrize("inverse", [False, True])
def test_conv_gdn(inverse, report):
    ops = _ops()
    C = 128
    x = rnd16(randn(1, C, 12, 20, seed=21))
    gamma = rnd16(torch.rand(C, C, generator=torch.Generator().manual_seed(22)) * 0.02 + 0.1 * torch.eye(C))
    beta = torch.rand(C, generator=torch.Generator().manual_seed(23)) + 0.5
    r = rnd16(randn(1, C, 12, 20, seed=24))
    norm = F.conv2d(rnd16(x * x), gamma.view(C, C, 1, 1), beta)
    ref = x * (torch.sqrt(norm) if inverse else torch.rsqrt(norm)) + r
    xf = to_fm(x, ops)
    y = ops.conv(xf, ops.pack_conv(gamma.view(C, C, 1, 1), beta, stride=1, pad=0), square=True,
                 gdn=ops.GDN_INV if inverse else ops.GDN_FWD, aux=xf, res=to_fm(r, ops))
    assert_close(fm_to_cpu(y), ref, 3e-3, 3e-3, f"GDN inverse={inverse}", report)


def test_conv_f32_out_flow_residual_and_nchw(report):
    ops = _ops()
    x = rnd16(randn(1, 16, 17, 30, seed=25))
    w = rnd16(randn(2, 16, 7, 7, seed=26) * 0.05)
    b = randn(2, seed=27) * 0.1
    up = randn(1, 2, 17, 30, seed=28)
    ref = F.conv2d(x, w, b, padding=3) + up
    y = ops.conv(to_fm(x, ops), ops.pack_conv(w, b, stride=1, pad=3), res=to_fm(up, ops, Cpad=2, dtype=torch.float32),
                 out_dtype=torch.float32)
    assert y.f32 and y.C == 2
    assert_close(fm_to_cpu(y), ref, 1e-4, 1e-4, "7x7 16->2 fp32 out + fp32 residual", report)
    # featdown: 64 -> 3, clamp, planar fp32
    x = rnd16(randn(2, 64, 20, 36, seed=29))
    w = rnd16(randn(3, 64, 3, 3, seed=30) * 0.1)
    b = torch.tensor([0.5, 0.4, 0.6])
    ref = F.conv2d(x, w, b, padding=1).clamp(0, 1)
    out = torch.empty(2, 3, 20, 36, device="cuda")
    ops.conv(to_fm(x, ops), ops.pack_conv(w, b, stride=1, pad=1), act=ops.ACT_CLAMP01, nchw_out=out)
    assert_close(out.cpu(), ref, 1e-4, 1e-4, "featdown NCHW fp32 + clamp", report)


def test_conv_errors():
    ops = _ops()
    from tdvc_amd._lib import TdvcHipError
    x = to_fm(randn(1, 64, 16, 16), ops)
    pc = ops.pack_conv(randn(64, 64, 3, 3), None, stride=1, pad=1)
    bad = ops.FM.empty(1, 8, 8, 64)
    with pytest.raises(TdvcHipError):
        ops.conv(x, pc, out=bad)


# ---- shapes large enough for the persistent pipelined kernel (conv_mfma_v2) -------------------
V2_CASES = [
    # name, N, cin, cout, k, H, W
    ("v2_3x3_64_64", 1, 64, 64, 3, 70, 100),
    ("v2_3x3_64_64_tileedge", 1, 64, 64, 3, 64, 96),
    ("v2_3x3_128_64", 2, 128, 64, 3, 50, 70),
    ("v2_3x3_64_216", 1, 64, 216, 3, 48, 80),
    ("v2_3x3_128_128", 1, 128, 128, 3, 33, 65),
    ("v2_7x7_32_64", 1, 32, 64, 7, 40, 70),
    ("v2_3x3_40_64", 1, 40, 64, 3, 48, 64),       # Cin not a multiple of the 32-channel chunk
    ("v2_3x3_192_256", 1, 192, 256, 3, 40, 60),
    ("v2_3x3_64_64_many_tiles", 1, 64, 64, 3, 272, 480),   # > 256 tiles: several tiles per workgroup
]


@pytest.mark.parametrize("case", V2_CASES, ids=[c[0] for c in V2_CASES])
def test_conv_v2(case, report):
    ops = _ops()
    name, N, cin, cout, k, H, W = case
    x = rnd16(randn(N, cin, H, W, seed=41))
    w = rnd16(randn(cout, cin, k, k, seed=42) * (1.0 / (cin * k * k) ** 0.5))
    b = randn(cout, seed=43) * 0.1
    r1 = rnd16(randn(N, cout, H, W, seed=44))
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=k // 2), 0.1) + r1
    pc = ops.pack_conv(w, b, stride=1, pad=k // 2)
    assert pc.ck == 32
    y = ops.conv(to_fm(x, ops), pc, act=ops.ACT_LRELU, slope=0.1, res=to_fm(r1, ops))
    assert_close(fm_to_cpu(y, cout), ref, RT, AT, f"conv {name}", report)


def test_conv_v2_masked_shuffle_slices(report):
    ops = _ops()
    # masked 5x5 context conv on a latent-sized map
    x = rnd16(randn(1, 128, 68, 120, seed=45))
    w = rnd16(randn(256, 128, 5, 5, seed=46) * 0.02)
    b = randn(256, seed=47) * 0.1
    mask = torch.ones_like(w)
    mask[:, :, 2, 2:] = 0
    mask[:, :, 3:] = 0
    taps = [(dy, dx) for dy in range(5) for dx in range(5) if dy < 2 or (dy == 2 and dx < 2)]
    y = ops.conv(to_fm(x, ops), ops.pack_conv(w, b, stride=1, pad=2, taps=taps))
    assert_close(fm_to_cpu(y), F.conv2d(x, w * mask, b, padding=2), RT, AT, "v2 masked 5x5", report)
    # sub-pixel conv with residual at the shuffled resolution
    x = rnd16(randn(1, 128, 48, 64, seed=48))
    w = rnd16(randn(512, 128, 3, 3, seed=49) * 0.03)
    b = randn(512, seed=50) * 0.1
    r = rnd16(randn(1, 128, 96, 128, seed=51))
    ref = F.leaky_relu(F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2), 0.01) + r
    y = ops.conv(to_fm(x, ops), ops.pack_conv(w, b, stride=1, pad=1, shuffle=True), act=ops.ACT_LRELU, slope=0.01, res=to_fm(r, ops))
    assert_close(fm_to_cpu(y), ref, RT, AT, "v2 subpel 128->128", report)
    # frames as channel slices (Conv3d(1,3,3)) with residual buffer
    B, T, H, W = 1, 4, 48, 64
    x = rnd16(randn(B, T * 64, H, W, seed=52))
    rr = rnd16(randn(B, T * 64, H, W, seed=53))
    w = rnd16(randn(64, 64, 3, 3, seed=54) * 0.05)
    b = randn(64, seed=55) * 0.1
    src, dst, rb = to_fm(x, ops), ops.FM.zeros(B, H, W, T * 64), to_fm(rr, ops)
    ops.conv(src.as_slices(0, T, 64), ops.pack_conv(w, b, stride=1, pad=1), out=dst.as_slices(0, T, 64), res=rb.as_slices(0, T, 64))
    ref = torch.cat([F.conv2d(x[:, t * 64:(t + 1) * 64], w, b, padding=1) for t in range(T)], 1) + rr
    assert_close(fm_to_cpu(dst), ref, RT, AT, "v2 frame slices + residual", report)


@pytest.mark.parametrize("inverse", [False, True])
def test_conv_gdn_large(inverse, report):
    """GDN / iGDN at a size that takes the weight-stationary kernel (1x1, squared-input prologue,
    aux * (r)sqrt epilogue through LDS) with a residual"""
    ops = _ops()
    C, H, W = 128, 96, 112
    x = rnd16(randn(1, C, H, W, seed=61))
    gamma = rnd16(torch.rand(C, C, generator=torch.Generator().manual_seed(62)) * 0.02 + 0.1 * torch.eye(C))
    beta = torch.rand(C, generator=torch.Generator().manual_seed(63)) + 0.5
    r = rnd16(randn(1, C, H, W, seed=64))
    norm = F.conv2d(rnd16(x * x), gamma.view(C, C, 1, 1), beta)
    ref = x * (torch.sqrt(norm) if inverse else torch.rsqrt(norm)) + r
    xf = to_fm(x, ops)
    pc = ops.pack_conv(gamma.view(C, C, 1, 1), beta, stride=1, pad=0)
    assert pc.ck == 32
    y = ops.conv(xf, pc, square=True, gdn=ops.GDN_INV if inverse else ops.GDN_FWD, aux=xf, res=to_fm(r, ops))
    assert_close(fm_to_cpu(y), ref, 4e-3, 4e-3, f"GDN large inverse={inverse}", report)


def test_conv_1x1_large_and_shuffle_large(report):
    ops = _ops()
    # 1x1 256 -> 64 (multi-frame fusion) on the weight-stationary kernel
    x = rnd16(randn(1, 256, 96, 128, seed=65))
    w = rnd16(randn(64, 256, 1, 1, seed=66) * 0.06)
    b = randn(64, seed=67) * 0.1
    y = ops.conv(to_fm(x, ops), ops.pack_conv(w, b, stride=1, pad=0), act=ops.ACT_LRELU, slope=0.1)
    assert_close(fm_to_cpu(y), F.leaky_relu(F.conv2d(x, w, b), 0.1), RT, AT, "1x1 256->64 large", report)
    # sub-pixel conv 64-ch in, 128 -> 4*64 packed rows, on the weight-stationary kernel with residual
    x = rnd16(randn(1, 64, 96, 128, seed=68))
    w = rnd16(randn(256, 64, 3, 3, seed=69) * 0.04)
    b = randn(256, seed=70) * 0.1
    r = rnd16(randn(1, 64, 192, 256, seed=71))
    ref = F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2) + r
    y = ops.conv(to_fm(x, ops), ops.pack_conv(w, b, stride=1, pad=1, shuffle=True), res=to_fm(r, ops))
    assert_close(fm_to_cpu(y), ref, RT, AT, "subpel 64->64 large (v7)", report)
    # 128 -> 512 shuffle at a v3 size, LeakyReLU(0.01)
    x = rnd16(randn(1, 128, 68, 120, seed=72))
    w = rnd16(randn(512, 128, 3, 3, seed=73) * 0.03)
    b = randn(512, seed=74) * 0.1
    ref = F.leaky_relu(F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2), 0.01)
    y = ops.conv(to_fm(x, ops), ops.pack_conv(w, b, stride=1, pad=1, shuffle=True), act=ops.ACT_LRELU, slope=0.01)
    assert_close(fm_to_cpu(y), ref, RT, AT, "subpel 128->128 (v3)", report)


@pytest.mark.parametrize("H,W", [(64, 96), (34, 60), (270, 480)])
def test_conv_stride2_s2d(H, W, report):
    """3x3 stride-2 conv executed as a 2x2 conv over the space-to-depth view"""
    ops = _ops()
    x = rnd16(randn(2, 64, H, W, seed=75))
    w = rnd16(randn(128, 64, 3, 3, seed=76) * 0.04)
    b = randn(128, seed=77) * 0.1
    pc = ops.pack_conv(w, b, stride=2, pad=1)
    assert pc.s2d
    y = ops.conv(to_fm(x, ops), pc, act=ops.ACT_LRELU, slope=0.01)
    assert_close(fm_to_cpu(y), F.leaky_relu(F.conv2d(x, w, b, stride=2, padding=1), 0.01), RT, AT, f"s2d {H}x{W}", report)
    # fp32 output (the last analysis conv) takes the generic epilogue
    y = ops.conv(to_fm(x, ops), pc, out_dtype=torch.float32)
    assert y.f32
    assert_close(fm_to_cpu(y), F.conv2d(x, w, b, stride=2, padding=1), 1e-4, 1e-4, f"s2d fp32 out {H}x{W}", report)


# ---- 3x3 stride-1 convs at sizes that take the LDS-DMA double-buffered kernels: conv_mfma_v10 (Cin = 64: one wave per
# SIMD, 2 x 4 register tile with row reuse) and conv_mfma_v7 (Cin = 32, and Cin = 64 with v10 switched off) ---------
V7_CASES = [
    # name, N, cin, cout, H, W, act, n_res
    ("v7_64_64_ragged", 1, 64, 64, 100, 150, "lrelu", 1),          # partial tiles on both edges
    ("v7_64_64_exact", 1, 64, 64, 96, 128, "relu", 0),             # every tile full (counted-store path)
    ("v7_32_64", 1, 32, 64, 96, 100, "none", 2),                   # one 32-channel chunk per tile
    ("v7_64_216_batch2", 2, 64, 216, 90, 120, "lrelu", 0),         # Cout not a multiple of 64, batch in grid.z
    ("v7_64_128_many_tiles", 1, 64, 128, 272, 480, "relu", 1),     # several tiles per persistent workgroup
    ("v7_64_64_narrow", 1, 64, 64, 300, 40, "none", 0),            # 2 tile columns, 19 tile rows
]


@pytest.mark.parametrize("v10", [True, False], ids=["v10", "v7"])
@pytest.mark.parametrize("case", V7_CASES, ids=[c[0] for c in V7_CASES])
def test_conv_v7(case, v10, report):
    ops = _ops()
    name, N, cin, cout, H, W, act, n_res = case
    _enable("conv_row", False)           # the row-streaming kernel takes the Cin = 64 -> 64 k layers first in the product dispatch
    _enable("conv_v10", v10)
    try:
        _run_conv_dma_case(ops, name, N, cin, cout, H, W, act, n_res, "conv_mfma_v10" if (v10 and cin == 64) else "conv_mfma_v7", report)
    finally:
        _enable("conv_v10", True)
        _enable("conv_row", True)


def _run_conv_dma_case(ops, name, N, cin, cout, H, W, act, n_res, kernel, report):
    x = rnd16(randn(N, cin, H, W, seed=81))
    w = rnd16(randn(cout, cin, 3, 3, seed=82) * (1.0 / (cin * 9) ** 0.5))
    b = randn(cout, seed=83) * 0.1
    res = [rnd16(randn(N, cout, H, W, seed=84 + i)) for i in range(n_res)]
    ref = F.conv2d(x, w, b, padding=1)
    ref = {"lrelu": lambda t: F.leaky_relu(t, 0.1), "relu": F.relu, "none": lambda t: t}[act](ref)
    for r in res:
        ref = ref + r
    pc = ops.pack_conv(w, b, stride=1, pad=1)
    assert pc.ck == 32
    kw = dict(act={"lrelu": ops.ACT_LRELU, "relu": ops.ACT_RELU, "none": ops.ACT_NONE}[act], slope=0.1)
    if n_res > 0:
        kw["res"] = to_fm(res[0], ops)
    if n_res > 1:
        kw["res2"] = to_fm(res[1], ops)
    xf = to_fm(x, ops)
    y = ops.conv(xf, pc, **kw)
    assert ops.L.lib().tdvc_last_conv_kernel().decode() == kernel, (ops.L.lib().tdvc_last_conv_kernel(), kernel)
    assert_close(fm_to_cpu(y, cout), ref, RT, AT, f"conv {name} on {kernel}", report)
    # the DMA / barrier protocol must give the same bits on every launch
    first = y.t.clone()
    for _ in range(5):
        ops.conv(xf, pc, out=y, **kw)
        assert torch.equal(y.t, first), f"{name}: launch-to-launch mismatch"


@pytest.mark.parametrize("cin,cout,H,W", [(64, 128, 128, 160), (128, 128, 101, 131)])
def test_conv_1x1_stride2_large(cin, cout, H, W, report):
    """ResidualBlockWithStride skip conv (1x1, stride 2) at a size that takes the weight-stationary 1x1 kernel"""
    ops = _ops()
    x = rnd16(randn(2, cin, H, W, seed=91))
    w = rnd16(randn(cout, cin, 1, 1, seed=92) * 0.08)
    b = randn(cout, seed=93) * 0.1
    pc = ops.pack_conv(w, b, stride=2, pad=0)
    assert pc.ck == 32
    y = ops.conv(to_fm(x, ops), pc)
    assert_close(fm_to_cpu(y), F.conv2d(x, w, b, stride=2), RT, AT, f"1x1 s2 {cin}->{cout} {H}x{W}", report)


# ---- 3x3 stride-1 convs with Cin >= 128 (weights streamed per stage by LDS-DMA: conv_mfma_v11; v3 with it switched off)
V11_CASES = [
    # name, N, cin, cout, H, W, act, n_res
    ("v11_128_128_ragged", 1, 128, 128, 100, 150, "lrelu", 1),       # partial tiles on both edges, 4 stages per tile
    ("v11_128_64_exact", 1, 128, 64, 96, 128, "relu", 0),            # every tile full (counted-store path)
    ("v11_128_128_batch2", 2, 128, 128, 90, 120, "lrelu", 2),        # batch in grid.z, two residuals
    ("v11_192_64", 1, 192, 64, 96, 100, "none", 0),                  # 6 stages per tile
    ("v11_128_128_many_tiles", 1, 128, 128, 272, 480, "lrelu", 1),   # several tiles per persistent workgroup
    ("v11_256_64_narrow", 1, 256, 64, 300, 40, "none", 1),           # 8 stages, 2 tile columns
]


@pytest.mark.parametrize("v11", [True, False], ids=["v11", "v3"])
@pytest.mark.parametrize("case", V11_CASES, ids=[c[0] for c in V11_CASES])
def test_conv_v11(case, v11, report):
    ops = _ops()
    name, N, cin, cout, H, W, act, n_res = case
    _enable("conv_row", False)           # the row-streaming kernel (below) takes the Cin = 128 layers first in the product dispatch
    _enable("conv_v11", v11)
    try:
        _run_conv_dma_case(ops, name, N, cin, cout, H, W, act, n_res, "conv_mfma_v11" if v11 else "conv_mfma_v3", report)
    finally:
        _enable("conv_v11", True)
        _enable("conv_row", True)


# ---- 3x3 stride-1 convs with Cin = 128, Cout a multiple of 128 (weights in registers, row streaming: conv_row)
ROW_CASES = [
    # name, N, cin, cout, H, W, act, n_res
    ("row_128_128_ragged", 1, 128, 128, 100, 150, "lrelu", 1),       # a partial strip on the right, ragged row segments
    ("row_128_128_batch2", 2, 128, 128, 90, 120, "lrelu", 1),        # batch in the job walk
    ("row_128_256_two_blocks", 1, 128, 256, 96, 128, "none", 0),     # two cout blocks share the workgroup slots
    ("row_128_128_many_jobs", 1, 128, 128, 272, 480, "lrelu", 1),    # the coders' H/4 size: several jobs per persistent workgroup
    ("row_128_128_narrow", 1, 128, 128, 300, 40, "relu", 0),         # two strips, one of 8 columns; long segments
    ("row_128_128_min_rows", 1, 128, 128, 16, 520, "lrelu", 0),      # the fewest rows the kernel takes (one segment of 16)
    ("row_128_384_three_blocks", 1, 128, 384, 64, 160, "none", 1),   # a cout-block count that does not divide the 32 slots of an XCD
]


ROW_CASES += [
    ("row64_64_64_ragged", 1, 64, 64, 100, 150, "lrelu", 1),         # 64-column strips: 2 full + one of 22 columns
    ("row64_64_64_slices", 3, 64, 64, 70, 260, "relu", 0),           # batch of three, 4 full strips + 4 columns
    ("row64_64_128_two_blocks", 1, 64, 128, 96, 128, "none", 1),
    ("row64_64_64_1080p_rows", 1, 64, 64, 1088, 64, "lrelu", 1),     # one strip, 1088 rows: 256 runs of 4.25 rows
    ("row_128_64_wide", 1, 128, 64, 100, 150, "lrelu", 0),           # Cin 128 -> 64: 64-column strips, 17-piece rows in a 6-row ring
    ("row_128_64_wide_batch_res", 2, 128, 64, 90, 200, "none", 1),
    ("row_128_192_wide_three_blocks", 1, 128, 192, 96, 136, "relu", 0),
]


@pytest.mark.parametrize("case", ROW_CASES, ids=[c[0] for c in ROW_CASES])
def test_conv_row(case, report):
    ops = _ops()
    name, N, cin, cout, H, W, act, n_res = case
    _run_conv_dma_case(ops, name, N, cin, cout, H, W, act, n_res, "conv_row", report)


@pytest.mark.parametrize("cq,H,W,nres", [(128, 96, 128, 1), (64, 100, 150, 0), (64, 90, 120, 1)])
def test_conv_row_pixel_shuffle(cq, H, W, nres, report):
    """sub-pixel convs 128 -> 4 cq with the PixelShuffle(2) store on the row-streaming kernel: a 128-channel block of the packed rows
    is one sub-pixel (cq = 128) or two (cq = 64)"""
    ops = _ops()
    x = rnd16(randn(1, 128, H, W, seed=91))
    w = rnd16(randn(4 * cq, 128, 3, 3, seed=92) * 0.03)
    b = randn(4 * cq, seed=93) * 0.1
    rs = [rnd16(randn(1, cq, 2 * H, 2 * W, seed=94 + i)) for i in range(nres)]
    ref = F.leaky_relu(F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2), 0.01)
    for r_ in rs:
        ref = ref + r_
    kw = dict(act=ops.ACT_LRELU, slope=0.01)
    if nres > 0:
        kw["res"] = to_fm(rs[0], ops)
    xf = to_fm(x, ops)
    pc = ops.pack_conv(w, b, stride=1, pad=1, shuffle=True)
    y = ops.conv(xf, pc, **kw)
    kern = ops.L.lib().tdvc_last_conv_kernel().decode()
    assert kern == "conv_row", kern
    assert_close(fm_to_cpu(y), ref, RT, AT, f"subpel 128->{cq} @{H}x{W} on {kern}", report)
    first = y.t.clone()
    for _ in range(4):
        ops.conv(xf, pc, out=y, **kw)
        assert torch.equal(y.t, first), "launch-to-launch mismatch"


@pytest.mark.parametrize("N,cout,H,W", [(1, 128, 270, 480), (2, 128, 192, 264), (1, 256, 256, 264)])
def test_conv_row_stride2(N, cout, H, W, report):
    """3x3 stride-2 convs with 64 input channels on the row-streaming kernel's space-to-depth geometry (512-byte ring pixels gathered from
    two image rows, 18 of 32 fragments): against torch, and against conv_mfma_v3's space-to-depth form (same packed weights)"""
    ops = _ops()
    x = rnd16(randn(N, 64, H, W, seed=75))
    w = rnd16(randn(cout, 64, 3, 3, seed=76) * 0.04)
    b = randn(cout, seed=77) * 0.1
    pc = ops.pack_conv(w, b, stride=2, pad=1)
    assert pc.s2d
    xf = to_fm(x, ops)
    y = ops.conv(xf, pc, act=ops.ACT_LRELU, slope=0.1)
    kern = ops.L.lib().tdvc_last_conv_kernel().decode()
    assert kern == "conv_row(s2d)", kern
    ref = F.leaky_relu(F.conv2d(x, w, b, stride=2, padding=1), 0.1)
    assert_close(fm_to_cpu(y), ref, RT, AT, f"stride 2 64->{cout} @{N}x{H}x{W} on {kern}", report)
    first = y.t.clone()
    for _ in range(4):
        ops.conv(xf, pc, out=y, act=ops.ACT_LRELU, slope=0.1)
        assert torch.equal(y.t, first), "launch-to-launch mismatch"
    _enable("conv_row", False)
    try:
        y3 = ops.conv(xf, pc, act=ops.ACT_LRELU, slope=0.1)
        assert ops.L.lib().tdvc_last_conv_kernel().decode() == "conv_mfma_v3(s2d)"
    finally:
        _enable("conv_row", True)
    a, c = fm_to_cpu(y), fm_to_cpu(y3)
    report(f"conv_row(s2d) vs conv_mfma_v3(s2d): {float((a != c).float().mean()):.2e} of the outputs differ, max |d| {float((a - c).abs().max()):.2e}")
    assert float((a - c).abs().max()) <= 4e-3


def test_conv_row_equals_v11_arithmetic(report):
    """the two kernels round alike (fp16 conv result, packed-fp16 activation, fp16 residual add): on one layer they differ only by
    the fp32 summation order inside the contraction"""
    ops = _ops()
    x = rnd16(randn(1, 128, 136, 240, seed=71))
    w = rnd16(randn(128, 128, 3, 3, seed=72) * 0.03)
    b = randn(128, seed=73) * 0.1
    r = rnd16(randn(1, 128, 136, 240, seed=74))
    pc = ops.pack_conv(w, b, stride=1, pad=1)
    kw = dict(act=ops.ACT_LRELU, slope=0.01, res=to_fm(r, ops))
    y_row = ops.conv(to_fm(x, ops), pc, **kw)
    assert ops.L.lib().tdvc_last_conv_kernel().decode() == "conv_row"
    _enable("conv_row", False)
    try:
        y_v11 = ops.conv(to_fm(x, ops), pc, **kw)
        assert ops.L.lib().tdvc_last_conv_kernel().decode() == "conv_mfma_v11"
    finally:
        _enable("conv_row", True)
    a, c = fm_to_cpu(y_row), fm_to_cpu(y_v11)
    neq = float((a != c).float().mean())
    report(f"conv_row vs conv_mfma_v11, 128->128 @136x240 + residual: {neq:.2e} of the outputs differ, max |d| {float((a - c).abs().max()):.2e}")
    assert neq < 0.05 and float((a - c).abs().max()) <= 4e-3


@pytest.mark.parametrize("v11", [True, False], ids=["v11", "v3"])
@pytest.mark.parametrize("cout,H,W,nres", [(128, 96, 128, 1), (64, 100, 150, 0), (128, 90, 120, 2)])
def test_conv_v11_pixel_shuffle(cout, H, W, nres, v11, report):
    """sub-pixel convs 128 -> 4 * cout with the PixelShuffle(2) store (the coders' ResidualBlockUpsample and final g_s layers)"""
    ops = _ops()
    _enable("conv_row", False)
    _enable("conv_v11", v11)
    try:
        x = rnd16(randn(1, 128, H, W, seed=91))
        w = rnd16(randn(4 * cout, 128, 3, 3, seed=92) * 0.03)
        b = randn(4 * cout, seed=93) * 0.1
        rs = [rnd16(randn(1, cout, 2 * H, 2 * W, seed=94 + i)) for i in range(nres)]
        ref = F.leaky_relu(F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2), 0.01)
        for r_ in rs:
            ref = ref + r_
        kw = dict(act=ops.ACT_LRELU, slope=0.01)
        if nres > 0:
            kw["res"] = to_fm(rs[0], ops)
        if nres > 1:
            kw["res2"] = to_fm(rs[1], ops)
        xf = to_fm(x, ops)
        pc = ops.pack_conv(w, b, stride=1, pad=1, shuffle=True)
        y = ops.conv(xf, pc, **kw)
        kern = ops.L.lib().tdvc_last_conv_kernel().decode()
        assert kern == ("conv_mfma_v11" if v11 else "conv_mfma_v3"), kern
        assert_close(fm_to_cpu(y), ref, RT, AT, f"subpel 128->{cout} @{H}x{W} on {kern}", report)
        first = y.t.clone()
        for _ in range(4):
            ops.conv(xf, pc, out=y, **kw)
            assert torch.equal(y.t, first), "launch-to-launch mismatch"
    finally:
        _enable("conv_v11", True)
        _enable("conv_row", True)


def test_conv_c8_equals_direct_kernel(report):
    """the first-layer kernel (conv_c8: persistent waves, weights in registers, B fragments straight from L1) against the
    direct kernel on the same layer: same MFMA chain per output element, so the results are bit-equal"""
    import ctypes

    from tdvc_amd import _lib
    ops = _ops()
    fn = _lib.lib().tdvc_debug_enable_conv_c8
    fn.argtypes = [ctypes.c_int]
    fn.restype = None
    x = rnd16(randn(2, 3, 96, 200, seed=41).abs())
    w = rnd16(randn(64, 3, 3, 3, seed=42) * 0.2)
    b = randn(64, seed=43) * 0.1
    pc = ops.pack_conv(w, b, stride=1, pad=1)
    xf = to_fm(x, ops)
    names, outs = [], []
    try:
        for on in (1, 0):
            fn(on)
            y = ops.conv(xf, pc, act=ops.ACT_LRELU, slope=0.01)
            names.append(ops.L.lib().tdvc_last_conv_kernel().decode())
            outs.append(fm_to_cpu(y))
    finally:
        fn(1)
    report(f"first layer 3x3 8->64 @96x200 x2: kernels {names}, max |diff| {float((outs[0] - outs[1]).abs().max()):.3e}")
    assert names[0] == "conv_c8" and names[1] != "conv_c8"
    assert torch.equal(outs[0], outs[1])
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.01)
    assert_close(outs[0], ref, RT, AT, "conv_c8 vs torch", report)


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("with_res", [True, False])
def test_gdn128_equals_conv_path(inverse, with_res, report):
    """the one-pass GDN kernel (conv_gdn128: x read once, gamma in registers, norm rounded to fp16 like the conv path) against
    the same call on conv_mfma_v5 (1x1 conv over x^2 + GDN epilogue): bit-equal, on a batch of two images whose pixel count
    is not a multiple of the 32-pixel tile (tiles straddle rows and the image boundary) and on channel-slice views"""
    import ctypes

    from tdvc_amd import _lib
    ops = _ops()
    fn = _lib.lib().tdvc_debug_enable_gdn128
    fn.argtypes = [ctypes.c_int]
    fn.restype = None
    C, H, W = 128, 95, 101
    x = rnd16(randn(2, 2 * C, H, W, seed=71))
    gamma = rnd16(torch.rand(C, C, generator=torch.Generator().manual_seed(72)) * 0.02 + 0.1 * torch.eye(C))
    beta = torch.rand(C, generator=torch.Generator().manual_seed(73)) + 0.5
    r = rnd16(randn(2, C, H, W, seed=74))
    xf = to_fm(x, ops).ch(C, C)                              # a channel-slice view: pixel stride 2C
    rf = to_fm(r, ops) if with_res else None
    pc = ops.pack_conv(gamma.view(C, C, 1, 1), beta, stride=1, pad=0)
    names, outs = [], []
    try:
        for on in (1, 0):
            fn(on)
            y = ops.conv(xf, pc, square=True, gdn=ops.GDN_INV if inverse else ops.GDN_FWD, aux=xf, res=rf)
            names.append(ops.L.lib().tdvc_last_conv_kernel().decode())
            outs.append(fm_to_cpu(y))
    finally:
        fn(1)
    report(f"GDN inverse={inverse} res={with_res} @2x{H}x{W}: kernels {names}, max |diff| {float((outs[0] - outs[1]).abs().max()):.3e}")
    assert names[0] == "gdn128" and names[1] != "gdn128"
    assert torch.equal(outs[0], outs[1])
    xs = x[:, C:]
    norm = F.conv2d(rnd16(xs * xs), gamma.view(C, C, 1, 1), beta)
    ref = xs * (torch.sqrt(norm) if inverse else torch.rsqrt(norm)) + (r if with_res else 0)
    assert_close(outs[0], ref, 4e-3, 4e-3, f"gdn128 inverse={inverse}", report)


def test_conv_bcast_add_act_fused(report):
    """Bottleneck3D's temporal conv + broadcast add + LeakyReLU (pnet.py:304-314) as one launch (tdvc_conv_desc::bcast_T on
    conv_mfma_v5) against the two-launch form (1x1 conv, then tdvc_bcast_add_act): bit-equal, in place over a 4-slice buffer,
    ragged tile edges, batch 2"""
    ops = _ops()
    B, H, W = 2, 90, 101
    s0 = rnd16(randn(B, 256, H, W, seed=81))
    w = rnd16(randn(64, 192, 1, 1, seed=82) * 0.07)
    pc = ops.pack_conv(w, None, stride=1, pad=0)
    a = to_fm(s0, ops)
    tm = ops.conv(a.ch(0, 192), pc)
    ops.bcast_add_act(a, tm, 4, 0.1)
    b = to_fm(s0, ops)
    ops.conv(b.ch(0, 192), pc, out=b.ch(0, 64), bcast_T=4, bcast_slope=0.1)
    name = ops.L.lib().tdvc_last_conv_kernel().decode()
    ga, gb = fm_to_cpu(a), fm_to_cpu(b)
    report(f"temporal conv + broadcast add fused: kernel {name}, max |diff| vs two launches {float((ga - gb).abs().max()):.3e}")
    assert name == "conv_mfma_v5(bcast)"
    assert torch.equal(ga, gb)
    t = F.conv2d(s0[:, :192], w)
    ref = torch.cat([F.leaky_relu(s0[:, 64 * k:64 * k + 64] + rnd16(t), 0.1) for k in range(4)], 1)
    assert_close(gb, ref, RT, AT, "fused temporal conv + broadcast add", report)


@pytest.mark.parametrize("cin,act,with_res,B,H,W", [(128, False, False, 1, 96, 160), (256, True, True, 2, 90, 101), (128, False, True, 1, 136, 240)])
def test_conv_fused_channel_sums(cin, act, with_res, B, H, W, report):
    """tdvc_conv_desc::chan_sum: the 1x1 convs in front of the three full-resolution SELayers (OffsetGen.feat_fusion_, LoopFilter.feat_fusion,
    FeatureFix.featfusion2) leave the per-channel sums of the values they store -- the SELayer's average pool (inflate.py:204) -- next to y.
    The output is bit-equal to the plain launch, the sums are those of the stored fp16 values (ragged tile edges, batch 2, activation and
    residual in the epilogue), and se_gate() on them equals se_gate() on a separate pass over y to fp32 summation order."""
    ops = _ops()
    x = rnd16(randn(B, cin, H, W, seed=91))
    w = rnd16(randn(64, cin, 1, 1, seed=92) * (1.0 / cin ** 0.5))
    bias = randn(64, seed=93) * 0.1
    r = rnd16(randn(B, 64, H, W, seed=94)) if with_res else None
    pc = ops.pack_conv(w, bias, stride=1, pad=0)
    kw = dict(act=ops.ACT_LRELU if act else ops.ACT_NONE, slope=0.1 if act else 0.0, res=to_fm(r, ops) if with_res else None)
    xf = to_fm(x, ops)
    plain = ops.conv(xf, pc, **kw)
    sums = []
    fused = ops.conv(xf, pc, chan_sum=sums, **kw)
    name = ops.L.lib().tdvc_last_conv_kernel().decode()
    assert name == "conv_mfma_v5" and len(sums) == 1
    part, rows = sums[0]
    assert tuple(part.shape) == (B, rows, 64)
    y = fm_to_cpu(fused)
    assert torch.equal(y, fm_to_cpu(plain))
    want = y.double().sum(dim=(2, 3))
    got = part.double().sum(dim=1).cpu()
    err = float(((got - want).abs() / (want.abs() + 1.0)).max())
    report(f"fused channel sums {cin}->64 @{B}x{H}x{W}: {rows} rows per image, max rel error vs the sum of the stored values {err:.2e}")
    assert err < 2e-6
    p = ops.SEParams(randn(4, 64, seed=95).cuda() * 0.2, randn(4, seed=96).cuda() * 0.1, randn(64, 4, seed=97).cuda() * 0.5, randn(64, seed=98).cuda() * 0.1, 64, 4)
    g0, g1 = ops.se_gate(fused, p), ops.se_gate(fused, p, partial=sums[0])
    assert float((g0 - g1).abs().max()) < 1e-6
    # a conv whose kernel has no fused sum leaves the list empty (the caller falls back to tdvc_channel_sum)
    none = []
    ops.conv(to_fm(rnd16(randn(1, 64, 96, 96, seed=99)), ops), ops.pack_conv(rnd16(randn(64, 64, 3, 3, seed=100) * 0.04), None, stride=1, pad=1), chan_sum=none)
    assert none == []


N16_CASES = [
    # name, N, cin, cout, k, H, W: the few-output-channel layers at >= 8192 output pixels (conv_n16: 16-row MFMA tiles)
    ("n16_7x7_8_32", 1, 8, 32, 7, 90, 140),        # SPyNet basic module, first conv: two cout blocks, four taps per k-step
    ("n16_7x7_32_16", 2, 32, 16, 7, 70, 130),      # one tap per k-step, ragged tile edges, batch 2
    ("n16_3x3_64_16", 1, 64, 16, 3, 96, 96),       # half a tap per k-step
    ("n16_3x3_8_24", 1, 8, 24, 3, 100, 90),        # four taps per k-step, 9 taps (a padded half step and a k-step past the last tap), 24 of 32 channels
    ("n16_3x3_16_8", 1, 16, 8, 3, 90, 100),        # two taps per k-step
]


@pytest.mark.parametrize("case", N16_CASES, ids=[c[0] for c in N16_CASES])
def test_conv_n16(case, report):
    import ctypes

    from tdvc_amd import _lib
    ops = _ops()
    name, N, cin, cout, k, H, W = case
    fn = _lib.lib().tdvc_debug_enable_conv_n16
    fn.argtypes = [ctypes.c_int]
    fn.restype = None
    x = rnd16(randn(N, cin, H, W, seed=51))
    w = rnd16(randn(cout, cin, k, k, seed=52) * (1.0 / (cin * k * k) ** 0.5))
    b = randn(cout, seed=53) * 0.1
    ref = F.relu(F.conv2d(x, w, b, padding=k // 2))
    pc = ops.pack_conv(w, b, stride=1, pad=k // 2)
    xf = to_fm(x, ops)
    names, outs = [], []
    try:
        for on in (1, 0):
            fn(on)
            y = ops.conv(xf, pc, act=ops.ACT_RELU)
            names.append(ops.L.lib().tdvc_last_conv_kernel().decode())
            outs.append(fm_to_cpu(y))
    finally:
        fn(1)
    report(f"{name}: kernels {names}, max |n16 - direct| {float((outs[0] - outs[1]).abs().max()):.3e}")
    assert names[0] == "conv_n16" and names[1] != "conv_n16"
    assert_close(outs[0][:, :cout], ref, RT, AT, f"conv_n16 {name}", report)
    if outs[0].shape[1] > cout:      # padded channels must be exactly zero (they feed the next layer)
        assert float(outs[0][:, cout:].abs().max()) == 0.0


def test_conv_n16_flow_head_and_featdown(report):
    """the two heads at sizes the few-channel kernel takes: SPyNet's 7x7 16 -> 2 with fp32 output + fp32 residual
    (flownet.py:227 + the flow update) and FeatureFix.featdown 3x3 64 -> 3, clamp, planar fp32 (pnet.py:262)"""
    ops = _ops()
    x = rnd16(randn(1, 16, 80, 120, seed=25))
    w = rnd16(randn(2, 16, 7, 7, seed=26) * 0.05)
    b = randn(2, seed=27) * 0.1
    up = randn(1, 2, 80, 120, seed=28)
    ref = F.conv2d(x, w, b, padding=3) + up
    y = ops.conv(to_fm(x, ops), ops.pack_conv(w, b, stride=1, pad=3), res=to_fm(up, ops, Cpad=2, dtype=torch.float32), out_dtype=torch.float32)
    assert ops.L.lib().tdvc_last_conv_kernel() == b"conv_n16" and y.f32 and y.C == 2
    assert_close(fm_to_cpu(y), ref, 1e-4, 1e-4, "conv_n16 7x7 16->2 fp32 out + fp32 residual", report)
    x = rnd16(randn(2, 64, 90, 100, seed=29))
    w = rnd16(randn(3, 64, 3, 3, seed=30) * 0.1)
    b = torch.tensor([0.5, 0.4, 0.6])
    ref = F.conv2d(x, w, b, padding=1).clamp(0, 1)
    out = torch.empty(2, 3, 90, 100, device="cuda")
    ops.conv(to_fm(x, ops), ops.pack_conv(w, b, stride=1, pad=1), act=ops.ACT_CLAMP01, nchw_out=out)
    assert ops.L.lib().tdvc_last_conv_kernel() == b"conv_n16"
    assert_close(out.cpu(), ref, 1e-4, 1e-4, "conv_n16 featdown NCHW fp32 + clamp", report)
