"""tdvc_conv_pair: two 3x3 64->64 convs in one launch (Res_Block, main/utils/utils.py:52-56; the LeakyReLU pairs of
main/model/pnet.py:132-166) against (a) a torch fp32 reference with the intermediate rounded to fp16 like the stored map
and (b) the same pair as two tdvc_conv2d launches."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import assert_close, fm_to_cpu, randn, rnd16, to_fm

pytestmark = pytest.mark.gpu


def _ops():
    from tdvc_amd import ops
    return ops


@pytest.fixture(params=[2, 4], ids=["strips30", "strips62"])
def geom(request):
    """both strip geometries of the kernel (conv_pair.hip, PairGeo<NCB>) instead of the per-width choice"""
    import ctypes

    from tdvc_amd import _lib
    fn = _lib.lib().tdvc_debug_set_pair_geometry
    fn.argtypes, fn.restype = [ctypes.c_int], None
    fn(request.param)
    yield request.param
    fn(0)


def _weights(seed):
    w1 = rnd16(randn(64, 64, 3, 3, seed=seed) * 0.05)
    w2 = rnd16(randn(64, 64, 3, 3, seed=seed + 1) * 0.05)
    b1, b2 = randn(64, seed=seed + 2) * 0.1, randn(64, seed=seed + 3) * 0.1
    return w1, b1, w2, b2


def _act(v, kind, slope):
    return v if kind == "none" else (F.relu(v) if kind == "relu" else F.leaky_relu(v, slope))


def _ref(x, w1, b1, w2, b2, act1, act2, slope, add, res2):
    t = rnd16(_act(rnd16(F.conv2d(x, w1, b1, padding=1)), act1, slope))
    y = rnd16(_act(rnd16(F.conv2d(t, w2, b2, padding=1)), act2, slope))
    if add:
        y = rnd16(y + x)
    if res2 is not None:
        y = rnd16(y + res2)
    return y


def test_pair_weight_packing_matches_c_packer():
    ops = _ops()
    w1, b1, w2, b2 = _weights(5)
    pp = ops.pack_conv_pair(w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda())
    lib = ops.L.lib()
    nb = lib.tdvc_conv_pair_packed_bytes()
    assert nb == pp.w.numel() * 2
    dst = np.zeros(nb // 2, dtype=np.uint16)
    a, b = np.ascontiguousarray(w1.numpy(), dtype=np.float32), np.ascontiguousarray(w2.numpy(), dtype=np.float32)
    ops.L.check(lib.tdvc_pack_conv_pair_weights(a.ctypes.data, b.ctypes.data, dst.ctypes.data), "pack")
    assert np.array_equal(dst, pp.w.cpu().view(torch.int16).numpy().view(np.uint16).reshape(-1))
    assert torch.equal(pp.bias.cpu(), torch.cat([b1, b2]))


CASES = [
    # N, H, W, act1, act2, add_input, res2, views
    (1, 96, 120, "relu", "none", True, False, False),       # Res_Block, 4 full strips
    (1, 100, 131, "relu", "none", True, False, False),      # ragged: last strip 11 px, rows not a multiple of the segments
    (2, 64, 160, "relu", "none", True, True, True),         # MCNet's last block: + res2, x / y / res2 as slices of wider buffers
    (1, 128, 90, "lrelu", "lrelu", False, False, False),    # LeakyReLU pair without identity
    (4, 48, 192, "none", "relu", False, True, False),       # batch of slices, activation only after conv2
    (1, 272, 480, "relu", "none", True, False, False),      # a real geometry of the model (quarter resolution)
    (5, 16, 1920, "relu", "none", True, False, False),      # 320 jobs on 256 workgroups: several jobs per workgroup, uneven XCD ranges
    (1, 33, 250, "lrelu", "none", True, True, False),       # odd everything: 9 strips (last 10 px), segments of 17 + 16 rows
]


@pytest.mark.parametrize("N,H,W,act1,act2,add,with_res2,views", CASES)
def test_conv_pair_vs_reference(N, H, W, act1, act2, add, with_res2, views, geom, report):
    ops = _ops()
    w1, b1, w2, b2 = _weights(11)
    x = rnd16(randn(N, 64, H, W, seed=21))
    r2 = rnd16(randn(N, 64, H, W, seed=22)) if with_res2 else None
    ref = _ref(x, w1, b1, w2, b2, act1, act2, 0.1, add, r2)
    A = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU}
    if views:
        xb = ops.FM.empty(N, H, W, 192, device="cuda"); xb.t.zero_()
        yb = ops.FM.empty(N, H, W, 256, device="cuda"); yb.t.fill_(7.0)
        rb = ops.FM.empty(N, H, W, 128, device="cuda")
        xf, yf, rf = xb.ch(64, 64), yb.ch(128, 64), rb.ch(64, 64)
        ops.from_nchw(x.cuda(), out=xf)
        ops.from_nchw(r2.cuda(), out=rf)
    else:
        xf, yf = to_fm(x, ops), None
        rf = to_fm(r2, ops) if with_res2 else None
    assert ops.conv_pair_supported(xf, yf, rf)
    pp = ops.pack_conv_pair(w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda())
    kw = dict(act1=A[act1], slope1=0.1, act2=A[act2], slope2=0.1, add_input=add, res2=rf)
    y = ops.conv_pair(xf, pp, out=yf, **kw)
    assert_close(fm_to_cpu(y), ref, 4e-3, 4e-3, f"conv_pair N={N} {H}x{W} {act1}/{act2} add={add} res2={with_res2}", report)
    if views:       # nothing outside the output slice was touched
        assert bool((yb.t[..., :128] == 7.0).all()) and bool((yb.t[..., 192:] == 7.0).all())
    # against the same pair as two launches (same fp16 intermediate): only the fp32 summation order differs
    pc1, pc2 = ops.pack_conv(w1, b1, stride=1, pad=1), ops.pack_conv(w2, b2, stride=1, pad=1)
    t = ops.conv(xf, pc1, act=A[act1], slope=0.1)
    y2 = ops.conv(t, pc2, act=A[act2], slope=0.1, res=xf if add else rf, res2=rf if add else None)
    d = (fm_to_cpu(y).double() - fm_to_cpu(y2).double()).abs()
    report(f"conv_pair vs two launches N={N} {H}x{W}: max|d|={float(d.max()):.3e}, differing {float((d > 0).double().mean()):.4f}")
    assert float(d.max()) <= 2e-2 and float((d > 0).double().mean()) < 0.2
    first = y.t.clone()
    for _ in range(3):
        ops.conv_pair(xf, pp, out=y, **kw)
        assert torch.equal(y.t, first), "launch-to-launch mismatch"


def test_conv_pair_random_shapes(geom, report):
    """seeded sweep over geometries, activations and residual options against the same pair as two tdvc_conv2d launches
    (strips / segments / multi-job walks are all functions of N, H, W)"""
    import random
    ops = _ops()
    rng = random.Random(2024)
    A = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU}
    w1, b1, w2, b2 = _weights(41)
    pp = ops.pack_conv_pair(w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda())
    pc1, pc2 = ops.pack_conv(w1, b1, stride=1, pad=1), ops.pack_conv(w2, b2, stride=1, pad=1)
    worst = 0.0
    for it in range(14):
        N = rng.choice([1, 1, 2, 3])
        H = rng.randint(16, 150)
        W = max(rng.randint(31, 420), -(-8192 // H))
        act1, act2 = rng.choice(["relu", "lrelu", "none"]), rng.choice(["none", "lrelu", "relu"])
        add, with_r2 = rng.random() < 0.6, rng.random() < 0.4
        x = to_fm(rnd16(randn(N, 64, H, W, seed=100 + it)), ops)
        r2 = to_fm(rnd16(randn(N, 64, H, W, seed=200 + it)), ops) if with_r2 else None
        assert ops.conv_pair_supported(x, None, r2), (N, H, W)
        y = ops.conv_pair(x, pp, act1=A[act1], slope1=0.2, act2=A[act2], slope2=0.05, add_input=add, res2=r2)
        t = ops.conv(x, pc1, act=A[act1], slope=0.2)
        res = [m for m in ((x if add else None), r2) if m is not None]
        y2 = ops.conv(t, pc2, act=A[act2], slope=0.05, res=res[0] if res else None, res2=res[1] if len(res) > 1 else None)
        d = float((fm_to_cpu(y).double() - fm_to_cpu(y2).double()).abs().max())
        worst = max(worst, d)
        assert d <= 2e-2, (it, N, H, W, act1, act2, add, with_r2, d)
    report(f"conv_pair random sweep (14 shapes): max |pair - two launches| = {worst:.3e}")


@pytest.mark.parametrize("N,H,W,r2", [(1, 33, 250, True), (4, 48, 192, True), (1, 100, 131, False)])
def test_conv_pair_launches_are_reproducible(N, H, W, r2, geom, report):
    """150 launches of the shapes most exposed to a synchronisation hole (short row segments, a ragged last strip, the second
    residual slowing the conv2 waves) must all equal the first: with one barrier per FOUR row steps and a 16-row input ring
    the 33x250 case lost a write-after-read race on every few launches (tools/stress_pair.py; conv_pair.hip, `BI`)"""
    ops = _ops()
    w1, b1, w2, b2 = _weights(51)
    pp = ops.pack_conv_pair(w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda())
    x = to_fm(rnd16(randn(N, 64, H, W, seed=52)), ops)
    res2 = to_fm(rnd16(randn(N, 64, H, W, seed=53)), ops) if r2 else None
    kw = dict(act1=ops.ACT_LRELU, slope1=0.1, res2=res2)
    first = ops.conv_pair(x, pp, **kw).t.clone()
    bad = sum(0 if torch.equal(ops.conv_pair(x, pp, **kw).t, first) else 1 for _ in range(150))
    report(f"conv_pair {N}x{H}x{W} res2={r2}: {bad} of 150 launches differ from the first")
    assert bad == 0


@pytest.mark.parametrize("N,H,W", [(1, 100, 131), (2, 64, 160), (1, 33, 250), (1, 136, 240)])
def test_conv_pair_strip_geometries_bit_equal(N, H, W, report):
    """30- and 62-column strips feed every accumulator its (tap, channel chunk) terms in the same order: the two kernels agree bit for bit
    (ragged last strips, both residual forms, LeakyReLU pair)"""
    import ctypes

    from tdvc_amd import _lib
    ops = _ops()
    fn = _lib.lib().tdvc_debug_set_pair_geometry
    fn.argtypes, fn.restype = [ctypes.c_int], None
    w1, b1, w2, b2 = _weights(61)
    pp = ops.pack_conv_pair(w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda())
    x = to_fm(rnd16(randn(N, 64, H, W, seed=62)), ops)
    r2 = to_fm(rnd16(randn(N, 64, H, W, seed=63)), ops)
    try:
        for kw in (dict(act1=ops.ACT_RELU, add_input=True), dict(act1=ops.ACT_LRELU, slope1=0.1, act2=ops.ACT_LRELU, slope2=0.1, add_input=False, res2=r2),
                   dict(act1=ops.ACT_RELU, add_input=True, res2=r2)):
            outs = []
            for g in (2, 4):
                fn(g)
                outs.append(ops.conv_pair(x, pp, **kw).t.clone())
            assert torch.equal(outs[0], outs[1]), kw
    finally:
        fn(0)
    report(f"conv_pair {N}x{H}x{W}: 30- and 62-column strips bit-equal on three epilogue forms")


def test_conv_pair_refuses_training_and_bad_shapes():
    ops = _ops()
    x = ops.FM.empty(1, 96, 96, 64, device="cuda")
    assert ops.conv_pair_supported(x)
    assert not ops.conv_pair_supported(ops.FM.empty(1, 96, 96, 128, device="cuda"))
    assert not ops.conv_pair_supported(ops.FM.empty(1, 32, 32, 64, device="cuda"))          # below the size floor
    w1, b1, w2, b2 = _weights(3)
    pp = ops.pack_conv_pair(w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda())
    with pytest.raises(ops.L.TdvcHipError):
        ops.conv_pair(x, pp, out=x)                                                           # in place
    with pytest.raises(ops.L.TdvcHipError):
        ops.pack_conv_pair(torch.zeros(64, 32, 3, 3).cuda(), None, w2.cuda(), None)
