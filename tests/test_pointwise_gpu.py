"""HBM-bound kernels vs the CPU oracle / torch fp32 on the same seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import assert_close, fm_to_cpu, randn, rnd16, to_fm

pytestmark = pytest.mark.gpu


def _ops():
    from tdvc_amd import ops
    return ops


def test_layout_roundtrip(report):
    ops = _ops()
    x = randn(2, 3, 19, 37, seed=1)
    f = to_fm(x, ops, Cpad=8)
    assert torch.equal(fm_to_cpu(f, 3), rnd16(x))
    assert float(fm_to_cpu(f)[:, 3:].abs().max()) == 0.0
    f32 = to_fm(x, ops, Cpad=4, dtype=torch.float32)
    assert torch.equal(fm_to_cpu(f32, 3), x)


def test_upsample2x(report):
    ops = _ops()
    x = rnd16(randn(2, 64, 9, 15, seed=2))
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    assert_close(fm_to_cpu(ops.upsample2x(to_fm(x, ops))), ref, 1e-3, 1e-3, "upsample2x", report)


def test_avgpool2(report):
    ops = _ops()
    x = randn(2, 3, 32, 64, seed=3)
    ref = F.avg_pool2d(x, 2, 2)
    got = fm_to_cpu(ops.avgpool2(to_fm(x, ops, Cpad=4, dtype=torch.float32)), 3)
    assert_close(got, ref, 1e-6, 1e-6, "avgpool2", report)


@pytest.mark.parametrize("with_flow", [False, True])
def test_spynet_level_input(with_flow, report):
    """flow x2 upsample (align_corners=True, *2) + border warp, vs the oracle's flow_warp"""
    ops = _ops()
    from oracle.tdvc_ref.blocks import flow_warp_border
    H, W = 34, 60
    ref_img, supp = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(4)), \
        torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(5))
    flow_lo = randn(2, 2, H // 2, W // 2, seed=6, scale=4.0) if with_flow else None
    if with_flow:
        up = F.interpolate(flow_lo, scale_factor=2, mode="bilinear", align_corners=True) * 2.0
    else:
        up = torch.zeros(2, 2, H, W)
    warped = flow_warp_border(supp, up.permute(0, 2, 3, 1))
    cat_ref = torch.cat([ref_img, warped, up], 1)
    f_up = ops.FM.empty(2, H, W, 2, dtype=torch.float32)
    cat8 = ops.FM.empty(2, H, W, 8)
    ops.spynet_level_input(to_fm(ref_img, ops, 4, torch.float32), to_fm(supp, ops, 4, torch.float32),
                           to_fm(flow_lo, ops, 2, torch.float32) if with_flow else None, f_up, cat8)
    assert_close(fm_to_cpu(f_up), up, 1e-5, 1e-5, "flow_up", report)
    assert_close(fm_to_cpu(cat8), cat_ref, 1e-3, 2e-3, "spynet 8-ch input (warp)", report)


def test_resize_bilinear(report):
    ops = _ops()
    x = randn(1, 3, 30, 50, seed=7)
    ref = F.interpolate(x, size=(32, 64), mode="bilinear", align_corners=False)
    got = fm_to_cpu(ops.resize_bilinear(to_fm(x, ops, 4, torch.float32), 32, 64), 3)
    assert_close(got, ref, 1e-5, 1e-5, "resize up", report)
    fl = randn(1, 2, 32, 64, seed=8)
    sc = torch.tensor([50 / 64.0, 30 / 32.0])
    ref = F.interpolate(fl, size=(30, 50), mode="bilinear", align_corners=False) * sc.view(1, 2, 1, 1)
    got = fm_to_cpu(ops.resize_bilinear(to_fm(fl, ops, 2, torch.float32), 30, 50, sc.cuda()))
    assert_close(got, ref, 1e-5, 1e-5, "resize down + flow rescale", report)


def test_se_layer(report):
    ops = _ops()
    from oracle.tdvc_ref.blocks import SELayer as RefSE
    from tdvc_amd.model.modules import SELayer
    from tdvc_amd.synth import fill_parameters
    for C in (64, 128):
        ref = RefSE(C).eval()
        fill_parameters(ref)
        m = SELayer(C)
        m.load_state_dict(ref.state_dict())
        m.cuda()
        x = rnd16(randn(2, C, 33, 47, seed=9))
        r = rnd16(randn(2, C, 33, 47, seed=10))
        with torch.no_grad():
            want = F.leaky_relu(ref(x), 0.1) + r
            gate_ref = ref.gate(x).view(2, C)
        xf = to_fm(x, ops)
        assert_close(m.gate(xf).cpu(), gate_ref, 1e-5, 1e-5, f"SE gate C={C}", report)
        got = fm_to_cpu(m.run(xf, act=ops.ACT_LRELU, slope=0.1, res=to_fm(r, ops)))
        assert_close(got, want, 1e-3, 1e-3, f"SE scale C={C}", report)


def test_add_flow_and_bcast(report):
    ops = _ops()
    off = rnd16(randn(1, 64, 16, 24, seed=11))
    flow = randn(1, 2, 16, 24, seed=12)
    f = to_fm(off, ops)
    ops.add_flow(f, to_fm(flow, ops, 2, torch.float32))
    assert_close(fm_to_cpu(f), off + flow.repeat(1, 32, 1, 1), 1e-3, 1e-3, "offset + flow.repeat", report)
    x = rnd16(randn(2, 256, 8, 16, seed=13))
    b = rnd16(randn(2, 64, 8, 16, seed=14))
    xf = to_fm(x, ops)
    ops.bcast_add_act(xf, to_fm(b, ops), 4, 0.1)
    assert_close(fm_to_cpu(xf), F.leaky_relu(x + b.repeat(1, 4, 1, 1), 0.1), 1e-3, 1e-3, "temporal broadcast add", report)


def test_scale_act_res_sub(report):
    ops = _ops()
    a, b = rnd16(randn(1, 64, 8, 16, seed=15)), rnd16(randn(1, 64, 8, 16, seed=16))
    y = ops.scale_act_res(to_fm(a, ops), ops.FM.empty(1, 8, 16, 64), res=to_fm(b, ops), res_sign=-1.0)
    assert_close(fm_to_cpu(y), a - b, 1e-3, 1e-3, "a - b", report)


# the last three: W (and H) not a multiple of `scale` (floor pooling drops the remainder; the gather blocks of
# 3 * scale pixels overhang the frame) -- the geometry of the 1080p frame, scale 136 (pnet.py:219-225)
@pytest.mark.parametrize("H,W,scale", [(64, 96, 8), (32, 64, 4), (128, 192, 16), (64, 64, 8), (96, 160, 12), (88, 120, 11), (100, 150, 12)])
def test_feature_matching(H, W, scale, report):
    """pool -> 3x3 patch cosine argmax -> block gather -> cosine weight, vs the oracle's
    unfold/bmm/max/gather/fold formulation (pnet.py:219-257)."""
    ops = _ops()
    from oracle.tdvc_ref.blocks import FeatureFix as RefFF
    ff = RefFF()
    fin, fref = rnd16(randn(2, 64, H, W, seed=17)), rnd16(randn(2, 64, H, W, seed=18))
    # make some input patches resemble reference patches so the argmax is meaningful
    h2 = H // 2
    fin[:, :, :h2] = rnd16(fref[:, :, h2: 2 * h2] + 0.05 * fin[:, :, :h2])
    with torch.no_grad():
        ind, out = ff.match(fin, fref, scale)
        cor = F.cosine_similarity(fin, out).unsqueeze(1)
        want = torch.cat([fin, out], 1) * cor
    f_in, f_ref = to_fm(fin, ops), to_fm(fref, ops)
    pin, pref = ops.avgpool_k(f_in, scale), ops.avgpool_k(f_ref, scale)
    assert_close(pin.permute(0, 3, 1, 2).cpu(), F.avg_pool2d(fin, scale, scale), 1e-5, 1e-5, "avgpool_k", report)
    idx = ops.patch_match(pin, pref)
    assert torch.equal(idx.cpu().long(), ind), f"patch argmax differs: {idx.cpu().tolist()} vs {ind.tolist()}"
    cat = ops.FM.empty(2, H, W, 128)
    ops.match_gather(f_in, f_ref, idx, scale, cat)
    assert_close(fm_to_cpu(cat), want, 2e-3, 2e-3, f"match_gather {H}x{W} scale {scale}", report)


def test_entropy_rate_terms(report):
    """factorised + Gaussian-conditional likelihoods (bits) and quantisation vs the oracle"""
    ops = _ops()
    from oracle.tdvc_ref.coder import EntropyBottleneck as RefEB, GaussianConditional as RefGC
    from tdvc_amd.model.coder import EntropyBottleneck
    from tdvc_amd.synth import fill_parameters
    C = 128
    ref = RefEB(C).eval()
    h = torch.nn.Module(); h.add_module("entropy_bottleneck", ref); fill_parameters(h)
    with torch.no_grad():
        for i in range(4):
            getattr(ref, f"_factor{i}").copy_(randn(C, 3, 1, seed=30 + i, scale=0.3))
        ref.quantiles[:, 0, 1] = randn(C, seed=40, scale=0.4)
    eb = EntropyBottleneck(C)
    eb.load_state_dict(ref.state_dict())
    eb.cuda()
    z = randn(2, C, 5, 9, seed=19, scale=3.0)
    with torch.no_grad():
        z_hat_ref, lik = ref(z)
    bits_ref = float(-torch.log2(lik).double().sum())
    zf = to_fm(z, ops, C, torch.float32)
    z_hat = ops.FM.empty(2, 5, 9, C)
    bits = torch.zeros(1, dtype=torch.float64, device="cuda")
    ops.eb_forward(zf, eb.packed_params(), z_hat, bits)
    report(f"EB bits got {float(bits):.4f} ref {bits_ref:.4f}")
    assert_close(fm_to_cpu(z_hat), z_hat_ref, 1e-3, 1e-3, "z_hat", report)
    assert abs(float(bits) - bits_ref) <= 1e-4 * bits_ref + 1e-2
    # Gaussian conditional
    gc = RefGC()
    y = randn(2, C, 9, 15, seed=20, scale=4.0)
    scales = torch.rand(2, C, 9, 15, generator=torch.Generator().manual_seed(21)) * 3.0 - 0.2
    means = randn(2, C, 9, 15, seed=22)
    with torch.no_grad():
        _, lik = gc(y, scales, means, False)
    bits_ref = float(-torch.log2(lik).double().sum())
    gp = to_fm(torch.cat([scales, means], 1), ops, 2 * C, torch.float32)
    bits = torch.zeros(1, dtype=torch.float64, device="cuda")
    ops.gc_forward(to_fm(y, ops, C, torch.float32), gp, bits)
    report(f"GC bits got {float(bits):.4f} ref {bits_ref:.4f}")
    assert abs(float(bits) - bits_ref) <= 2e-4 * bits_ref + 1e-2
    yh = ops.quantize(to_fm(y, ops, C, torch.float32), ops.FM.empty(2, 9, 15, C))
    assert torch.equal(fm_to_cpu(yh), torch.round(y))


@pytest.mark.parametrize("src,dst", [((64, 96), (57, 90)), ((30, 50), (32, 64)), ((96, 64), (90, 61))])
def test_resize_bilinear_backward(src, dst, report):
    """adjoint of resize_bilinear (flownet.py:153-173: the flow's way back from the x32-padded size, with the per-channel
    W/Wu, H/Hu rescale) against torch autograd of F.interpolate(bilinear, align_corners=False), down- and up-scaling"""
    ops = _ops()
    (h, w), (H, W) = src, dst
    fl = randn(2, 2, h, w, seed=91)
    g = randn(2, 2, H, W, seed=92)
    sc = torch.tensor([W / w, H / h], dtype=torch.float32)
    x = fl.clone().requires_grad_(True)
    (F.interpolate(x, size=(H, W), mode="bilinear", align_corners=False) * sc.view(1, 2, 1, 1) * g).sum().backward()
    dx = ops.FM.zeros(2, h, w, 2, dtype=torch.float32)
    ops.resize_bilinear_backward(to_fm(g, ops, 2, torch.float32), dx, sc.cuda())
    assert_close(fm_to_cpu(dx), x.grad, 1e-5, 1e-5, f"resize_bilinear backward {src}->{dst}", report)
    # accumulation: a second call adds
    ops.resize_bilinear_backward(to_fm(g, ops, 2, torch.float32), dx, sc.cuda())
    assert_close(fm_to_cpu(dx), 2 * x.grad, 2e-5, 2e-5, "resize_bilinear backward accumulates", report)


def test_spynet_backward_tail_reproducible_under_concurrent_stream(report):
    """zero mirror -> accumulate -> spynet_level_input_backward, repeated while a second stream runs 7x7 weight-gradient
    launches: every repetition must equal the unloaded result bit for bit (the reproducer of the packed-FP32 finding of round
    3: with v_pk_add_f32 in the kernel 28-30 of 30 loaded runs differed, always in lanes 48..63 of a wave)"""
    from tdvc_amd import ops
    N, H, W = 4, 256, 256
    g = torch.Generator(device="cuda").manual_seed(5)
    supp = ops.FM(torch.rand(N, H, W, 4, device="cuda", generator=g))
    flow_up = ops.FM(torch.randn(N, H, W, 2, device="cuda", generator=g) * 2.0)
    dcat8 = ops.FM((torch.randn(N, H, W, 8, device="cuda", generator=g) * 1e-3).half())
    dflow = ops.FM(torch.randn(N, H, W, 2, device="cuda", generator=g) * 1e-3)
    x = ops.FM(torch.randn(N, H, W, 32, device="cuda", generator=g).half())
    gy = ops.FM(torch.randn(N, H, W, 64, device="cuda", generator=g).half())
    pc = ops.pack_conv(torch.randn(64, 32, 7, 7) * 0.02, torch.zeros(64), stride=1, pad=3)
    dw, db = torch.zeros(64 * 32 * 49, device="cuda"), torch.zeros(64, device="cuda")
    side = torch.cuda.Stream()

    def tail():
        dup = ops.FM(torch.zeros_like(flow_up.t))
        dlo = ops.FM(torch.zeros(N, H // 2, W // 2, 2, device="cuda"))
        ops.scale_act_res(dup, dup, res=dflow, res_sign=1.0)
        ops.spynet_level_input_backward(supp, flow_up, dcat8, dup, dlo)
        return dup.t, dlo.t

    ref = [t.clone() for t in tail()]
    torch.cuda.synchronize()
    bad = 0
    for _ in range(12):
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            for _ in range(3):
                ops.conv_wgrad(pc, gy, x, dw, scale=1.0, db=db)
        out = tail()
        torch.cuda.synchronize()
        bad += not (torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
    report(f"SPyNet level backward tail under a concurrent weight-gradient stream: {bad}/12 runs differ from the unloaded result")
    assert bad == 0
