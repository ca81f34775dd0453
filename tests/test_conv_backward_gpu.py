"""Conv backward (data / weight / bias gradients) vs torch autograd in fp32 on the same fp16-rounded operands."""
import pytest
import torch
import torch.nn.functional as F

from util import assert_close, fm_to_cpu, randn, rnd16, to_fm

pytestmark = pytest.mark.gpu


def _ops():
    from tdvc_amd import ops
    return ops


CASES = [
    # name, N, cin, cout, k, stride, pad, H, W, shuffle
    ("3x3_64_64", 2, 64, 64, 3, 1, 1, 40, 72, False),
    ("3x3_64_64_large", 1, 64, 64, 3, 1, 1, 96, 128, False),       # dgrad on the LDS-DMA kernel
    ("3x3_3_64", 2, 3, 64, 3, 1, 1, 33, 47, False),
    ("3x3_128_192", 1, 128, 192, 3, 1, 1, 17, 30, False),
    ("1x1_128_64", 2, 128, 64, 1, 1, 0, 20, 36, False),
    ("7x7_8_32", 1, 8, 32, 7, 1, 3, 34, 60, False),
    ("7x7_32_16", 1, 32, 16, 7, 1, 3, 19, 25, False),
    ("3x3_s2_64_128", 2, 64, 128, 3, 2, 1, 32, 48, False),         # space-to-depth forward, sub-pixel dgrad
    ("1x1_s2_64_128", 1, 64, 128, 1, 2, 0, 32, 64, False),
    ("subpel_128_64", 1, 128, 256, 3, 1, 1, 12, 20, True),          # conv + PixelShuffle(2)
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_backward(case, report):
    ops = _ops()
    name, N, cin, cout, k, stride, pad, H, W, shuffle = case
    x = rnd16(randn(N, cin, H, W, seed=101))
    w = rnd16(randn(cout, cin, k, k, seed=102) * (1.0 / (cin * k * k) ** 0.5))
    b = randn(cout, seed=103) * 0.1
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y = F.conv2d(xr, wr, br, stride=stride, padding=pad)
    if shuffle:
        y = F.pixel_shuffle(y, 2)
    gy = rnd16(randn(*y.shape, seed=104) * 0.5)
    gx, gw, gb = torch.autograd.grad(y, (xr, wr, br), gy)

    wd, bd = w.cuda(), b.cuda()
    pc = ops.pack_conv(wd, bd, stride=stride, pad=pad, shuffle=shuffle)
    xf = to_fm(x, ops)
    yf = ops.conv(xf, pc)
    assert_close(fm_to_cpu(yf, y.shape[1]), y.detach(), 2e-3, 2e-3, f"fwd {name}", report)
    g = to_fm(gy, ops)
    gq = ops.pixel_unshuffle(g) if shuffle else g
    # data gradient, accumulated onto a non-zero buffer
    base = rnd16(randn(N, cin, H, W, seed=105) * 0.1)
    dx = to_fm(base, ops)
    ops.conv_dgrad(pc, gq, dx, accumulate=True)
    scale = float(gx.abs().max())
    assert_close(fm_to_cpu(dx, cin), gx + base, 3e-3, 3e-3 * max(1.0, scale), f"dgrad {name}", report)
    # weight / bias gradients accumulate into fp32 buffers
    dw = torch.full_like(wd, 0.25)
    db = torch.full_like(bd, -0.5)
    ops.conv_wgrad(pc, gq, xf, dw)
    ops.conv_bgrad(pc, gq, db)
    tol_w = 2e-3 * float(gw.abs().max()) + 1e-3
    assert_close(dw.cpu() - 0.25, gw, 2e-3, tol_w, f"wgrad {name}", report)
    assert_close(db.cpu() + 0.5, gb, 2e-3, 2e-3 * float(gb.abs().max()) + 1e-3, f"bgrad {name}", report)
    # bitwise reproducible; the fused form (bias gradient from the same launch) gives the same dW and a matching db
    dw2 = torch.full_like(wd, 0.25)
    db2 = torch.full_like(bd, -0.5)
    ops.conv_wgrad(pc, gq, xf, dw2, db=db2)
    assert torch.equal(dw, dw2)
    assert_close(db2.cpu() + 0.5, gb, 2e-3, 2e-3 * float(gb.abs().max()) + 1e-3, f"fused bgrad {name}", report)
    db3 = torch.full_like(bd, -0.5)
    ops.conv_wgrad(pc, gq, xf, torch.zeros_like(wd), db=db3)
    assert torch.equal(db2, db3)


def test_repack_follows_parameter_updates(report):
    """PackedConv references the device parameter: an in-place update + repack() changes forward and dgrad"""
    ops = _ops()
    w = torch.nn.Parameter(rnd16(randn(64, 64, 3, 3, seed=111) * 0.05).cuda())
    b = torch.nn.Parameter((randn(64, seed=112) * 0.1).cuda())
    pc = ops.pack_conv(w, b, stride=1, pad=1)
    x = rnd16(randn(1, 64, 24, 40, seed=113))
    xf = to_fm(x, ops)
    y0 = fm_to_cpu(ops.conv(xf, pc))
    g = to_fm(rnd16(randn(1, 64, 24, 40, seed=114)), ops)
    dx0 = fm_to_cpu(ops.conv_dgrad(pc, g, ops.FM.zeros(1, 24, 40, 64), accumulate=False))
    with torch.no_grad():
        w.mul_(0.5)
        b.add_(1.0)
    pc.repack()
    y1 = fm_to_cpu(ops.conv(xf, pc))
    dx1 = fm_to_cpu(ops.conv_dgrad(pc, g, ops.FM.zeros(1, 24, 40, 64), accumulate=False))
    ref = F.conv2d(x, w.detach().cpu(), b.detach().cpu(), padding=1)
    assert_close(y1, ref, 2e-3, 2e-3, "forward after repack", report)
    assert_close(dx1, 0.5 * dx0, 2e-3, 2e-3, "dgrad after repack", report)
    assert not torch.allclose(y0, y1)


def test_pack_batch_equals_per_layer_repack(report):
    """the one-launch re-pack of many layers (forward + dgrad forms, biases incl. the sub-pixel row order) writes
    exactly the bytes the per-layer repack() writes"""
    ops = _ops()
    geo = [(64, 64, 3, 1, 1, False), (128, 64, 3, 2, 1, False), (64, 128, 1, 1, 0, False), (256, 64, 3, 1, 1, True),
           (32, 64, 7, 1, 3, False), (64, 8, 3, 1, 1, False)]
    layers = []
    for i, (co, ci, k, s_, pd, shuf) in enumerate(geo):
        w = torch.nn.Parameter((randn(co, ci, k, k, seed=300 + i) * 0.05).cuda())
        b = torch.nn.Parameter((randn(co, seed=320 + i) * 0.1).cuda())
        pc = ops.pack_conv(w, b, stride=s_, pad=pd, shuffle=shuf)
        g = to_fm(rnd16(randn(1, co, 32 // s_, 64 // s_, seed=340 + i)), ops)
        layers.append((w, b, pc))
        ops.conv_dgrad(pc, g, ops.FM.zeros(1, 32, 64, max(ci, 8)), accumulate=False)      # builds the dgrad form
    batch = ops.PackBatch([pc for _, _, pc in layers])
    assert len(batch.pcs) >= len(layers)
    with torch.no_grad():
        for w, b, _ in layers:
            w.mul_(0.7).add_(0.01)
            b.sub_(0.3)
    for _, _, pc in layers:
        pc.repack()
    want = [(pc.w.clone(), None if pc.bias is None else pc.bias.clone()) for pc in batch.pcs]
    for pc in batch.pcs:
        pc.w.zero_()
        if pc.bsrc is not None:
            pc.bias.zero_()
    batch.run()
    torch.cuda.synchronize()
    for pc, (w0, b0) in zip(batch.pcs, want):
        assert torch.equal(pc.w, w0)
        if pc.bsrc is not None:
            assert torch.equal(pc.bias, b0)
    report(f"pack batch: {len(batch.pcs)} packed forms in {batch.total_blocks} blocks, bytes identical to repack()")


def test_act_backward_and_unshuffle(report):
    ops = _ops()
    y = rnd16(randn(2, 64, 9, 13, seed=121))
    r = rnd16(randn(2, 64, 9, 13, seed=122))
    g = rnd16(randn(2, 64, 9, 13, seed=123))
    out = ops.act_backward(to_fm(g, ops), to_fm(y, ops), ops.ACT_LRELU, 0.1, res=to_fm(r, ops), out=ops.FM.empty(2, 9, 13, 64))
    ref = torch.where(y - r > 0, g, rnd16(g * 0.1))
    assert_close(fm_to_cpu(out), ref, 1e-3, 1e-3, "act_backward lrelu + residual", report)
    out = ops.act_backward(to_fm(g, ops), to_fm(y, ops), ops.ACT_RELU)
    assert_close(fm_to_cpu(out), torch.where(y > 0, g, torch.zeros_like(g)), 0, 0, "act_backward relu", report)
    z = rnd16(randn(1, 16, 8, 10, seed=124))
    u = fm_to_cpu(ops.pixel_unshuffle(to_fm(z, ops)))
    assert torch.equal(u, z.view(1, 16, 4, 2, 5, 2).permute(0, 3, 5, 1, 2, 4).reshape(1, 64, 4, 5))
