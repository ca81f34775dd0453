"""CPU: host-side logic of the product (no GPU compute)."""
import numpy as np
import torch

from tdvc_amd import ops
from tdvc_amd.synth import fill_parameters, hash_uniform, make_gop, ref_list, split_optim_params


def test_filler_is_a_pure_function_of_name_and_index():
    u = hash_uniform(5, 12345)
    np.testing.assert_allclose(u, hash_uniform(5, 12345))
    assert 0.0 <= u.min() and u.max() < 1.0 and len(set(np.round(u, 9))) == 5
    a, b = torch.nn.Conv2d(8, 8, 3), torch.nn.Conv2d(8, 8, 3)
    ha, hb = torch.nn.Module(), torch.nn.Module()
    ha.add_module("x", a); hb.add_module("x", b)
    fill_parameters(ha); fill_parameters(hb)
    assert torch.equal(a.weight, b.weight) and torch.equal(a.bias, b.bias)
    assert abs(float(a.weight.detach().mean())) < 0.02 and 0.05 < float(a.weight.detach().std()) < 0.2


def test_make_gop_and_ref_list():
    g = make_gop(1234, 7, 64, 96)
    assert g.shape == (7, 3, 64, 96) and g.dtype == torch.float32
    assert torch.equal(g, make_gop(1234, 7, 64, 96)) and not torch.equal(g, make_gop(1235, 7, 64, 96))
    q = g * 255.0
    assert float((q - q.round()).abs().max()) < 1e-4 and 0.0 <= float(g.min()) and float(g.max()) <= 1.0
    # frame t is frame 0 shifted by (t, 2t) px up to noise
    d = (g[1, :, :-1, :-2] - g[0, :, 1:, 2:]).abs().mean()
    assert float(d) < 0.03
    I, r1, r2, r3 = [torch.full((1, 3, 2, 2), float(v)) for v in range(4)]
    f = lambda refs: ref_list(refs)[0, :, 0, 0, 0].tolist()
    assert f([I]) == [0, 0, 0, 0]
    assert f([I, r1]) == [0, 0, 1, 1]
    assert f([I, r1, r2]) == [0, 0, 1, 2]
    assert f([I, r1, r2, r3]) == [0, 1, 2, 3]


def test_state_dict_is_reference_compatible():
    from oracle.tdvc_ref import VideoCompressor as Ref
    from tdvc_amd.model import VideoCompressor
    m, r = VideoCompressor(), Ref()
    a, b = m.state_dict(), r.state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(a[k].shape == b[k].shape for k in a)
    m.load_state_dict(b, strict=True)
    main, aux = split_optim_params(m)
    assert len(aux) == 2 and len(main) + len(aux) == len(list(m.parameters()))
    for k in ("motion_est.spynet.basic_module.5.basic_module.4.conv.weight", "mcnet.dconv.conv_offset_mask.bias",
              "mvCoder.entropy_bottleneck._matrix0", "resCoder.g_a.0.gdn.beta_reparam.lower_bound.bound",
              "mcfilter.layer1.temporal_conv3d.weight", "loopfilter.conv_13.weight", "motion_est.offset_conv12.l1.weight"):
        assert k in a


def test_fm_views_address_arithmetic():
    t = torch.zeros(2, 4, 6, 256, dtype=torch.float16)
    f = ops.FM(t)
    base = t.data_ptr()
    d = f.ch(64, 64).desc()
    assert d.p == base + 64 * 2 and d.C == 64 and d.sp == 256 and d.sn == 4 * 6 * 256
    d = f.batch(1, 1).ch(128, 128).desc()
    assert d.p == base + (4 * 6 * 256 + 128) * 2 and d.N == 1
    s = f.as_slices(1, 4, 64)
    d = s.desc()
    assert d.p == base + 4 * 6 * 256 * 2 and d.N == 4 and d.sn == 64 and d.C == 64 and d.sp == 256
    d = s.batch(1, 3).desc()
    assert d.p == base + (4 * 6 * 256 + 64) * 2 and d.N == 3
    # size-1 dimensions carry arbitrary torch strides (e.g. after permute): FM must not trust them
    odd = torch.zeros(1, 128, 1, 1, dtype=torch.float16).permute(0, 2, 3, 1)
    d = ops.FM(odd).desc()
    assert d.sp == 128 and d.sn == 128 and d.C == 128
    nar = ops.FM(torch.zeros(1, 1, 8, 64, dtype=torch.float16)[:, :, :3])
    assert nar.W == 3 and nar.desc().sp == 64
    f32 = ops.FM(torch.zeros(1, 2, 2, 4))
    assert f32.f32 and f32.desc().dtype == 1


def test_pack_conv_host_options():
    w = torch.randn(16, 6, 3, 3)
    pc = ops.pack_conv(w, torch.randn(16), stride=1, pad=1, device="cpu")
    assert pc.cin == 8 and pc.cin_real == 6 and pc.ck == 8 and pc.bias.numel() == 32 and len(pc.taps) == 9
    pc = ops.pack_conv(torch.randn(512, 128, 3, 3), None, stride=1, pad=1, shuffle=True, device="cpu")
    assert pc.shuffle and pc.cout == 512 and pc.bias.numel() == 512 and pc.ck == 32
    from tdvc_amd.model.coder import MaskedConv2d
    mc = MaskedConv2d(8, 16, kernel_size=5, padding=2)
    taps = mc.live_taps()
    assert len(taps) == 12 and all(mc.mask[0, 0, dy, dx] == 1 for dy, dx in taps) and int(mc.mask[0, 0].sum()) == 12


def test_pad_crop_match_oracle():
    from oracle.tdvc_ref.blocks import crop_to, pad_to
    from tdvc_amd.codec_utils import crop, pad
    x = torch.randn(2, 3, 1080, 1920)
    p = pad(x, 64)
    assert p.shape[-2:] == (1088, 1920) and torch.equal(p, pad_to(x, 64))
    assert torch.equal(p[..., :4, :], torch.zeros(2, 3, 4, 1920)) and torch.equal(p[..., -4:, :], torch.zeros(2, 3, 4, 1920))
    assert torch.equal(crop(p, (1080, 1920)), x) and torch.equal(crop(p, (1080, 1920)), crop_to(p, (1080, 1920)))
    y = torch.randn(1, 3, 59, 67)
    assert torch.equal(pad(y, 64), pad_to(y, 64)) and torch.equal(crop(pad(y, 64), (59, 67)), y)
