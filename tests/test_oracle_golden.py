"""CPU: the oracle's blocks reproduce the golden vectors captured from the REFERENCE's own
modules (tests/golden/make_golden.py) — this is what pins the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle.tdvc_ref import blocks as ob
from tdvc_amd.synth import fill_parameters, make_gop, ref_list

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_blocks.npz"))
GOLD_CH = [0, 7, 21, 42, 63]
SIZES = {"a": (32, 64), "b": (64, 96)}


def filled(mod, prefix):
    holder = nn.Module()
    cur = holder
    parts = prefix.split(".")
    for p in parts[:-1]:
        nxt = nn.Module()
        cur.add_module(p, nxt)
        cur = nxt
    cur.add_module(parts[-1], mod)
    fill_parameters(holder)
    return mod.eval()


def check(name, out, tol=2e-6, loose=False):
    """loose: stages downstream of the DCN's unconditional fp16 rounding (dcn_v2_amp.py:67-68): a
    1e-7 difference in its input (fp32 summation order between two runs) can move single outputs
    by one fp16 ulp, so those stages are held to a few fp16 ulps max and 1e-5 on average."""
    want = torch.from_numpy(G[name])
    o = out.float()
    if loose:
        tol = 1e-3
        sel = o[:, GOLD_CH] if name + "_chsum" in G.files else o
        assert float((sel - want).abs().mean()) < 1e-5, name
    if name + "_chsum" in G.files:
        assert torch.allclose(o[:, GOLD_CH], want, atol=tol, rtol=1e-5), name
        np.testing.assert_allclose(o.double().sum(dim=(0, 2, 3)).numpy(), G[name + "_chsum"], rtol=1e-6, atol=0.05 if loose else 1e-3)
        np.testing.assert_allclose(o.double().abs().sum(dim=(0, 2, 3)).numpy(), G[name + "_chabs"], rtol=1e-6, atol=0.05 if loose else 1e-3)
    else:
        assert o.shape == want.shape
        assert torch.allclose(o, want, atol=tol, rtol=1e-5), name


@pytest.mark.parametrize("tag", ["a", "b"])
def test_blocks_match_reference_goldens(tag):
    H, W = SIZES[tag]
    g = make_gop(1234, 4, H, W)
    cur, prev = g[1:2], g[0:1]
    refs = ref_list([g[0:1], g[1:2], g[2:3]])
    with torch.no_grad():
        fe = filled(ob.FeaExtra(2), "extra_fea")
        f_cur, f_ref = fe(cur), fe(prev)
        check(f"{tag}_feaextra", f_cur)
        check(f"{tag}_spynet", filled(ob.SPyNet(), "motion_est.spynet")(cur, prev))
        estmv = filled(ob.OffsetGen(), "motion_est")(f_cur, f_ref, cur, prev)
        check(f"{tag}_offsetgen", estmv)
        check(f"{tag}_se", filled(ob.SELayer(64), "motion_est.attn")(f_cur))
        pred1 = filled(ob.MCNet(3), "mcnet")(estmv * 0.5, f_ref)
        check(f"{tag}_mcnet", pred1, loose=True)
        check(f"{tag}_mcfilter", filled(ob.LoopFilter(), "mcfilter")(pred1, refs), loose=True)
        check(f"{tag}_loopfilter", filled(ob.FeatureFix(), "loopfilter")(pred1, refs), loose=True)
        check(f"{tag}_resblock", filled(ob.Res_Block(64), "extra_fea.residual_layer.0")(f_cur))
        fl = torch.from_numpy(np.random.default_rng(7).standard_normal((1, H, W, 2))).float() * 6.0
        check(f"{tag}_warp", ob.flow_warp_border(cur, fl))
        x = g[0:1, :, : H - 5, : W - 3]
        p = ob.pad_to(x, 64)
        assert list(p.shape) == G[f"{tag}_pad_shape"].tolist()
        assert abs(float(p.double().sum()) - G[f"{tag}_pad_sum"][0]) < 1e-6
        assert torch.equal(ob.crop_to(p, x.shape[-2:]), x)
        if tag == "b":
            ff = filled(ob.FeatureFix(), "loopfilter")
            ff.train()
            check("b_loopfilter_train", ff(pred1, refs), loose=True)


def test_optimizer_partition_matches_reference():
    from oracle.tdvc_ref import VideoCompressor
    main, aux = ob.split_optim_params(VideoCompressor())
    assert [len(main), len(aux)] == G["optim_counts"].tolist()
    assert all(n.endswith(".quantiles") for n in aux)


def test_dcn_zero_offset_known_answer():
    """the reference's own KAT (main/utils/dcnv2/testcpu.py:34-69) applied to the oracle DCN"""
    N, C, H, W, k = 2, 2, 4, 4, 3
    w = torch.zeros(C, C, k, k)
    for c in range(C):
        w[c, c, 1, 1] = 1.0
    x = torch.randn(N, C, H, W, generator=torch.Generator().manual_seed(0))
    out = ob.dcn_v2_forward_ref(x, w, torch.zeros(C), torch.zeros(N, 2 * k * k, H, W),
                                torch.sigmoid(torch.zeros(N, k * k, H, W)), k, k, 1, 1, 1, 1, 1, 1, 1)
    assert float((x - 2 * out).abs().max()) < 1e-10


def test_dcn_integer_offsets_equal_shifted_conv():
    """integer offsets turn the deformable conv into a plain conv on a shifted image"""
    import torch.nn.functional as F
    x = torch.randn(1, 8, 9, 11, generator=torch.Generator().manual_seed(1))
    w = torch.randn(4, 8, 3, 3, generator=torch.Generator().manual_seed(2))
    b = torch.randn(4, generator=torch.Generator().manual_seed(3))
    off = torch.zeros(1, 2 * 2 * 9, 9, 11)
    off[:, 0::2] = 1.0      # dh = +1
    off[:, 1::2] = -2.0     # dw = -2
    out = ob.dcn_v2_forward_ref(x, w, b, off, torch.ones(1, 2 * 9, 9, 11), 3, 3, 1, 1, 1, 1, 1, 1, 2)
    xs = torch.zeros_like(x)
    xs[:, :, :-1, 2:] = x[:, :, 1:, :-2]
    # equal away from the border (the plain conv zero-pads the SHIFTED image, the DCN the original)
    ref = F.conv2d(xs, w, b, padding=1)
    assert torch.allclose(out[:, :, 2:-2, 3:-3], ref[:, :, 2:-2, 3:-3], atol=1e-5)
