#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE's own Python blocks.

Runs ONLY in the build container (needs /root/reference, read-only); the committed
`.npz` files are what travels.  Nothing of the reference's source is stored: fixtures
hold inputs' seeds and the reference's numeric outputs.

How the reference is imported (no file of it is modified or copied):
  * `mmcv.cnn.ConvModule` / `mmcv.runner.BaseModule`, `compressai.*` names and `_ext` are
    absent from this image -> tiny in-memory `sys.modules` stand-ins (ConvModule =
    Conv2d(bias=True) + optional ReLU/Sigmoid with attributes `conv` / `activate`, which is
    what mmcv builds for `norm_cfg=None`); the compressai stand-ins are never executed.
  * `main.model.flownet.load_state_dict_from_url` is patched to return a local dict
    (no network); weights are then overwritten by the closed-form filler.
  * `_ext.dcn_v2_forward` is routed to the oracle's DCN so `MCNet` can run; the golden for
    MCNet therefore pins everything in it EXCEPT the DCN arithmetic (pinned separately by
    the reference's `check_zero_offset` known-answer test).

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402
from oracle.tdvc_ref import blocks as ob  # noqa: E402


GOLD_CH = [0, 7, 21, 42, 63]


def install_shims():
    class ConvModule(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                     conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), **kw):
            super().__init__()
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=True)
            t = None if act_cfg is None else act_cfg["type"]
            self.activate = {"ReLU": nn.ReLU, "Sigmoid": nn.Sigmoid}[t]() if t else None

        def forward(self, x):
            x = self.conv(x)
            return x if self.activate is None else self.activate(x)

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("mmcv")
    mod("mmcv.cnn", ConvModule=ConvModule)
    mod("mmcv.runner", BaseModule=BaseModule)
    ph = type("Placeholder", (nn.Module,), {})
    mod("compressai")
    mod("compressai.layers", ResidualBlock=ph, ResidualBlockUpsample=ph, ResidualBlockWithStride=ph,
        conv3x3=ph, subpel_conv3x3=ph)
    mod("compressai.models")
    mod("compressai.models.waseda", Cheng2020Anchor=ph)
    mod("_ext", dcn_v2_forward=lambda *a: ob.dcn_v2_forward_ref(*a))


def main():
    install_shims()
    import main.model.flownet as rflow
    rflow.load_state_dict_from_url = lambda url: {
        k: v for k, v in ob.SPyNet().state_dict().items() if k not in ("mean", "std")}
    import main.model.pnet as rp
    import main.model.inflate as rinf
    import main.utils.utils as ru

    out = {}
    full = {}
    torch.manual_seed(0)

    def store(name, t):
        """keep fixtures small: tensors with > 8 channels are stored as 5 full channels
        (GOLD_CH) plus per-channel float64 sum / abs-sum over ALL channels."""
        full[name] = t
        a = t.numpy()
        if a.ndim == 4 and a.shape[1] > 8:
            out[name] = a[:, GOLD_CH]
            out[name + "_chsum"] = a.astype(np.float64).sum(axis=(0, 2, 3))
            out[name + "_chabs"] = np.abs(a.astype(np.float64)).sum(axis=(0, 2, 3))
        else:
            out[name] = a

    def run_pair(name, ref_mod, ora_mod, prefix, *inputs):
        """fill oracle module under its full-model key prefix, copy into the reference
        module strictly, run both, store the reference output."""
        holder = nn.Module()
        # give parameters their full-model names so the filler salts match everywhere
        parts = prefix.split(".")
        cur = holder
        for p in parts[:-1]:
            nxt = nn.Module()
            cur.add_module(p, nxt)
            cur = nxt
        cur.add_module(parts[-1], ora_mod)
        fill_parameters(holder)
        ref_mod.load_state_dict(ora_mod.state_dict(), strict=True)
        ref_mod.eval()
        ora_mod.eval()
        with torch.no_grad():
            r = ref_mod(*inputs)
            o = ora_mod(*inputs)
        err = float((r.float() - o.float()).abs().max())
        print(f"{name:28s} ref-vs-oracle max abs err {err:.3e}  out absmax {float(r.abs().max()):.3f}")
        store(name, r.float())
        return r

    for tag, (H, W) in {"a": (32, 64), "b": (64, 96)}.items():
        g = make_gop(1234, 4, H, W)
        cur, prev = g[1:2], g[0:1]
        refs = ref_list([g[0:1], g[1:2], g[2:3]])         # (1,4,3,H,W)
        # a2 FeaExtra
        f_cur = run_pair(f"{tag}_feaextra", rp.FeaExtra(2), ob.FeaExtra(2), "extra_fea", cur)
        with torch.no_grad():
            fe = ob.FeaExtra(2)
            h = nn.Module(); h.add_module("extra_fea", fe); fill_parameters(h)
            f_ref = fe.eval()(prev)
        # a4/a5 SPyNet (includes flow_warp)
        run_pair(f"{tag}_spynet", rflow.SPyNet(pretrained="local"), ob.SPyNet(), "motion_est.spynet", cur, prev)
        # a3 OffsetGen (incl. SPyNet + SE)
        estmv = run_pair(f"{tag}_offsetgen", rp.OffsetGen(), ob.OffsetGen(), "motion_est", f_cur, f_ref, cur, prev)
        # a6 SELayer alone
        run_pair(f"{tag}_se", rinf.SELayer(64), ob.SELayer(64), "motion_est.attn", f_cur)
        # a13 MCNet (DCN arithmetic from the oracle, everything else reference)
        pred1 = run_pair(f"{tag}_mcnet", rp.MCNet(3), ob.MCNet(3), "mcnet", estmv * 0.5, f_ref)
        # a14 LoopFilter (multi-frame fusion)
        run_pair(f"{tag}_mcfilter", rp.LoopFilter(), ob.LoopFilter(), "mcfilter", pred1, refs)
        # a15 FeatureFix (in-loop filter), eval-mode scale = H/8
        run_pair(f"{tag}_loopfilter", rp.FeatureFix(), ob.FeatureFix(), "loopfilter", pred1, refs)
        # Res_Block, pad / crop (a16)
        run_pair(f"{tag}_resblock", ru.Res_Block(64), ob.Res_Block(64), "extra_fea.residual_layer.0", f_cur)
        x = g[0:1, :, : H - 5, : W - 3]
        p_ref = ru.pad(x, 64)
        assert torch.equal(p_ref, ob.pad_to(x, 64))
        assert torch.equal(ru.crop(p_ref, x.shape[-2:]), ob.crop_to(p_ref, x.shape[-2:]))
        out[f"{tag}_pad_shape"] = np.array(p_ref.shape)
        out[f"{tag}_pad_sum"] = np.array([float(p_ref.double().sum()), float(p_ref[..., 0, :].abs().sum())])
        # flow_warp alone with a large flow that leaves the image (border clamp)
        fl = (torch.from_numpy(np.random.default_rng(7).standard_normal((1, H, W, 2))).float() * 6.0)
        w_ref = rflow.flow_warp(cur, fl, padding_mode="border")
        assert float((w_ref - ob.flow_warp_border(cur, fl)).abs().max()) < 1e-6
        out[f"{tag}_warp"] = w_ref.numpy()

    # FeatureFix in TRAINING mode (scale = 8) on the larger size
    g = make_gop(1234, 4, 64, 96)
    refs = ref_list([g[0:1], g[1:2], g[2:3]])
    ff_r, ff_o = rp.FeatureFix(), ob.FeatureFix()
    h = nn.Module(); h.add_module("loopfilter", ff_o); fill_parameters(h)
    ff_r.load_state_dict(ff_o.state_dict(), strict=True)
    ff_r.train(); ff_o.train()
    x = full["b_mcnet"]
    with torch.no_grad():
        r, o = ff_r(x, refs), ff_o(x, refs)
    print("b_loopfilter_train err", float((r - o).abs().max()))
    store("b_loopfilter_train", r)

    # a17 configure_optimizers partition on the oracle's full model
    from oracle.tdvc_ref import VideoCompressor
    m = VideoCompressor()
    opt, aux = ru.configure_optimizers({"lr": 1e-4}, m)
    main_names, aux_names = ob.split_optim_params(m)
    assert len(opt.param_groups[0]["params"]) == len(main_names)
    assert len(aux.param_groups[0]["params"]) == len(aux_names) == 2
    out["optim_counts"] = np.array([len(main_names), len(aux_names)])

    dst = os.path.join(REPO, "tests", "golden", "ref_blocks.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst) / 1e6, "MB")


if __name__ == "__main__":
    main()
