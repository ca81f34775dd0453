"""MI355X-native blocks of TDVC's P-frame path.

Each class mirrors a reference module (same attribute names => same state-dict keys, so a
reference checkpoint loads with strict=True) but holds parameters only: the `nn.Conv2d`
objects are containers, their torch forward is never called.  Compute goes through
`tdvc_amd.ops` -> libtdvc_hip.so (hand-written gfx950 kernels).  Weights are packed once into
MFMA fragment order (fp16) and cached; `VideoCompressor.clear_packed()` drops the cache.

Activations are channel-innermost fp16 feature maps (`ops.FM`); concatenations are channel
slices of a shared buffer, never copies.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_LRELU, ACT_NONE, ACT_RELU, FM


class PackCache:
    """mixin: lazily packed weights keyed by name"""

    def _pk(self, key, fn):
        c = self.__dict__.setdefault("_packed", {})
        if key not in c:
            c[key] = fn()
        return c[key]


def _dev(m: nn.Module):
    return next(m.parameters()).device


def pk_conv(owner: PackCache, name: str, conv: nn.Module, **kw) -> ops.PackedConv:
    """pack an nn.Conv2d / nn.Conv3d parameter holder for the MFMA conv kernel"""
    def build():
        w = conv.weight
        if w.dim() == 5:                       # Conv3d holders
            co, ci, kt, kh, kw_ = w.shape
            if kt == 1:                        # (1,3,3): a 2-D conv applied to every frame (a view of the parameter)
                return ops.pack_conv(w.view(co, ci, kh, kw_), conv.bias, stride=1, pad=conv.padding[-1], device=w.device,
                                     param_w=conv.weight, param_b=conv.bias, **kw)
            # (3,1,1) stride 3: 1x1 conv over T*C channels, t-major, gathered straight from the 5-D parameter
            return ops.pack_conv(w, conv.bias, stride=1, pad=0, device=w.device,
                                 layout=ops.convpack.WeightLayout.conv3d_temporal(co, ci, kt), param_w=conv.weight, param_b=conv.bias, **kw)
        return ops.pack_conv(w, conv.bias, stride=conv.stride[0], pad=conv.padding[0], device=w.device, **kw)
    return owner._pk(name, build)


# --------------------------------------------------------------------------------------
class ConvAct(nn.Module):
    """parameter twin of mmcv ConvModule (`conv` + optional `activate`)"""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, padding, bias=True)


class SELayer(nn.Module, PackCache):
    """`main/model/inflate.py:159-208`"""

    def __init__(self, channels, ratio=16):
        super().__init__()
        self.channels = channels
        self.conv1 = ConvAct(channels, int(channels / ratio), 1)
        self.conv2 = ConvAct(int(channels / ratio), channels, 1)

    def params(self) -> ops.SEParams:
        def build():
            c, m = self.channels, self.conv1.conv.out_channels
            f = lambda t: t.detach().float().contiguous()
            return ops.SEParams(f(self.conv1.conv.weight.view(m, c)), f(self.conv1.conv.bias),
                                f(self.conv2.conv.weight.view(c, m)), f(self.conv2.conv.bias), c, m,
                                params=(self.conv1.conv.weight, self.conv1.conv.bias, self.conv2.conv.weight, self.conv2.conv.bias))
        return self._pk("se", build)

    def gate(self, x: FM, partial=None) -> torch.Tensor:
        return ops.se_gate(x, self.params(), partial=partial)

    def run(self, x: FM, out: FM | None = None, act=ACT_NONE, slope=0.0, res: FM | None = None, out2: FM | None = None, sums: list | None = None):
        """`sums`: what `ops.conv(..., chan_sum=sums)` left when it produced `x` (empty: no fused sums, one more pass over x)"""
        out = FM.empty(x.N, x.H, x.W, x.C, dtype=x.t.dtype, device=x.t.device) if out is None else out
        return ops.scale_act_res(x, out, gate=self.gate(x, sums[0] if sums else None), act=act, slope=slope, res=res, out2=out2)


class Res_Block(nn.Module, PackCache):
    """`main/utils/utils.py:43-56`"""

    def __init__(self, channels=64):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, 3, 1, 1)
        self.conv2 = nn.Conv2d(channels, channels, 3, 1, 1)

    def run(self, x: FM, out: FM | None = None, res2: FM | None = None) -> FM:
        # inference: both convs in one launch, the intermediate map stays in LDS (tdvc_conv_pair); under the tape (training)
        # and for shapes the fused kernel does not take, two launches
        if self.conv1.in_channels == 64 and ops.conv_pair_supported(x, out, res2) and (out is None or out.desc().p != x.desc().p):
            pp = self._pk("pair", lambda: ops.pack_conv_pair(self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias))
            return ops.conv_pair(x, pp, out=out, act1=ACT_RELU, act2=ACT_NONE, add_input=True, res2=res2)
        t = ops.conv(x, pk_conv(self, "c1", self.conv1), act=ACT_RELU)
        return ops.conv(t, pk_conv(self, "c2", self.conv2), out=out, res=x, res2=res2)


def res_stack(n, ch=64):
    return nn.Sequential(*[Res_Block(ch) for _ in range(n)])


def run_stack(stack, x: FM, out: FM | None = None, res2: FM | None = None) -> FM:
    n = len(stack)
    for i, blk in enumerate(stack):
        last = i == n - 1
        x = blk.run(x, out=out if last else None, res2=res2 if last else None)
    return x


# --------------------------------------------------------------------------------------
class SPyNetBasicModule(nn.Module):
    def __init__(self):
        super().__init__()
        ch = [8, 32, 64, 32, 16, 2]
        self.basic_module = nn.Sequential(*[ConvAct(ch[i], ch[i + 1], 7, 1, 3) for i in range(5)])


class SPyNet(nn.Module, PackCache):
    """`main/model/flownet.py:51-175`; one fused warp/upsample/concat kernel + five 7x7 MFMA
    convs per pyramid level, flow kept in fp32."""

    def __init__(self):
        super().__init__()
        self.basic_module = nn.ModuleList([SPyNetBasicModule() for _ in range(6)])
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def compute_flow(self, ref: FM, supp: FM) -> FM:
        """ref/supp: fp32 fmaps (C=4), H, W multiples of 32 -> flow fp32 fmap (C=2)"""
        refs, supps = [ref], [supp]
        for _ in range(5):
            refs.append(ops.avgpool2(refs[-1]))
            supps.append(ops.avgpool2(supps[-1]))
        refs, supps = refs[::-1], supps[::-1]
        dev = ref.t.device
        flow = None
        for lvl in range(6):
            r, s = refs[lvl], supps[lvl]
            up = FM.empty(r.N, r.H, r.W, 2, dtype=torch.float32, device=dev)
            x = FM.empty(r.N, r.H, r.W, 8, device=dev)
            ops.spynet_level_input(r, s, flow, up, x)
            bm = self.basic_module[lvl].basic_module
            for i in range(4):
                x = ops.conv(x, pk_conv(self, f"l{lvl}c{i}", bm[i].conv), act=ACT_RELU)
            flow = ops.conv(x, pk_conv(self, f"l{lvl}c4", bm[4].conv), res=up, out_dtype=torch.float32)
        return flow

    def run(self, ref: FM, supp: FM) -> FM:
        H, W = ref.H, ref.W
        Hu = H if H % 32 == 0 else 32 * (H // 32 + 1)
        Wu = W if W % 32 == 0 else 32 * (W // 32 + 1)
        if (Hu, Wu) != (H, W):
            ref, supp = ops.resize_bilinear(ref, Hu, Wu), ops.resize_bilinear(supp, Hu, Wu)
        flow = self.compute_flow(ref, supp)
        if (Hu, Wu) != (H, W):
            sc = torch.tensor([float(W) / float(Wu), float(H) / float(Hu)], dtype=torch.float32, device=flow.t.device)
            flow = ops.resize_bilinear(flow, H, W, sc)
        return flow


# --------------------------------------------------------------------------------------
class FeaExtra(nn.Module, PackCache):
    """`main/model/pnet.py:86-96`"""

    def __init__(self, num_block):
        super().__init__()
        self.conv_first = nn.Conv2d(3, 64, 3, 1, 1)
        self.residual_layer = res_stack(num_block)

    def run(self, img8: FM, out: FM) -> FM:
        x = ops.conv(img8, pk_conv(self, "first", self.conv_first), act=ACT_LRELU, slope=0.1)
        return run_stack(self.residual_layer, x, out=out)


class OffsetGen(nn.Module, PackCache):
    """`main/model/pnet.py:99-167`"""

    def __init__(self, nf=64):
        super().__init__()
        self.offset_conv11 = nn.ModuleDict()
        self.offset_conv11_1 = nn.ModuleDict()
        self.offset_conv12 = nn.ModuleDict()
        self.feat_fusion = nn.ModuleDict()
        for i in (3, 2, 1):
            lv = f"l{i}"
            self.offset_conv11[lv] = nn.Conv2d(nf * 2, nf, 3, 1, 1)
            self.offset_conv11_1[lv] = nn.Conv2d(nf, nf, 3, 1, 1)
            self.offset_conv12[lv] = nn.Conv2d(nf, nf, 3, 1, 1)
            if i < 3:
                self.feat_fusion[lv] = nn.Conv2d(nf * 2, nf, 1, 1, 0)
        self.upsample_conv = nn.Conv2d(nf, nf, 3, 1, 1)
        self.conv_l2_1 = nn.Conv2d(nf, nf, 3, 2, 1)
        self.conv_l2_2 = nn.Conv2d(nf, nf, 3, 1, 1)
        self.conv_l3_1 = nn.Conv2d(nf, nf, 3, 2, 1)
        self.conv_l3_2 = nn.Conv2d(nf, nf, 3, 1, 1)
        self.spynet = SPyNet()
        self.attn = SELayer(64)
        self.feat_fusion_ = nn.Conv2d(nf, nf, 3, 1, 1)

    def run(self, feats: FM, cur32: FM, ref32: FM) -> FM:
        """feats: (B,H,W,>=128) buffer with [f_cur | f_ref] in channels 0..127;
        cur32 / ref32: fp32 RGB fmaps.  Returns estmv (B,H,W,64) fp16."""
        B, H, W = feats.N, feats.H, feats.W
        dev = feats.t.device
        lr = dict(act=ACT_LRELU, slope=0.1)
        cat1 = feats.ch(0, 128)
        cat2 = FM.empty(B, H // 2, W // 2, 128, device=dev)
        cat3 = FM.empty(B, H // 4, W // 4, 128, device=dev)
        for k in (0, 1):   # 0 = current frame features, 1 = reference
            t = ops.conv(cat1.ch(64 * k, 64), pk_conv(self, "l2_1", self.conv_l2_1), **lr)
            ops.conv(t, pk_conv(self, "l2_2", self.conv_l2_2), out=cat2.ch(64 * k, 64), **lr)
            t = ops.conv(cat2.ch(64 * k, 64), pk_conv(self, "l3_1", self.conv_l3_1), **lr)
            ops.conv(t, pk_conv(self, "l3_2", self.conv_l3_2), out=cat3.ch(64 * k, 64), **lr)
        cats = {1: cat1, 2: cat2, 3: cat3}
        up_o1 = None
        off = None
        for i in (3, 2, 1):
            lv = f"l{i}"
            c = cats[i]
            o1 = ops.conv(c, pk_conv(self, "c11" + lv, self.offset_conv11[lv]), **lr)
            if i == 3:
                if ops.conv_pair_supported(o1):            # inference: offset_conv11_1 + offset_conv12 (both LeakyReLU) in one launch
                    c1_, c2_ = self.offset_conv11_1[lv], self.offset_conv12[lv]
                    pp = self._pk("pair_l3", lambda: ops.pack_conv_pair(c1_.weight, c1_.bias, c2_.weight, c2_.bias))
                    off = ops.conv_pair(o1, pp, act1=ACT_LRELU, slope1=0.1, act2=ACT_LRELU, slope2=0.1, add_input=False)
                else:
                    o1 = ops.conv(o1, pk_conv(self, "c11_1" + lv, self.offset_conv11_1[lv]), **lr)
                    off = ops.conv(o1, pk_conv(self, "c12" + lv, self.offset_conv12[lv]), **lr)
            else:
                ops.conv(o1, pk_conv(self, "c11_1" + lv, self.offset_conv11_1[lv]), out=up_o1.ch(64, 64), **lr)
                off = ops.conv(up_o1, pk_conv(self, "ff" + lv, self.feat_fusion[lv]), **lr)
            if i > 1:
                up = ops.upsample2x(off)
                up_o1 = FM.empty(B, up.H, up.W, 128, device=dev)      # [upsampled_offset | offset1]
                ops.conv(up, pk_conv(self, "upc", self.upsample_conv), out=up_o1.ch(0, 64))
        flow = self.spynet.run(cur32, ref32)
        if ops.TAPE is not None:
            off = ops.clone(off)      # keep the fusion conv's own output (LeakyReLU sign) for the backward pass
        ops.add_flow(off, flow)
        sums = []
        e = ops.conv(off, pk_conv(self, "ff_", self.feat_fusion_), chan_sum=sums)
        return self.attn.run(e, sums=sums)


class DCN(nn.Module, PackCache):
    """`main/utils/dcnv2/dcn_v2_amp.py:125-234`: parameters of DCNv2 + conv_offset_mask"""

    def __init__(self, cin, cout, k, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        assert k == 3 and stride == 1 and padding == 1 and dilation == 1
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.zeros(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout))
        self.conv_offset_mask = nn.Conv2d(cin, deformable_groups * 3 * k * k, k, stride, padding)

    def run(self, x: FM, y: FM, out: FM, act=ACT_NONE, slope=0.0) -> FM:
        om = ops.conv(y, pk_conv(self, "om", self.conv_offset_mask))
        pc = self._pk("w", lambda: ops.pack_conv(self.weight, self.bias, stride=1, pad=1, ck=8 * self.deformable_groups,
                                                 device=self.weight.device))
        # output rounded to fp16 BEFORE the activation (reference: `output.half()`, then lrelu on fp16)
        return ops.dcn_fused(x, om, pc, out, groups=self.deformable_groups, act=act, slope=slope, round16=True)


class MCNet(nn.Module, PackCache):
    """`main/model/pnet.py:170-184`"""

    def __init__(self, num_block):
        super().__init__()
        self.dconv = DCN(64, 64, 3, stride=1, padding=1, deformable_groups=8)
        self.recon_layer = res_stack(num_block)
        self.feat_down = nn.Conv2d(64, 3, 3, 1, 1)
        self.conv = nn.Conv2d(128, 64, 3, 1, 1)

    def run(self, offset: FM, feats: FM, out: FM) -> FM:
        """feats: (B,H,W,192) buffer [f_cur | f_ref | dcn_out]; writes prediction1 into `out`."""
        ref = feats.ch(64, 64)
        dcn_out = feats.ch(128, 64)
        self.dconv.run(ref, offset, dcn_out, act=ACT_LRELU, slope=0.1)
        # reference: conv(cat([out, ref])); our buffer order is [ref | out] -> permute input channels
        perm = list(range(64, 128)) + list(range(0, 64))
        pc = self._pk("conv", lambda: ops.pack_conv(self.conv.weight, self.conv.bias, stride=1, pad=1, cin_perm=perm,
                                                    device=self.conv.weight.device))
        o2 = ops.conv(feats.ch(64, 128), pc, act=ACT_LRELU, slope=0.1)
        return run_stack(self.recon_layer, o2, out=out, res2=dcn_out)


class Bottleneck3D(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv3d(64, 64, (1, 3, 3), padding=(0, 1, 1))
        self.spatial_conv3d = nn.Conv3d(64, 64, (1, 3, 3), padding=(0, 1, 1))
        self.temporal_conv3d = nn.Conv3d(64, 64, (3, 1, 1), stride=(3, 1, 1), bias=False)
        self.conv3 = nn.Conv3d(64, 64, (1, 3, 3), padding=(0, 1, 1))


LOOPFILTER_PAIR = True          # A/B switch (tools/ab_infer.py): layer1.conv1 + spatial_conv3d as one tdvc_conv_pair launch
LOOPFILTER_BCAST = True         # A/B switch: temporal conv + broadcast add + LeakyReLU as one conv_mfma_v5 launch (bcast_T)


class LoopFilter(nn.Module, PackCache):
    """`main/model/pnet.py:266-317`: multi-frame feature fusion.  The (B,64,T=4,H,W) tensor of the
    reference is one (B,H,W,256) buffer whose four 64-channel slices are the frames, so the
    Conv3d(1,3,3) layers are 2-D convs over slices and the final flatten is free."""

    def __init__(self):
        super().__init__()
        self.conv01 = nn.Conv2d(3, 64, 3, 1, 1)
        self.conv02 = nn.Conv2d(64, 64, 3, 1, 1)
        self.conv1 = nn.Conv3d(64, 64, (1, 3, 3), padding=(0, 1, 1))
        self.layer1 = Bottleneck3D()
        self.attn = SELayer(64)
        self.feat_fusion = nn.Conv2d(256, 64, 1, 1)

    def _slices(self, name, conv, src: FM, dst: FM, res_buf: FM | None = None, **kw):
        pc = pk_conv(self, name, conv)
        for b in range(src.N):
            ops.conv(src.as_slices(b, 4, 64), pc, out=dst.as_slices(b, 4, 64),
                     res=res_buf.as_slices(b, 4, 64) if res_buf is not None else None, **kw)

    def run(self, xt: FM, refs8: FM, out: FM) -> FM:
        """xt: (B,H,W,256) with prediction1 already in slice 3; refs8: (B*4,H,W,8) fp16 frames
        [I, r-3, r-2, r-1] per batch item; writes the fused prediction into `out`."""
        B, H, W = xt.N, xt.H, xt.W
        dev = xt.t.device
        lr = dict(act=ACT_LRELU, slope=0.1)
        p01, p02 = pk_conv(self, "c01", self.conv01), pk_conv(self, "c02", self.conv02)
        a = FM.empty(B, H, W, 256, device=dev)
        fuse = LOOPFILTER_PAIR and ops.conv_pair_supported(a.as_slices(0, 4, 64).batch(0, 3))
        if fuse:
            # inference: conv02 (no activation) + conv1 (LeakyReLU) of the three reference slices in one launch -- nothing else
            # reads conv02's output; slice 3 (prediction1, already in xt) takes conv1 alone
            pp = self._pk("pair_c02_c1", lambda: ops.pack_conv_pair(self.conv02.weight, self.conv02.bias,
                                                                     self.conv1.weight.view(64, 64, 3, 3), self.conv1.bias))
            pc1 = pk_conv(self, "c1", self.conv1)
        for b in range(B):
            t = ops.conv(refs8.batch(4 * b + 1, 3), p01, **lr)                 # (3,H,W,64)
            if fuse:
                ops.conv_pair(t, pp, out=a.as_slices(b, 4, 64).batch(0, 3), act1=ACT_NONE, act2=ACT_LRELU, slope2=0.1, add_input=False)
                ops.conv(xt.as_slices(b, 4, 64).batch(3, 1), pc1, out=a.as_slices(b, 4, 64).batch(3, 1), **lr)
            else:
                ops.conv(t, p02, out=xt.as_slices(b, 4, 64).batch(0, 3))
        if not fuse:
            self._slices("c1", self.conv1, xt, a, **lr)
        l1 = self.layer1
        bf = FM.empty(B, H, W, 256, device=dev)
        s = FM.empty(B, H, W, 256, device=dev)
        if LOOPFILTER_PAIR and ops.conv_pair_supported(a.as_slices(0, 4, 64), s.as_slices(0, 4, 64)):
            # inference: layer1.conv1 (LeakyReLU) and layer1.spatial_conv3d in one launch per batch item, their intermediate
            # map stays in LDS (`bf` is then only the block's output buffer below)
            pp = self._pk("pair_b1_bs", lambda: ops.pack_conv_pair(l1.conv1.weight.view(64, 64, 3, 3), l1.conv1.bias,
                                                                    l1.spatial_conv3d.weight.view(64, 64, 3, 3), l1.spatial_conv3d.bias))
            for b in range(B):
                ops.conv_pair(a.as_slices(b, 4, 64), pp, out=s.as_slices(b, 4, 64), act1=ACT_LRELU, slope1=0.1, act2=ACT_NONE,
                              add_input=False)
        else:
            self._slices("b1", l1.conv1, a, bf, **lr)
            self._slices("bs", l1.spatial_conv3d, bf, s)
        if ops.TAPE is None and LOOPFILTER_BCAST and H * W >= 8192:      # per IMAGE: conv_v5_eligible (csrc/conv_mfma_v5.hip) counts one map
            # inference: temporal (3,1,1) conv over frames 0-2 + `out + temporal` + LeakyReLU (pnet.py:304-314) as ONE pass over
            # the 4-frame buffer: a wave computes the 64 temporal channels of its pixels and rewrites their four slices in place
            ops.conv(s.ch(0, 192), pk_conv(self, "bt", l1.temporal_conv3d), out=s.ch(0, 64), bcast_T=4, bcast_slope=0.1)
        else:
            tm = ops.conv(s.ch(0, 192), pk_conv(self, "bt", l1.temporal_conv3d))
            if ops.TAPE is not None:
                s = ops.clone(s)          # `s` itself is the temporal conv's input: keep it for the backward pass
            ops.bcast_add_act(s, tm, 4, 0.1)
        # inference re-uses `bf` for the block output; under the tape `bf` is still needed by the backward of `bs`
        o = bf if ops.TAPE is None else FM.empty(B, H, W, 256, device=dev)
        self._slices("b3", l1.conv3, s, o, res_buf=a)
        sums = []
        f = ops.conv(o, pk_conv(self, "ff", self.feat_fusion), chan_sum=sums, **lr)
        return self.attn.run(f, out=out, res=xt.ch(192, 64), sums=sums)


class FeatureExtract(nn.Module, PackCache):
    """`main/model/pnet.py:320-332` (F.leaky_relu default slope 0.01)"""

    def __init__(self, cin, mid, nblocks):
        super().__init__()
        self.conv_first = nn.Conv2d(cin, mid, 3, 1, 1)
        self.body = res_stack(nblocks, mid)
        self.conv_last = nn.Conv2d(mid, mid, 3, 1, 1)

    def run(self, x: FM, out: FM | None = None) -> FM:
        x1 = ops.conv(x, pk_conv(self, "first", self.conv_first), act=ACT_LRELU, slope=0.01)
        t = run_stack(self.body, x1)
        return ops.conv(t, pk_conv(self, "last", self.conv_last), out=out, res=x1)


class FeatureFix(nn.Module, PackCache):
    """`main/model/pnet.py:187-263`: reference-based in-loop filter"""

    def __init__(self):
        super().__init__()
        self.FeatureExtract_input = FeatureExtract(64, 64, 2)
        self.FeatureExtract_ref = FeatureExtract(3, 64, 2)
        self.recon_layer = res_stack(2)
        self.conv_10 = nn.Conv2d(64, 64, 3, 2, 1)
        self.conv_11 = nn.Conv2d(64, 64, 3, 1, 1)
        self.conv_12 = nn.Conv2d(64, 64, 3, 2, 1)
        self.conv_13 = nn.Conv2d(64, 64, 3, 1, 1)
        self.featfusion = nn.Conv2d(128, 64, 3, 1, 1)
        self.featfusion2 = nn.Conv2d(128, 64, 3, 1, 1)
        self.featdown = nn.Conv2d(64, 3, 3, 1, 1)
        self.attn = SELayer(64)

    def run(self, x: FM, iframe8: FM, training: bool, trace=None) -> torch.Tensor:
        """x: reconstructed features (B,H,W,64); iframe8: I-frame (B,H,W,8) fp16.
        Returns the clamped RGB reconstruction as NCHW fp32."""
        B, H, W = x.N, x.H, x.W
        dev = x.t.device
        lr = dict(act=ACT_LRELU, slope=0.1)
        fin = self.FeatureExtract_input.run(x)
        y = FM.empty(B, H, W, 128, device=dev)            # [o | fref]
        fref = self.FeatureExtract_ref.run(iframe8, out=y.ch(64, 64))
        scale = 8 if training else int(H / 8)
        pin, pref = ops.avgpool_k(fin, scale), ops.avgpool_k(fref, scale)
        idx = ops.patch_match(pin, pref)
        cat = FM.empty(B, H, W, 128, device=dev)          # [fin*cor | out*cor]
        ops.match_gather(fin, fref, idx, scale, cat)
        ops.conv(cat, pk_conv(self, "ff", self.featfusion), out=y.ch(0, 64), **lr)
        sums = []
        o2 = ops.conv(y, pk_conv(self, "ff2", self.featfusion2), chan_sum=sums)
        o2 = self.attn.run(o2, sums=sums, **lr)
        o = run_stack(self.recon_layer, o2, res2=x)
        rgb = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
        ops.conv(o, pk_conv(self, "down", self.featdown), act=ops.ACT_CLAMP01, nchw_out=rgb)
        if trace is not None:
            trace.update(ff_idx=idx, ff_fin=fin, ff_fref=fref)
        return rgb
