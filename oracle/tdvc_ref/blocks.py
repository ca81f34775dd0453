"""CPU ORACLE (test infrastructure, never shipped, never timed as the product).

Pure-PyTorch fp32 restatement of the non-entropy blocks of TDVC's P-frame path, with
state-dict keys identical to the reference so one checkpoint / filler serves both.
No mmcv / compressai / `_ext` imports.  Every class cites the reference lines it follows
(paths relative to /root/reference).

Parity status: PINNED for every block in this file against golden vectors produced by
importing the reference's own `main/model/{pnet,flownet,inflate}.py` and
`main/utils/utils.py` in the build container (tests/golden/make_golden.py), and for the
deformable conv against the reference's `check_zero_offset` known-answer test and
`gradcheck` settings (`main/utils/dcnv2/testcpu.py:34-99`).

AMP emulation (`amp_region`, switched on by `codec.VideoCompressor.amp_emulation`): what the
reference computes on a GPU with `enable_amp: True` (`cfg/predict.yaml:7`) inside its three
`torch.cuda.amp.autocast` regions (`main/model/pnet.py:27-31,51-55,75-78`), restated with REAL
fp16 CPU tensors so that torch's own type promotion decides every intermediate dtype as it does
on the GPU: convolutions cast input, weight and bias to fp16, accumulate in fp32 and store fp16
(autocast's fp16 list); element-wise ops, pooling, bilinear resizing, cat run in the dtype(s)
they are given; `grid_sample`, `norm` (inside `F.normalize`) and `cosine_similarity` run in fp32
(autocast's fp32 list).  The deformable conv keeps its explicit `.float()` operands and fp32
weight (`dcn_v2_amp.py:37-42`).  One deliberate deviation: the patch-matching similarity
(`pnet.py:230-236`) stays an fp32 product of the (fp16-valued) pooled features instead of an fp16
`bmm` -- its argmax is discrete, and the rounding of a cuBLAS result is not a function this file
could pin.  Off (the default): every class below is the plain fp32 CPU path, bit for bit what the
golden vectors pin.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# AMP emulation (see the module docstring)
# --------------------------------------------------------------------------------------
class _Amp:
    on = False          # True while inside an emulated `autocast(enabled=True)` region


@contextlib.contextmanager
def amp_region(enabled: bool):
    prev = _Amp.on
    _Amp.on = bool(enabled)
    try:
        yield
    finally:
        _Amp.on = prev


def _h(t):
    return None if t is None else t.to(torch.float16).float()


def _f32(t):
    """autocast's fp32 cast policy (grid_sampler, norm, cosine_similarity)"""
    return t.float() if _Amp.on else t


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose forward follows autocast's fp16 policy inside an `amp_region`: operands rounded to fp16, fp32
    accumulation, fp16 result.  State-dict keys and the fp32 path are nn.Conv2d's."""

    def forward(self, x):
        if not _Amp.on:
            return super().forward(x)
        return F.conv2d(_h(x), _h(self.weight), _h(self.bias), self.stride, self.padding, self.dilation, self.groups).half()


class Conv3d(nn.Conv3d):
    def forward(self, x):
        if not _Amp.on:
            return super().forward(x)
        return F.conv3d(_h(x), _h(self.weight), _h(self.bias), self.stride, self.padding, self.dilation, self.groups).half()


# --------------------------------------------------------------------------------------
# small shared pieces
# --------------------------------------------------------------------------------------
class ConvAct(nn.Module):
    """State-dict twin of mmcv's ConvModule as used by the reference
    (`main/model/flownet.py:187-227`, `main/model/inflate.py:189-202`): attribute
    `conv` (Conv2d, bias=True because no norm layer) and optional `activate`."""

    def __init__(self, cin, cout, k, stride=1, padding=0, act=None):
        super().__init__()
        self.conv = Conv2d(cin, cout, k, stride, padding, bias=True)
        if act == "relu":
            self.activate = nn.ReLU()
        elif act == "sigmoid":
            self.activate = nn.Sigmoid()
        else:
            self.activate = None

    def forward(self, x):
        x = self.conv(x)
        return x if self.activate is None else self.activate(x)


class SELayer(nn.Module):
    """`main/model/inflate.py:159-208`: global avg-pool -> 1x1 C->C/16 ReLU ->
    1x1 ->C Sigmoid -> channel scale."""

    def __init__(self, channels, ratio=16):
        super().__init__()
        self.conv1 = ConvAct(channels, int(channels / ratio), 1, act="relu")
        self.conv2 = ConvAct(int(channels / ratio), channels, 1, act="sigmoid")

    def gate(self, x):
        return self.conv2(self.conv1(x.mean(dim=(2, 3), keepdim=True)))

    def forward(self, x):
        return x * self.gate(x)


class Res_Block(nn.Module):
    """`main/utils/utils.py:43-56`: x + conv2(relu(conv1(x)))."""

    def __init__(self, channels=64):
        super().__init__()
        self.conv1 = Conv2d(channels, channels, 3, 1, 1)
        self.conv2 = Conv2d(channels, channels, 3, 1, 1)

    def forward(self, x):
        return x + self.conv2(F.relu(self.conv1(x)))


def res_stack(n, ch=64):
    return nn.Sequential(*[Res_Block(ch) for _ in range(n)])


def pad_to(x, p=64):
    """`main/utils/utils.py:59-72`: centred zero pad to a multiple of p."""
    h, w = x.shape[-2:]
    H, W = (h + p - 1) // p * p, (w + p - 1) // p * p
    l = (W - w) // 2
    t = (H - h) // 2
    return F.pad(x, (l, W - w - l, t, H - h - t))


def crop_to(x, size):
    """`main/utils/utils.py:75-87`: inverse of pad_to."""
    H, W = x.shape[-2:]
    h, w = size
    l = (W - w) // 2
    t = (H - h) // 2
    return x[..., t:t + h, l:l + w]


def split_optim_params(net):
    """`main/utils/utils.py:90-113`: (main, aux) name lists; aux = names ending `.quantiles`."""
    main = sorted(n for n, p in net.named_parameters() if not n.endswith(".quantiles") and p.requires_grad)
    aux = sorted(n for n, p in net.named_parameters() if n.endswith(".quantiles") and p.requires_grad)
    return main, aux


# --------------------------------------------------------------------------------------
# SPyNet (`main/model/flownet.py`)
# --------------------------------------------------------------------------------------
def flow_warp_border(x, flow_nhw2):
    """`main/model/flownet.py:8-48` with padding_mode='border', align_corners=True
    (the only configuration the hot path uses, `:134-137`)."""
    n, c, h, w = x.shape
    gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    gx = gx.to(x) + flow_nhw2[..., 0]
    gy = gy.to(x) + flow_nhw2[..., 1]
    nx = 2.0 * gx / max(w - 1, 1) - 1.0
    ny = 2.0 * gy / max(h - 1, 1) - 1.0
    return F.grid_sample(_f32(x), _f32(torch.stack((nx, ny), dim=3)), mode="bilinear",
                         padding_mode="border", align_corners=True)


class SPyNetBasicModule(nn.Module):
    """`main/model/flownet.py:178-238`: 7x7 convs 8->32->64->32->16->2, ReLU between."""

    def __init__(self):
        super().__init__()
        chans = [8, 32, 64, 32, 16, 2]
        self.basic_module = nn.Sequential(*[
            ConvAct(chans[i], chans[i + 1], 7, 1, 3, act="relu" if i < 4 else None) for i in range(5)
        ])

    def forward(self, x):
        return self.basic_module(x)


class SPyNet(nn.Module):
    """`main/model/flownet.py:51-175`.  mean/std buffers exist but normalisation is
    disabled in the reference (`:96-97`)."""

    def __init__(self):
        super().__init__()
        self.basic_module = nn.ModuleList([SPyNetBasicModule() for _ in range(6)])
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def compute_flow(self, ref, supp):
        n, _, h, w = ref.shape
        refs, supps = [ref], [supp]
        for _ in range(5):
            refs.append(F.avg_pool2d(refs[-1], 2, 2, count_include_pad=False))
            supps.append(F.avg_pool2d(supps[-1], 2, 2, count_include_pad=False))
        refs, supps = refs[::-1], supps[::-1]
        flow = ref.new_zeros(n, 2, h // 32, w // 32)
        for lvl in range(6):
            if lvl == 0:
                up = flow
            else:
                up = F.interpolate(flow, scale_factor=2, mode="bilinear", align_corners=True) * 2.0
            warped = flow_warp_border(supps[lvl], up.permute(0, 2, 3, 1))
            flow = up + self.basic_module[lvl](torch.cat([refs[lvl], warped, up], 1))
        return flow

    def forward(self, ref, supp):
        h, w = ref.shape[2:]
        wu = w if w % 32 == 0 else 32 * (w // 32 + 1)
        hu = h if h % 32 == 0 else 32 * (h // 32 + 1)
        ref = F.interpolate(ref, size=(hu, wu), mode="bilinear", align_corners=False)
        supp = F.interpolate(supp, size=(hu, wu), mode="bilinear", align_corners=False)
        flow = F.interpolate(self.compute_flow(ref, supp), size=(h, w), mode="bilinear", align_corners=False)
        flow = flow.clone()
        flow[:, 0] *= float(w) / float(wu)
        flow[:, 1] *= float(h) / float(hu)
        return flow


# --------------------------------------------------------------------------------------
# Modulated deformable conv (DCNv2) — gather formulation
# --------------------------------------------------------------------------------------
def dcn_v2_forward_ref(x, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, groups):
    """fp32 restatement of `_ext.dcn_v2_forward`:
    sampling per `src/cuda/dcn_v2_im2col_cuda.cu:125-195` (bilinear `:25-54`, open
    interval test `:180`), then `bias + W_flat @ columns` per `dcn_v2_cuda.cu:69-92`
    (NOT the CPU wrapper, which adds the bias into uninitialised memory,
    `src/cpu/dcn_v2_cpu.cpp:65,107-110`).  Columns are never materialised: one tap
    at a time is sampled and contracted.
    """
    B, C, H, W = x.shape
    Cout = weight.shape[0]
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    cpg = C // groups
    out = bias.view(1, Cout, 1, 1).expand(B, Cout, Ho, Wo).clone()
    hs = (torch.arange(Ho, dtype=x.dtype) * sh - ph).view(1, 1, Ho, 1)
    ws = (torch.arange(Wo, dtype=x.dtype) * sw - pw).view(1, 1, 1, Wo)
    off = offset.view(B, groups, kh * kw, 2, Ho, Wo)
    msk = mask.view(B, groups, kh * kw, Ho, Wo)
    xg = x.view(B, groups, cpg, H * W)
    for i in range(kh):
        for j in range(kw):
            t = i * kw + j
            h_im = hs + i * dh + off[:, :, t, 0]          # (B, G, Ho, Wo)
            w_im = ws + j * dw + off[:, :, t, 1]
            inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
            h_low = torch.floor(h_im)
            w_low = torch.floor(w_im)
            lh, lw = h_im - h_low, w_im - w_low
            hh, hw = 1 - lh, 1 - lw
            h_low, w_low = h_low.long(), w_low.long()
            h_high, w_high = h_low + 1, w_low + 1

            def corner(hi, wi, ok):
                ok = ok & inside
                idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(B, groups, 1, Ho * Wo)
                v = torch.gather(xg, 3, idx.expand(B, groups, cpg, Ho * Wo))
                return v * ok.view(B, groups, 1, Ho * Wo).to(x.dtype)

            v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
            v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
            v3 = corner(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
            v4 = corner(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))
            f = lambda a: a.reshape(B, groups, 1, Ho * Wo)
            val = f(hh * hw) * v1 + f(hh * lw) * v2 + f(lh * hw) * v3 + f(lh * lw) * v4
            col = (val * f(msk[:, :, t])).view(B, C, Ho * Wo)
            out += torch.matmul(weight[:, :, i, j], col).view(B, Cout, Ho, Wo)
    return out


class DCN(nn.Module):
    """`main/utils/dcnv2/dcn_v2_amp.py:125-234` (`DCNv2` + `DCN`).  Output is rounded
    to fp16 unconditionally (`:15,67-68`: module-global `use_amp = True`)."""

    def __init__(self, cin, cout, k, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        self.k, self.stride, self.padding, self.dilation = k, stride, padding, dilation
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.zeros(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout))
        self.conv_offset_mask = Conv2d(cin, deformable_groups * 3 * k * k, k, stride, padding)

    def offsets_and_mask(self, y):
        o = self.conv_offset_mask(y)
        o1, o2, m = torch.chunk(o, 3, dim=1)
        return torch.cat((o1, o2), 1), torch.sigmoid(m)

    def forward(self, x, y):
        offset, mask = self.offsets_and_mask(y)
        k, s, p, d = self.k, self.stride, self.padding, self.dilation
        out = dcn_v2_forward_ref(x.float(), self.weight.float(), self.bias.float(), offset.float(),
                                 mask.float(), k, k, s, s, p, p, d, d, self.deformable_groups)
        return out.half()


# --------------------------------------------------------------------------------------
# blocks of `main/model/pnet.py`
# --------------------------------------------------------------------------------------
class FeaExtra(nn.Module):
    """`main/model/pnet.py:86-96`."""

    def __init__(self, num_block):
        super().__init__()
        self.conv_first = Conv2d(3, 64, 3, 1, 1)
        self.residual_layer = res_stack(num_block)

    def forward(self, x):
        return self.residual_layer(F.leaky_relu(self.conv_first(x), 0.1))


class OffsetGen(nn.Module):
    """`main/model/pnet.py:99-167`: multi-scale motion estimation."""

    def __init__(self, nf=64):
        super().__init__()
        self.offset_conv11 = nn.ModuleDict()
        self.offset_conv11_1 = nn.ModuleDict()
        self.offset_conv12 = nn.ModuleDict()
        self.feat_fusion = nn.ModuleDict()
        for i in (3, 2, 1):
            lv = f"l{i}"
            self.offset_conv11[lv] = Conv2d(nf * 2, nf, 3, 1, 1)
            self.offset_conv11_1[lv] = Conv2d(nf, nf, 3, 1, 1)
            self.offset_conv12[lv] = Conv2d(nf, nf, 3, 1, 1)      # l2/l1 are dead params
            if i < 3:
                self.feat_fusion[lv] = Conv2d(nf * 2, nf, 1, 1, 0)
        self.upsample_conv = Conv2d(nf, nf, 3, 1, 1)
        self.conv_l2_1 = Conv2d(nf, nf, 3, 2, 1)
        self.conv_l2_2 = Conv2d(nf, nf, 3, 1, 1)
        self.conv_l3_1 = Conv2d(nf, nf, 3, 2, 1)
        self.conv_l3_2 = Conv2d(nf, nf, 3, 1, 1)
        self.spynet = SPyNet()
        self.attn = SELayer(64)
        self.feat_fusion_ = Conv2d(nf, nf, 3, 1, 1)

    def pyramid(self, f):
        a = lambda t: F.leaky_relu(t, 0.1)
        l2 = a(self.conv_l2_2(a(self.conv_l2_1(f))))
        l3 = a(self.conv_l3_2(a(self.conv_l3_1(l2))))
        return [f, l2, l3]

    def forward(self, cur_f, ref_f, cur_img, ref_img):
        a = lambda t: F.leaky_relu(t, 0.1)
        cur, ref = self.pyramid(cur_f), self.pyramid(ref_f)
        up = None
        for i in (3, 2, 1):
            lv = f"l{i}"
            o1 = a(self.offset_conv11[lv](torch.cat([cur[i - 1], ref[i - 1]], 1)))
            o1 = a(self.offset_conv11_1[lv](o1))
            if i == 3:
                off = a(self.offset_conv12[lv](o1))
            else:
                off = a(self.feat_fusion[lv](torch.cat([up, o1], 1)))
            if i > 1:
                up = F.interpolate(off, scale_factor=2, mode="bilinear", align_corners=False)
                up = self.upsample_conv(up)
        flow = self.spynet(cur_img, ref_img)
        off = off + flow.repeat(1, off.shape[1] // 2, 1, 1)
        return self.attn(self.feat_fusion_(off))


class MCNet(nn.Module):
    """`main/model/pnet.py:170-184`: DCN motion compensation + refinement."""

    def __init__(self, num_block):
        super().__init__()
        self.dconv = DCN(64, 64, 3, stride=1, padding=1, deformable_groups=8)
        self.recon_layer = res_stack(num_block)
        self.feat_down = Conv2d(64, 3, 3, 1, 1)      # dead param
        self.conv = Conv2d(128, 64, 3, 1, 1)

    def forward(self, offset, ref):
        # DCN output is fp16; LeakyReLU then runs on the fp16 tensor; cat/add promote to fp32
        out = F.leaky_relu(self.dconv(ref, offset), 0.1)
        if not _Amp.on:
            out = out.float()          # CPU path: cat / add with the fp32 features promote (under AMP everything here is fp16)
        out2 = F.leaky_relu(self.conv(torch.cat([out, ref], 1)), 0.1)
        return out + self.recon_layer(out2)


class Bottleneck3D(nn.Module):
    """`main/model/pnet.py:296-317`."""

    def __init__(self):
        super().__init__()
        self.conv1 = Conv3d(64, 64, (1, 3, 3), padding=(0, 1, 1))
        self.spatial_conv3d = Conv3d(64, 64, (1, 3, 3), padding=(0, 1, 1))
        self.temporal_conv3d = Conv3d(64, 64, (3, 1, 1), stride=(3, 1, 1), bias=False)
        self.conv3 = Conv3d(64, 64, (1, 3, 3), padding=(0, 1, 1))

    def forward(self, x):
        o = F.leaky_relu(self.conv1(x), 0.1)
        o = self.spatial_conv3d(o)
        o = F.leaky_relu(o + self.temporal_conv3d(o), 0.1)
        return self.conv3(o) + x


class LoopFilter(nn.Module):
    """`main/model/pnet.py:266-293`: the multi-frame feature FUSION block (sic)."""

    def __init__(self):
        super().__init__()
        self.conv01 = Conv2d(3, 64, 3, 1, 1)
        self.conv02 = Conv2d(64, 64, 3, 1, 1)
        self.conv1 = Conv3d(64, 64, (1, 3, 3), padding=(0, 1, 1))
        self.layer1 = Bottleneck3D()
        self.attn = SELayer(64)
        self.feat_fusion = Conv2d(256, 64, 1, 1)

    def forward(self, pred, refs):
        r = refs[:, 1:]
        N, M, C, H, W = r.shape
        r = self.conv02(F.leaky_relu(self.conv01(r.reshape(N * M, 3, H, W)), 0.1)).view(N, M, 64, H, W)
        x = torch.cat((r, pred.unsqueeze(1)), 1).permute(0, 2, 1, 3, 4)
        x = self.layer1(F.leaky_relu(self.conv1(x), 0.1))
        x = x.permute(0, 2, 1, 3, 4).reshape(N, -1, H, W)
        x = self.attn(F.leaky_relu(self.feat_fusion(x), 0.1))
        return pred + x


class FeatureExtract(nn.Module):
    """`main/model/pnet.py:320-332`; note F.leaky_relu default slope 0.01 (`:328`)."""

    def __init__(self, cin, mid, nblocks):
        super().__init__()
        self.conv_first = Conv2d(cin, mid, 3, 1, 1)
        self.body = res_stack(nblocks, mid)
        self.conv_last = Conv2d(mid, mid, 3, 1, 1)

    def forward(self, x):
        x1 = F.leaky_relu(self.conv_first(x))
        return self.conv_last(self.body(x1)) + x1


class FeatureFix(nn.Module):
    """`main/model/pnet.py:187-263`: the reference-based IN-LOOP FILTER (sic), pooled
    3x3-patch cosine matching against the I-frame."""

    def __init__(self):
        super().__init__()
        self.FeatureExtract_input = FeatureExtract(64, 64, 2)
        self.FeatureExtract_ref = FeatureExtract(3, 64, 2)
        self.recon_layer = res_stack(2)
        self.conv_10 = Conv2d(64, 64, 3, 2, 1)   # conv_10..13: dead params
        self.conv_11 = Conv2d(64, 64, 3, 1, 1)
        self.conv_12 = Conv2d(64, 64, 3, 2, 1)
        self.conv_13 = Conv2d(64, 64, 3, 1, 1)
        self.featfusion = Conv2d(128, 64, 3, 1, 1)
        self.featfusion2 = Conv2d(128, 64, 3, 1, 1)
        self.featdown = Conv2d(64, 3, 3, 1, 1)
        self.attn = SELayer(64)

    def match(self, fin, fref, scale):
        """-> (index (N, L) of best ref patch per input patch, gathered full-res map)."""
        N, C, H, W = fin.shape
        pin = F.avg_pool2d(fin, scale, scale)
        pref = F.avg_pool2d(fref, scale, scale)
        a = F.unfold(pin, 3, padding=3, stride=3).transpose(2, 1)            # (N, L, C*9)
        b = F.unfold(pref, 3, padding=3, stride=3).transpose(2, 1).reshape(N, -1, C * 9)
        a, b = _f32(a), _f32(b)          # AMP emulation: fp32 similarity of the fp16-valued pooled features (module docstring)
        sim = torch.bmm(F.normalize(a, dim=2), F.normalize(b.transpose(2, 1), dim=1))
        _, ind = sim.max(dim=2, keepdim=True)
        ks = 3 * scale
        ru = F.unfold(fref, ks, padding=ks, stride=ks).transpose(2, 1).reshape(N, -1, C * ks * ks)
        idx = ind.view(N, 1, -1).expand(-1, C * ks * ks, -1).permute(0, 2, 1)
        g = torch.gather(ru, 1, idx).view(N, -1, C, ks, ks).permute(0, 2, 3, 4, 1).reshape(N, -1, a.shape[1])
        out = F.fold(g, (H, W), ks, padding=ks, stride=ks)
        return ind.view(N, -1), out

    def forward(self, x, refs):
        N, C, H, W = x.shape
        iframe = refs[:, 0].reshape(-1, 3, H, W)
        fin = self.FeatureExtract_input(x)
        fref = self.FeatureExtract_ref(iframe)
        scale = 8 if self.training else int(fin.shape[2] / 8)
        ind, out = self.match(fin, fref, scale)
        self.last_match_index = ind          # (N, L) argmax per input patch: what the parity tests compare bit for bit
        cor = F.cosine_similarity(_f32(fin), _f32(out)).unsqueeze(1)
        o = F.leaky_relu(self.featfusion(torch.cat([fin, out], 1) * cor), 0.1)
        o = F.leaky_relu(self.attn(self.featfusion2(torch.cat([o, fref], 1))), 0.1)
        o = self.recon_layer(o)
        return self.featdown(x + o)
