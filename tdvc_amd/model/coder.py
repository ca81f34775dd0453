"""Motion / residual auto-encoders with hyperprior + context entropy model, MI355X-native.

Parameter layout mirrors `MVCoder` / `ResCoder` of `main/model/encoder_v3.py:14-69`, i.e.
CompressAI's `Cheng2020Anchor(N=128)` with g_a/g_s replaced (state-dict names of CompressAI
1.1.x; compressai itself is not a dependency).  All transforms run through the MFMA conv
kernel; GDN is the same kernel with a squared-input prologue and a `x * rsqrt(.)` epilogue;
PixelShuffle is a store pattern; the rate terms are fused elementwise kernels.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..ops import ACT_LRELU, ACT_NONE, FM, GDN_FWD, GDN_INV
from .modules import PackCache, SELayer, pk_conv


# ------------------------------------------------------------------ parameter holders
class _LowerBoundGrad(torch.autograd.Function):
    """max(x, bound) whose backward follows compressai's `LowerBoundFunction` (ops/bound_ops.py): the gradient passes
    where `x >= bound` or `grad < 0`.  Parameter-space use only (GDN beta / gamma chain rule); the activation-space
    bounds (likelihood floor, scale bound) apply the same rule inside `tdvc_eb_backward` / `tdvc_gc_backward`."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        return ((x >= bound) | (g < 0)).to(g.dtype) * g, None


class LowerBound(nn.Module):
    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return _LowerBoundGrad.apply(x, self.bound)


class NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum=0.0, reparam_offset=2 ** -18):
        super().__init__()
        ped = float(reparam_offset) ** 2
        self.register_buffer("pedestal", torch.Tensor([ped]))
        self.lower_bound = LowerBound((float(minimum) + ped) ** 0.5)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def effective(self, x):
        return self.lower_bound(x) ** 2 - self.pedestal


class GDN(nn.Module, PackCache):
    def __init__(self, ch, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = NonNegativeParametrizer(minimum=beta_min)
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(ch)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(gamma_init * torch.eye(ch)))

    def run(self, x: FM, out: FM | None = None, res: FM | None = None) -> FM:
        """out = x * (r)sqrt(beta + gamma @ x^2) + res"""
        def build():
            g = self.gamma_reparam.effective(self.gamma.detach())
            b = self.beta_reparam.effective(self.beta.detach())
            c = g.shape[0]
            pc = ops.pack_conv(g.reshape(c, c, 1, 1).contiguous(), b.contiguous(), stride=1, pad=0, device=self.gamma.device)
            pc.owner = self
            return pc
        pc = self._pk("gdn", build)
        return ops.conv(x, pc, out=out, square=True, gdn=GDN_INV if self.inverse else GDN_FWD, aux=x, res=res)

    def refresh_packed(self):
        """after an optimizer step: effective gamma / beta recomputed into the packed layer's source tensors, re-packed"""
        pc = self.__dict__.get("_packed", {}).get("gdn")
        if pc is not None:
            with torch.no_grad():
                pc.wsrc.copy_(self.gamma_reparam.effective(self.gamma).reshape(pc.wsrc.shape))
                pc.bsrc.copy_(self.beta_reparam.effective(self.beta))
            pc.repack()

    def accumulate_param_grads(self, dgamma_eff: torch.Tensor, dbeta_eff: torch.Tensor):
        """chain rule through the non-negative reparametrisation (parameter space, 128 x 128 values: plain autograd)"""
        with torch.enable_grad():
            g = self.gamma_reparam.effective(self.gamma)
            b = self.beta_reparam.effective(self.beta)
            torch.autograd.backward([g, b], [dgamma_eff.to(g.dtype), dbeta_eff.to(b.dtype)])


def gdn_refresh_batched(gdns) -> list:
    """`GDN.refresh_packed` for MANY layers at once (training: 24 layers per step, each ~10 tiny torch launches + its own packing launch): the
    effective gamma / beta of all layers from two stacked tensors, written into the packed layers' source tensors by one multi-tensor copy
    each; -> the packed layers, for the model's one batched re-pack (ops.PackBatch).  Same element-wise arithmetic as `effective()`."""
    gdns = [m for m in gdns if m.__dict__.get("_packed", {}).get("gdn") is not None]
    if not gdns:
        return []
    pcs = [m.__dict__["_packed"]["gdn"] for m in gdns]
    with torch.no_grad():
        for attr, rep, dst in (("gamma", "gamma_reparam", "wsrc"), ("beta", "beta_reparam", "bsrc")):
            rp = getattr(gdns[0], rep)
            eff = torch.maximum(torch.stack([getattr(m, attr) for m in gdns]), rp.lower_bound.bound) ** 2 - rp.pedestal
            torch._foreach_copy_([getattr(pc, dst).view(getattr(m, attr).shape) for pc, m in zip(pcs, gdns)], list(eff.unbind(0)))
    return pcs


def gdn_chain_batched(items, grad_of) -> None:
    """`GDN.accumulate_param_grads` for MANY layers at once: the chain rule through `effective(p) = max(p, bound)^2 - pedestal` with
    compressai's LowerBound gradient rule (pass where p >= bound or the incoming gradient is negative) in closed form on stacked tensors --
    what torch autograd computes layer by layer (d_lb = d_eff * 2 lb: the same products), a dozen launches instead of ~20 per layer.
    items: [(gdn, d gamma_eff (C, C), d beta_eff (C,))]; grad_of(param) -> the fp32 gradient accumulator of a parameter."""
    if not items:
        return
    with torch.no_grad():
        for k, attr, rep in ((1, "gamma", "gamma_reparam"), (2, "beta", "beta_reparam")):
            rp = getattr(items[0][0], rep)
            bound = rp.lower_bound.bound
            P = torch.stack([getattr(it[0], attr) for it in items])
            D = torch.stack([it[k].to(P.dtype).view(getattr(it[0], attr).shape) for it in items])
            d_lb = D * (torch.maximum(P, bound) * 2.0)
            G = d_lb * ((P >= bound) | (d_lb < 0)).to(d_lb.dtype)
            torch._foreach_add_([grad_of(getattr(it[0], attr)) for it in items], list(G.unbind(0)))


def conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride, 1)


def conv1x1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 1, stride)


def subpel_conv3x3(cin, cout, r=1):
    return nn.Sequential(nn.Conv2d(cin, cout * r ** 2, 3, padding=1), nn.PixelShuffle(r))


LR = dict(act=ACT_LRELU, slope=0.01)
STREAM_ORDERS = ("raster", "wavefront")

# host range coding off the critical path (compress(defer=True), VideoCompressor.encode): the coder library releases the GIL
_RANS_POOL = None


def _rans_pool():
    global _RANS_POOL
    if _RANS_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _RANS_POOL = ThreadPoolExecutor(max_workers=2, thread_name_prefix="tdvc-rans")
    return _RANS_POOL


class PendingStrings:
    """the byte strings of a compress(defer=True) call: `result()` -> [y_strings, z_strings] once the worker threads are done"""

    def __init__(self, y_futs, z_strings):
        self._y, self._z = y_futs, z_strings

    def result(self):
        return [[f.result() for f in self._y], self._z]


class ResidualBlock(nn.Module, PackCache):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = conv3x3(cin, cout)
        self.conv2 = conv3x3(cout, cout)
        self.skip = conv1x1(cin, cout) if cin != cout else None

    def run(self, x: FM, out: FM | None = None) -> FM:
        t = ops.conv(x, pk_conv(self, "c1", self.conv1), **LR)
        idt = x if self.skip is None else ops.conv(x, pk_conv(self, "sk", self.skip))
        return ops.conv(t, pk_conv(self, "c2", self.conv2), out=out, res=idt, **LR)


class ResidualBlockWithStride(nn.Module, PackCache):
    def __init__(self, cin, cout, stride=2):
        super().__init__()
        self.conv1 = conv3x3(cin, cout, stride)
        self.conv2 = conv3x3(cout, cout)
        self.gdn = GDN(cout)
        self.skip = conv1x1(cin, cout, stride) if (stride != 1 or cin != cout) else None

    def run(self, x: FM, out: FM | None = None) -> FM:
        t = ops.conv(x, pk_conv(self, "c1", self.conv1), **LR)
        t = ops.conv(t, pk_conv(self, "c2", self.conv2))
        idt = x if self.skip is None else ops.conv(x, pk_conv(self, "sk", self.skip))
        return self.gdn.run(t, out=out, res=idt)


class ResidualBlockUpsample(nn.Module, PackCache):
    def __init__(self, cin, cout, upsample=2):
        super().__init__()
        assert upsample == 2
        self.subpel_conv = subpel_conv3x3(cin, cout, upsample)
        self.conv = conv3x3(cout, cout)
        self.igdn = GDN(cout, inverse=True)
        self.upsample = subpel_conv3x3(cin, cout, upsample)

    def run(self, x: FM, out: FM | None = None) -> FM:
        t = ops.conv(x, pk_conv(self, "sp", self.subpel_conv[0], shuffle=True), **LR)
        t = ops.conv(t, pk_conv(self, "c", self.conv))
        up = ops.conv(x, pk_conv(self, "up", self.upsample[0], shuffle=True))
        return self.igdn.run(t, out=out, res=up)


class MaskedConv2d(nn.Conv2d):
    """type-A mask buffer kept for state-dict compatibility; masked taps are dropped at pack time"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("mask", torch.ones_like(self.weight.data))
        _, _, h, w = self.mask.shape
        self.mask[:, :, h // 2, w // 2:] = 0
        self.mask[:, :, h // 2 + 1:] = 0

    def live_taps(self):
        h, w = self.kernel_size
        return [(dy, dx) for dy in range(h) for dx in range(w) if dy < h // 2 or (dy == h // 2 and dx < w // 2)]


class _CdfBuffers:
    """mixin: the coder tables (`_quantized_cdf`, `_offset`, `_cdf_length`, `scale_table`) are buffers that are EMPTY
    until `update()` runs; a checkpoint saved after an update carries them filled.  Like compressai's
    `update_registered_buffers`, loading resizes the module's buffers to whatever the checkpoint holds, so
    `load_state_dict(strict=True)` works in both directions (tools/predict.py:150)."""

    _cdf_buffer_names = ("_quantized_cdf", "_offset", "_cdf_length", "scale_table")

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for n in self._cdf_buffer_names:
            src = state_dict.get(prefix + n)
            buf = self._buffers.get(n)
            if src is not None and buf is not None and buf.shape != src.shape:
                self._buffers[n] = torch.empty(src.shape, dtype=buf.dtype, device=buf.device)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class EntropyBottleneck(_CdfBuffers, nn.Module, PackCache):
    """factorised prior parameters (CompressAI 1.1.x names)"""

    def __init__(self, channels, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3), likelihood_bound=1e-9):
        super().__init__()
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        f = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / f[i + 1]))
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(torch.full((channels, f[i + 1], f[i]), float(init))))
            self.register_parameter(f"_bias{i:d}", nn.Parameter(torch.zeros(channels, f[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter(f"_factor{i:d}", nn.Parameter(torch.zeros(channels, f[i + 1], 1)))
        q = torch.Tensor([-self.init_scale, 0, self.init_scale])
        self.quantiles = nn.Parameter(q.repeat(channels, 1, 1))
        target = np.log(2 / self.tail_mass - 1)
        # compressai EntropyBottleneck.__init__: the three quantiles are driven to the logits (-t, 0, +t), i.e. to the tail_mass / 2,
        # 1 / 2 and 1 - tail_mass / 2 quantiles.  (Rounds 1-3 registered the single value [t]: with it all three quantiles chase the
        # upper tail and the median column drifts -- found in round 4 when the auxiliary loss became a kernel.)
        self.register_buffer("target", torch.Tensor([-target, 0, target]))

    def _packed_tensor(self) -> torch.Tensor:
        """[C][59] fp32: softplus(matrix0..4) | bias0..4 | tanh(factor0..3) | median"""
        C_ = self.channels
        parts = [F.softplus(getattr(self, f"_matrix{i}")).reshape(C_, -1) for i in range(5)]
        parts += [getattr(self, f"_bias{i}").reshape(C_, -1) for i in range(5)]
        parts += [torch.tanh(getattr(self, f"_factor{i}")).reshape(C_, -1) for i in range(4)]
        parts += [self.quantiles[:, 0, 1:2]]
        p = torch.cat(parts, 1).float().contiguous()
        assert p.shape == (C_, 59)
        return p

    def packed_params(self) -> torch.Tensor:
        def build():
            with torch.no_grad():
                p = self._packed_tensor()
            p.owner = self
            return p
        return self._pk("eb", build)

    def _raw_params(self):
        """the 14 parameter tensors behind the packed table, in its column order"""
        return ([getattr(self, f"_matrix{i}") for i in range(5)] + [getattr(self, f"_bias{i}") for i in range(5)] +
                [getattr(self, f"_factor{i}") for i in range(4)])

    def _ptr_table(self, name, tensors):
        """device array of the tensors' addresses (tdvc_eb_pack / tdvc_eb_param_chain), rebuilt when one of them moved"""
        ptrs = tuple(t.data_ptr() for t in tensors)
        c = self.__dict__.get(name)
        if c is None or c[0] != ptrs:
            assert all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors)
            c = self.__dict__[name] = (ptrs, torch.tensor(ptrs, dtype=torch.int64, device=tensors[0].device))
        return c[1]

    def refresh_packed(self):
        """after an optimizer step: the packed table follows the parameters -- in place, one launch (tdvc_eb_pack), when it exists"""
        p = self.__dict__.get("_packed", {}).get("eb")
        if p is None:
            return
        if not p.is_cuda:
            self.__dict__["_packed"].pop("eb", None)
            return
        raw = self._raw_params()
        ops.eb_pack(self._ptr_table("_raw_table", [t.data for t in raw]), self.quantiles.data, p, self.channels)

    def accumulate_param_grads(self, dpacked: torch.Tensor, scale: float = 1.0):
        """chain rule through softplus / tanh into the 14 raw parameters (128 x 58 values); the median column (quantiles) only receives
        the auxiliary loss.  On the device: one launch (tdvc_eb_param_chain) instead of autograd over ~40 element-wise kernels."""
        if dpacked.is_cuda:
            from ..autograd import param_grad
            raw = self._raw_params()
            grads = [param_grad(t) for t in raw]
            ops.eb_param_chain(dpacked, self._ptr_table("_raw_table", [t.data for t in raw]), self._ptr_table("_grad_table", grads), scale, self.channels)
            return
        with torch.enable_grad():
            p = self._packed_tensor()
            d = dpacked * scale
            d[:, 58] = 0.0
            torch.autograd.backward([p], [d])

    @torch.no_grad()
    def loss_fused(self, write_grad: bool = True) -> torch.Tensor:
        """`loss()` and its gradient in one launch (tdvc_eb_aux): -> the loss as a 1-element fp32 tensor; with `write_grad`
        `quantiles.grad` is OVERWRITTEN with d loss / d quantiles (otherwise the gradient goes to a scratch buffer: the training-mode
        forward under the tape only reports the value).  The matrices / biases / factors are those of the packed table, i.e. of the
        last forward -- what the reference's `aux_loss.backward()` differentiates too (its graph was built in the forward,
        tools/train.py:150)."""
        q = self.quantiles
        if write_grad:
            if q.grad is None or q.grad.shape != q.shape:
                q.grad = torch.empty_like(q)
            dq = q.grad
        else:
            dq = self.__dict__.get("_dq_scratch")
            if dq is None or dq.device != q.device:
                dq = self.__dict__["_dq_scratch"] = torch.empty_like(q)
        out = torch.empty(1, dtype=torch.float32, device=q.device)
        ops.eb_aux(self.packed_params(), q.data, float(np.log(2 / self.tail_mass - 1)), dq, out, self.channels)
        return out

    def logits_cumulative(self, x, stop_gradient):
        for i in range(len(self.filters) + 1):
            m, b = getattr(self, f"_matrix{i:d}"), getattr(self, f"_bias{i:d}")
            if stop_gradient:
                m, b = m.detach(), b.detach()
            x = torch.matmul(F.softplus(m), x) + b
            if i < len(self.filters):
                fac = getattr(self, f"_factor{i:d}")
                if stop_gradient:
                    fac = fac.detach()
                x = x + torch.tanh(fac) * torch.tanh(x)
        return x

    def loss(self):
        """aux loss on the 3 quantiles per channel (parameter space, 128x3 values) — plain autograd"""
        logits = self.logits_cumulative(self.quantiles, stop_gradient=True)
        return torch.abs(logits - self.target).sum()


class GaussianConditional(_CdfBuffers, nn.Module):
    def __init__(self, scale_bound=0.11, tail_mass=1e-9, likelihood_bound=1e-9):
        super().__init__()
        self.tail_mass = float(tail_mass)
        self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        self.register_buffer("scale_table", torch.Tensor())
        self.lower_bound_scale = LowerBound(scale_bound)


# ------------------------------------------------------------------ the coder
class Cheng2020Anchor(nn.Module, PackCache):
    def __init__(self, N=192):
        super().__init__()
        self.N = self.M = M = N
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.h_a = nn.Sequential(
            conv3x3(N, N), nn.LeakyReLU(inplace=True), conv3x3(N, N), nn.LeakyReLU(inplace=True),
            conv3x3(N, N, stride=2), nn.LeakyReLU(inplace=True), conv3x3(N, N), nn.LeakyReLU(inplace=True),
            conv3x3(N, N, stride=2))
        self.h_s = nn.Sequential(
            conv3x3(N, N), nn.LeakyReLU(inplace=True), subpel_conv3x3(N, N, 2), nn.LeakyReLU(inplace=True),
            conv3x3(N, N * 3 // 2), nn.LeakyReLU(inplace=True), subpel_conv3x3(N * 3 // 2, N * 3 // 2, 2),
            nn.LeakyReLU(inplace=True), conv3x3(N * 3 // 2, N * 2))
        self.entropy_parameters = nn.Sequential(
            nn.Conv2d(M * 12 // 3, M * 10 // 3, 1), nn.LeakyReLU(inplace=True),
            nn.Conv2d(M * 10 // 3, M * 8 // 3, 1), nn.LeakyReLU(inplace=True),
            nn.Conv2d(M * 8 // 3, M * 6 // 3, 1))
        self.context_prediction = MaskedConv2d(M, 2 * M, kernel_size=5, padding=2, stride=1)
        self.gaussian_conditional = GaussianConditional()

    # -- transforms ------------------------------------------------------------------------
    def run_g_a(self, x: FM):
        """-> (y fp32 FM, y FM for h_a: fp16 by default, the fp32 y itself when x is fp32 = the fp32-island mode)"""
        g = self.g_a
        t = g[0].run(x)
        t = g[1].run(t)
        t = g[2].run(t)
        t = g[3].run(t)            # SE
        t = g[4].run(t)
        t = g[5].run(t)
        t = g[6].run(t)
        # y feeds round(): keep the last analysis conv + SE scaling in fp32 (an fp16 store here
        # costs ~0.2 % symbol flips against the fp32 reference)
        f32 = x.f32
        t = ops.conv(t, pk_conv(self, "ga7", g[7]), out_dtype=torch.float32)
        y32 = FM.empty(t.N, t.H, t.W, t.C, dtype=torch.float32, device=t.t.device)
        if f32:
            g[8].run(t, out=y32)
            return y32, y32
        y16 = FM.empty(t.N, t.H, t.W, t.C, device=t.t.device)
        g[8].run(t, out=y32, out2=y16)
        return y32, y16

    def run_g_s(self, y_hat: FM, out: FM | None = None, res: FM | None = None) -> FM:
        g = self.g_s
        t = g[0].run(y_hat)
        for i in range(1, 5):
            t = g[i].run(t)
        t = g[5].run(t)
        for i in range(6, 9):
            t = g[i].run(t)
        # x_hat re-enters an autocast region (pnet.py:51,75): fp16 whatever the coder's precision; the fused `+ res` is
        # prediction + recon_res of pnet.py:76 fused into the last layer's epilogue.  Maps of < 8192 px (generic epilogue) sum in
        # fp32 and round once; the 1080p path (conv_mfma_v11, lean epilogue) rounds the conv result to fp16 and adds the
        # prediction in packed fp16 -- two roundings, <= 1 fp16 ulp of a [0,1] pixel; the 1080p trained-point parity test
        # (tests/test_model_gpu.py::test_trained_operating_point_parity_1080p) runs through that kernel
        return ops.conv(t, pk_conv(self, "gs9", g[9][0], shuffle=True), out=out, res=res, out_dtype=torch.float16)

    def run_h_a(self, y16: FM) -> FM:
        h = self.h_a
        t = ops.conv(y16, pk_conv(self, "ha0", h[0]), **LR)
        t = ops.conv(t, pk_conv(self, "ha2", h[2]), **LR)
        t = ops.conv(t, pk_conv(self, "ha4", h[4]), **LR)
        t = ops.conv(t, pk_conv(self, "ha6", h[6]), **LR)
        return ops.conv(t, pk_conv(self, "ha8", h[8]), out_dtype=torch.float32)

    def run_h_s(self, z_hat: FM, out: FM) -> FM:
        h = self.h_s
        t = ops.conv(z_hat, pk_conv(self, "hs0", h[0]), **LR)
        t = ops.conv(t, pk_conv(self, "hs2", h[2][0], shuffle=True), **LR)
        t = ops.conv(t, pk_conv(self, "hs4", h[4]), **LR)
        t = ops.conv(t, pk_conv(self, "hs6", h[6][0], shuffle=True), **LR)
        return ops.conv(t, pk_conv(self, "hs8", h[8]), out=out)

    def run_entropy_parameters(self, pc_in: FM) -> FM:
        e = self.entropy_parameters
        t = ops.conv(pc_in, pk_conv(self, "ep0", e[0]), **LR)
        t = ops.conv(t, pk_conv(self, "ep2", e[2]), **LR)
        return ops.conv(t, pk_conv(self, "ep4", e[4]), out_dtype=torch.float32)

    def ctx_conv(self) -> ops.PackedConv:
        cp = self.context_prediction

        def build():
            pc = ops.pack_conv(cp.weight, cp.bias, stride=1, pad=2, taps=cp.live_taps(), device=cp.weight.device)
            # compressai zeroes the masked taps in the forward (`weight.data *= mask`) but autograd still fills their
            # gradient, and the reference clips the norm over ALL of it (tools/train.py:147)
            pc.orig["wgrad_taps"] = [(dy, dx) for dy in range(5) for dx in range(5)]
            return pc
        return self._pk("ctx", build)

    # -- forward (`main/model/pnet.py:34,58`) ------------------------------------------------
    def _as_f32(self, x: FM) -> FM:
        """`estmv.float()` / `input_residual.float()` of pnet.py:34,58: the fp32 island's input"""
        return x if x.f32 else ops.copy_cast(x, FM.empty(x.N, x.H, x.W, x.C, dtype=torch.float32, device=x.t.device))

    def run(self, x: FM, training: bool, out: FM | None = None, res: FM | None = None, trace=None, noise=None, f32=False):
        """x: (B,H,W,64) fp16.  Returns (x_hat FM [+res], bits tensor (2,) float64 = [y, z]).
        `f32`: run the coder as the reference's fp32 island (pnet.py:33-49: autocast off, `estmv.float()`): fp32
        activations and weights on the fp32 MFMA form of every conv; inference only."""
        dev = x.t.device
        if f32:
            if training:
                raise RuntimeError("the fp32-island mode has no backward: train with the default (fp16-in / fp32-accumulate) coders")
            x = self._as_f32(x)
        adt = torch.float32 if f32 else torch.float16          # activation dtype inside the coder
        y32, y16 = self.run_g_a(x)
        z = self.run_h_a(y16)
        B, h, w, M = y32.N, y32.H, y32.W, self.M
        bits = torch.zeros(2, dtype=torch.float64, device=dev)
        z_hat = FM.empty(z.N, z.H, z.W, z.C, dtype=adt, device=dev)
        nz = ny = nl = None
        if training:
            # three independent U(-1/2, 1/2) draws, as compressai: factorised prior, y_hat, Gaussian likelihood
            # (`noise`: test hook, dict of fp32 FMs {"z", "y", "y_lik"})
            noise = noise or {}
            nz = noise.get("z") or FM(torch.rand((z.N, z.H, z.W, z.C), device=dev) - 0.5)
            ny = noise.get("y") or FM(torch.rand((B, h, w, M), device=dev) - 0.5)
            nl = noise.get("y_lik") or FM(torch.rand((B, h, w, M), device=dev) - 0.5)
        ops.eb_forward(z, self.entropy_bottleneck.packed_params(), z_hat, bits[1:2], noise=nz)
        pcat = FM.empty(B, h, w, 4 * M, dtype=adt, device=dev)  # [h_s params | context]
        self.run_h_s(z_hat, out=pcat.ch(0, 2 * M))
        y_hat = ops.quantize(y32, FM.empty(B, h, w, M, dtype=adt, device=dev), noise=ny)
        ops.conv(y_hat, self.ctx_conv(), out=pcat.ch(2 * M, 2 * M))
        gp = self.run_entropy_parameters(pcat)
        ops.gc_forward(y32, gp, bits[0:1], noise=nl)
        x_hat = self.run_g_s(y_hat, out=out, res=res)
        if trace is not None:
            trace.update(y=y32, z=z, z_hat=z_hat, y_hat=y_hat, gp=gp)
        return x_hat, bits

    def aux_loss(self):
        """compressai `aux_loss()`: the torch expression with its autograd graph over the quantiles (`aux_loss.backward()` works as in
        tools/train.py:150, also on the value the training-mode forward returns).  Under a tape whose owner takes the gradient from
        `EntropyBottleneck.loss_fused` instead (`tape.fused_aux`, set by TrainStep) only the value is needed: one kernel."""
        eb = self.entropy_bottleneck
        if ops.TAPE is not None and getattr(ops.TAPE, "fused_aux", False) and eb.quantiles.is_cuda:
            return eb.loss_fused(write_grad=False).reshape(())
        return eb.loss()

    # -- entropy coding (`main/model/pnet.py:45-49,69-73`) -----------------------------------
    @torch.no_grad()
    def update(self, force=False):
        """compressai `update(force)`: build the quantised CDF tables of both entropy models.
        Parameter-space arithmetic (128 x ~60 and 64 x ~3000 values) on the host; CDF quantisation
        in the C++ coder library."""
        import scipy.stats
        dev = self.context_prediction.weight.device
        eb, gc = self.entropy_bottleneck, self.gaussian_conditional
        changed = False
        if gc._offset.numel() == 0 or force:
            table = torch.exp(torch.linspace(math.log(0.11), math.log(256), 64))
            mult = -scipy.stats.norm.ppf(gc.tail_mass / 2)
            center = torch.ceil(table * mult).int()
            length = 2 * center + 1
            maxlen = int(length.max())
            samples = torch.abs(torch.arange(maxlen).int() - center[:, None]).float()
            sc = table.unsqueeze(1).float()
            cdfn = lambda v: 0.5 * torch.erfc(-(2 ** -0.5) * v)
            upper, lower = cdfn((0.5 - samples) / sc), cdfn((-0.5 - samples) / sc)
            pmf, tail = upper - lower, 2 * lower[:, :1]
            cdf = torch.zeros((64, maxlen + 2), dtype=torch.int32)
            for i in range(64):
                prob = torch.cat((pmf[i, : int(length[i])], tail[i])).numpy()
                c = ops.pmf_to_quantized_cdf(prob)
                cdf[i, : c.size] = torch.from_numpy(c)
            gc._quantized_cdf, gc._offset, gc._cdf_length = cdf.to(dev), (-center).to(dev), (length + 2).to(dev)
            gc.scale_table = table.to(dev)
            changed = True
        if eb._offset.numel() == 0 or force:
            q = eb.quantiles.detach().float().cpu()
            med = q[:, 0, 1]
            minima = torch.clamp(torch.ceil(med - q[:, 0, 0]).int(), min=0)
            maxima = torch.clamp(torch.ceil(q[:, 0, 2] - med).int(), min=0)
            start, length = med - minima, maxima + minima + 1
            maxlen = int(length.max())
            samples = torch.arange(maxlen)[None, :] + start[:, None, None]
            ebc = EntropyBottleneck(eb.channels)          # CPU twin for the logits chain
            ebc.load_state_dict({k: v.cpu() for k, v in eb.state_dict().items()}, strict=False)
            lower = ebc.logits_cumulative(samples - 0.5, True)
            upper = ebc.logits_cumulative(samples + 0.5, True)
            sign = -torch.sign(lower + upper)
            pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
            tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
            cdf = torch.zeros((eb.channels, maxlen + 2), dtype=torch.int32)
            for i in range(eb.channels):
                prob = torch.cat((pmf[i, : int(length[i])], tail[i])).numpy()
                c = ops.pmf_to_quantized_cdf(prob)
                cdf[i, : c.size] = torch.from_numpy(c)
            eb._quantized_cdf, eb._offset, eb._cdf_length = cdf.to(dev), (-minima).to(dev), (length + 2).to(dev)
            changed = True
        self.__dict__.pop("_tables", None)
        return changed

    def _coder_tables(self):
        t = self.__dict__.get("_tables")
        if t is None:
            eb, gc = self.entropy_bottleneck, self.gaussian_conditional
            if eb._offset.numel() == 0 or gc._offset.numel() == 0:
                raise RuntimeError("entropy coder tables are empty: call update(force=True) first (pnet.py:47)")
            t = (ops.CdfTables(eb._quantized_cdf, eb._cdf_length, eb._offset),
                 ops.CdfTables(gc._quantized_cdf, gc._cdf_length, gc._offset), gc.scale_table.float().contiguous())
            self.__dict__["_tables"] = t
        return t

    def _ctx_1x1(self) -> ops.PackedConv:
        """the masked 5x5 context conv as a 1x1 conv over gathered 12-tap neighbourhoods"""
        cp = self.context_prediction

        def build():
            # (2M, 12*M, 1, 1), tap-major, gathered straight from the parameter (a view, not a snapshot: PackBatch /
            # refresh_packed re-pack it from the live weights after every optimizer step, like every other layer form)
            co, ci, kh, kw = cp.weight.shape
            chan = np.array([c * kh * kw + dy * kw + dx for dy, dx in cp.live_taps() for c in range(ci)], dtype=np.int64)
            lay = ops.convpack.WeightLayout(co, len(chan), 1, 1, np.arange(co, dtype=np.int64) * ci * kh * kw, chan, np.zeros(1, dtype=np.int64))
            return ops.pack_conv(cp.weight, cp.bias, stride=1, pad=0, device=cp.weight.device, layout=lay, param_w=cp.weight, param_b=cp.bias)
        return self._pk("ctx1x1", build)

    @staticmethod
    def wavefront_steps(H, W):
        """positions with equal w + 3h are mutually independent under the type-A 5x5 mask"""
        steps = []
        for t in range(W + 3 * (H - 1)):
            ps = [(h, t - 3 * h) for h in range(H) if 0 <= t - 3 * h < W]
            if ps:
                steps.append(ps)
        return steps

    def _ar_setup(self, H, W, adt, dev, B):
        """per (grid, dtype): the wavefront position list, the step sizes, the staging chain and pinned host buffers for the
        symbols / indexes of B images -- built once (the position list alone is 2 ms of Python at 68 x 120)"""
        cache = self.__dict__.setdefault("_ar_cache", {})
        key = (H, W, adt, str(dev), B)
        c = cache.get(key)
        if c is not None and not all(a is b for a, b in zip(c["chain"]["pcs"], self._ar_packed())):
            c = None                     # the packed weights were rebuilt since: the cached descriptors point at the old blobs
        if c is None:
            steps = self.wavefront_steps(H, W)
            flat = torch.tensor([p for st in steps for p in st], dtype=torch.int32, device=dev)
            c = dict(flat=flat, fl=flat.long(), sizes=np.array([len(st) for st in steps], dtype=np.int32), chain=self._ar_chain(H, adt, dev),
                     host=[(torch.empty((H * W, self.M), dtype=torch.int32).pin_memory(), torch.empty((H * W, self.M), dtype=torch.int32).pin_memory())
                           for _ in range(B)])
            cache.clear()                # one geometry at a time (the staging buffers are not small)
            cache[key] = c
        return c

    @torch.no_grad()
    def compress(self, x: FM, f32=False, order="raster", defer=False):
        """-> {"strings": [y_strings, z_strings], "shape": (h, w)} like compressai's compress()
        (`f32`: the fp32-island mode, see run()).

        `defer`: "strings" is a PendingStrings instead: the symbols leave the device asynchronously into pinned memory and the
        host range coder runs on a worker thread while the caller goes on enqueueing GPU work (VideoCompressor.encode: the
        coder's 5.5 ms per million symbols disappear from the frame's critical path); `result()` joins.  At most one deferred
        call per coder may be outstanding (its pinned buffers are re-used by the next call).

        `order`: symbol order of the y stream.  "raster" is compressai's (h, w, c) order: the stream the reference writes, decoded
        position by position.  "wavefront" emits the same symbols anti-diagonal by anti-diagonal (the order the encoder computes
        them in, wavefront_steps()), which lets decompress() decode a whole diagonal per step: an extension, not readable by the
        reference's decoder."""
        if order not in STREAM_ORDERS:
            raise ValueError(f"order must be one of {STREAM_ORDERS}, got {order!r}")
        dev = x.t.device
        ebt, gct, table = self._coder_tables()
        M = self.M
        if f32:
            x = self._as_f32(x)
        adt = torch.float32 if f32 else torch.float16
        y32, y16 = self.run_g_a(x)
        z = self.run_h_a(y16)
        B, H, W = y32.N, y32.H, y32.W
        med = self.entropy_bottleneck.quantiles.detach()[:, 0, 1].float().contiguous()
        zsym = ops.round_symbols(z, med)                                           # (B, h, w, C) int32
        z_hat = FM((zsym.float() + med).to(adt))
        params = FM.empty(B, H, W, 2 * M, dtype=adt, device=dev)
        self.run_h_s(z_hat, out=params)
        zs = zsym.permute(0, 3, 1, 2).contiguous().cpu().numpy()                   # compressai order (C, h, w)
        zidx = np.broadcast_to(np.arange(M, dtype=np.int32)[:, None, None], zs.shape[1:])
        z_strings = [ops.rans_encode(zs[b], zidx, ebt) for b in range(B)]
        st = self._ar_setup(H, W, adt, dev, B)
        flat, sizes, chain, fl = st["flat"], st["sizes"], st["chain"], st["fl"]
        y_jobs, dbg = [], []
        for b in range(B):
            # every position is written exactly once and only causal (already written) neighbours are read: no zero fill
            y_hat = FM.empty(1, H, W, M, dtype=adt, device=dev)
            sym = torch.empty((H, W, M), dtype=torch.int32, device=dev)
            idx = torch.empty((H, W, M), dtype=torch.int32, device=dev)
            # the W + 3(H-1) steps run natively (tdvc_ar_wavefront: gather -> context conv -> entropy_parameters -> quantise)
            ops.ar_wavefront(None, None, y32.batch(b, 1), y_hat, params.batch(b, 1), chain["x1"], chain["pc"], chain["descs"], chain["gp"],
                             flat, sizes, M, W, table, idx, sym)
            hs, hi = st["host"][b]
            if order == "wavefront":
                hs.copy_(sym[fl[:, 0], fl[:, 1]], non_blocking=True)
                hi.copy_(idx[fl[:, 0], fl[:, 1]], non_blocking=True)
            else:
                hs.copy_(sym.view(H * W, M), non_blocking=True)                # raster (h, w, c) order
                hi.copy_(idx.view(H * W, M), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()

            def job(ev=ev, hs=hs, hi=hi):
                ev.synchronize()
                return ops.rans_encode(hs.numpy(), hi.numpy(), gct)
            y_jobs.append(_rans_pool().submit(job) if defer else job())
            dbg.append({"y_hat": y_hat, "symbols": sym, "indexes": idx})
        if defer:
            return {"strings": PendingStrings(y_jobs, z_strings), "shape": (z.H, z.W), "_debug": dbg}
        return {"strings": [y_jobs, z_strings], "shape": (z.H, z.W), "_debug": dbg}

    @torch.no_grad()
    def decompress(self, strings, shape, synth=True, f32=False, order="raster"):
        """strings as returned by compress(); "raster": serial position-by-position context decoding (the stream order of
        compressai's bitstream); "wavefront": one anti-diagonal per step (W + 3(H-1) steps instead of H*W).
        -> {"x_hat": FM, "y_hat": FM}.  `f32` and `order` must match the encoder's."""
        if order not in STREAM_ORDERS:
            raise ValueError(f"order must be one of {STREAM_ORDERS}, got {order!r}")
        dev = self.context_prediction.weight.device
        adt = torch.float32 if f32 else torch.float16
        ebt, gct, table = self._coder_tables()
        M = self.M
        zh, zw = shape
        B = len(strings[1])
        H, W = zh * 4, zw * 4
        med = self.entropy_bottleneck.quantiles.detach()[:, 0, 1].float().contiguous()
        zidx = np.broadcast_to(np.arange(M, dtype=np.int32)[:, None, None], (M, zh, zw))
        zs = np.stack([ops.RansDecoder(s).decode(zidx, ebt).reshape(M, zh, zw) for s in strings[1]])
        zsym = torch.from_numpy(zs).to(dev).permute(0, 2, 3, 1).contiguous()
        z_hat = FM((zsym.float() + med).to(adt))
        params = FM.empty(B, H, W, 2 * M, dtype=adt, device=dev)
        self.run_h_s(z_hat, out=params)
        y_hat_all = FM.zeros(B, H, W, M, dtype=adt, device=dev)
        sym = torch.zeros((H, W, M), dtype=torch.int32, device=dev)
        idx = torch.zeros((H, W, M), dtype=torch.int32, device=dev)
        if order == "wavefront":
            for b in range(B):
                self._decode_wavefront(strings[0][b], gct, table, y_hat_all.batch(b, 1), params.batch(b, 1))
            return {"x_hat": self.run_g_s(y_hat_all) if synth else None, "y_hat": y_hat_all}
        # the per-position chain over one-position staging buffers; the loop itself runs natively
        # (tdvc_ar_decode_serial: Python drove it at ~230 us per position)
        chain = self._ar_chain(1, adt, dev)
        pos_table = torch.tensor([[h, w] for h in range(H) for w in range(W)], dtype=torch.int32, device=dev)
        for b in range(B):
            ops.ar_decode_serial(strings[0][b], gct, y_hat_all.batch(b, 1), params.batch(b, 1), chain["x1"], chain["pc"], chain["descs"],
                                 chain["gp"], pos_table, M, W, table, idx, sym)
        return {"x_hat": self.run_g_s(y_hat_all) if synth else None, "y_hat": y_hat_all}

    def _ar_chain(self, cap, adt, dev):
        """the per-step chain as conv descriptors over staging buffers of `cap` positions: context conv (1x1 over the gathered
        neighbourhoods) into pc[2M:4M], entropy_parameters into gp"""
        M, e = self.M, self.entropy_parameters
        x1 = FM.empty(1, 1, cap, 12 * M, dtype=adt, device=dev)
        pc = FM.empty(1, 1, cap, 4 * M, dtype=adt, device=dev)
        t0 = FM.empty(1, 1, cap, ops.pad8(e[0].out_channels), dtype=adt, device=dev)
        t1 = FM.empty(1, 1, cap, ops.pad8(e[2].out_channels), dtype=adt, device=dev)
        gp = FM.empty(1, 1, cap, 2 * M, dtype=torch.float32, device=dev)
        v = lambda fm, c0, C_: FM(fm.t, c0, 1, C_)
        pcs = self._ar_packed()
        descs = [ops.conv_desc(v(x1, 0, x1.C), pcs[0], out=v(pc, 2 * M, 2 * M))[0],
                 ops.conv_desc(v(pc, 0, 4 * M), pcs[1], out=t0, **LR)[0],
                 ops.conv_desc(t0, pcs[2], out=t1, **LR)[0],
                 ops.conv_desc(t1, pcs[3], out=gp, out_dtype=torch.float32)[0]]
        # the descriptors hold raw device pointers into the packed weights (and their fp32 twins): the chain keeps the packed objects
        # alive, and _ar_setup() rebuilds it when the coder's packed forms are no longer these objects (clear_packed, refresh, .to())
        return {"x1": x1, "pc": pc, "gp": gp, "descs": descs, "keep": (t0, t1), "pcs": pcs}

    def _ar_packed(self):
        e = self.entropy_parameters
        return [self._ctx_1x1(), pk_conv(self, "ep0", e[0]), pk_conv(self, "ep2", e[2]), pk_conv(self, "ep4", e[4])]

    def _decode_wavefront(self, data, gct, table, y_hat, params):
        """one image of a wavefront-ordered y stream (tdvc_ar_wavefront, decoder direction)"""
        dev, M = y_hat.t.device, self.M
        H, W = y_hat.H, y_hat.W
        steps = self.wavefront_steps(H, W)
        flat = torch.tensor([p for st in steps for p in st], dtype=torch.int32, device=dev)
        chain = self._ar_chain(H, y_hat.t.dtype, dev)
        sym = torch.zeros((H * W, M), dtype=torch.int32, device=dev)          # wavefront order
        idx = torch.zeros((H * W, M), dtype=torch.int32, device=dev)
        ops.ar_wavefront(data, gct, None, y_hat, params, chain["x1"], chain["pc"], chain["descs"], chain["gp"], flat,
                         np.array([len(st) for st in steps], dtype=np.int32), M, W, table, idx, sym)

def _g_a(N):
    return nn.Sequential(
        ResidualBlockWithStride(64, N, stride=2), ResidualBlock(N, N),
        ResidualBlockWithStride(N, N, stride=2), SELayer(N), ResidualBlock(N, N),
        ResidualBlockWithStride(N, N, stride=2), ResidualBlock(N, N),
        conv3x3(N, N, stride=2), SELayer(N))


def _g_s(N):
    return nn.Sequential(
        SELayer(N), ResidualBlock(N, N), ResidualBlockUpsample(N, N, 2), ResidualBlock(N, N),
        ResidualBlockUpsample(N, N, 2), SELayer(N), ResidualBlock(N, N),
        ResidualBlockUpsample(N, N, 2), ResidualBlock(N, N), subpel_conv3x3(N, 64, 2))


class ResCoder(Cheng2020Anchor):
    """`main/model/encoder_v3.py:14-40`"""

    def __init__(self, N=192):
        super().__init__(N=N)
        self.g_a, self.g_s = _g_a(N), _g_s(N)


class MVCoder(Cheng2020Anchor):
    """`main/model/encoder_v3.py:43-69`"""

    def __init__(self, N=192):
        super().__init__(N=N)
        self.g_a, self.g_s = _g_a(N), _g_s(N)
