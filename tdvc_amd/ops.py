"""Host-side operator layer: torch tensors (device memory + streams only) -> C-ABI calls.

`FM` is a channel-innermost feature-map view ([N][H][W][C], fp16 or fp32) over a torch buffer;
channel slices of a wider buffer are views, so concatenations are never materialised.
Every function enqueues on torch's current HIP stream and returns immediately.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib as L
from . import convpack
from ._lib import (ACT_CLAMP01, ACT_LRELU, ACT_NONE, ACT_RELU, GDN_FWD, GDN_INV, GDN_NONE, OUT_NCHW_F32,  # noqa: F401
                   OUT_NHWC, OUT_SHUFFLE2)


# The launch stream.  `torch.cuda.current_stream()` costs ~8 us of Python per call (device-index and availability checks, an os.environ
# lookup, a Stream object): with ~2.4 k launches per training step it was a fifth of the HOST time that bounds the step
# (tools/train_host_bound.py, tools/train_host_profile.py); the raw handle comes from the same C++ call without the wrappers.
_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_dev = torch._C._cuda_getDevice
FORCE_STREAM = None       # a raw stream handle: every launch goes there instead of torch's current stream (Tape.off_path: the side
                          # stream without the ~10 us of a `with torch.cuda.stream(...)` block per weight-gradient launch)
FORCE_KEEP = None         # with FORCE_STREAM: a list that keeps temporaries of such launches alive (they were allocated by the main
                          # stream's pool, which must not reuse them before the streams have joined)


def _stream():
    if FORCE_STREAM is not None:
        return C.c_void_p(FORCE_STREAM)
    return C.c_void_p(_raw_stream(_cur_dev()))


# optional per-launch event trace (bench.py roofline leg): list of dicts, or None
PROFILE = None
# active autograd tape (tdvc_amd/autograd.py) or None: differentiable ops record their backward closure on it
TAPE = None
_IN_BACKWARD = False      # set by Tape.backward(): ops called from backward closures are not recorded


def _rec(name, *args):
    """record a differentiable op on the active tape (no-op outside `autograd.record()`)"""
    if TAPE is not None and not _IN_BACKWARD:
        from . import autograd
        getattr(autograd, "record_" + name)(TAPE, *args)


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


class FM:
    """Feature-map view over a torch buffer `t` of shape (N, H, W, Cbuf): N images of C channels
    starting `off` elements into the buffer, batch stride `sn`, pixel stride t.stride(2)."""

    __slots__ = ("t", "off", "N", "C", "H", "W", "sn", "sp", "_d")

    def __init__(self, t: torch.Tensor, off: int = 0, N: int | None = None, C_: int | None = None, sn: int | None = None):
        assert t.dim() == 4 and (t.shape[3] == 1 or t.stride(3) == 1)
        self.t, self.off = t, off
        self._d = None
        self.N = t.shape[0] if N is None else N
        self.C = t.shape[3] if C_ is None else C_
        self.H, self.W = t.shape[1], t.shape[2]
        # strides of size-1 dimensions are arbitrary in torch: derive them from the dense layout instead
        if self.W > 1:
            self.sp = t.stride(2)
        elif self.H > 1:
            self.sp = t.stride(1)
        else:
            self.sp = t.shape[3]
        assert self.H == 1 or t.stride(1) == self.W * self.sp, "rows of a feature map must be dense (row stride = W * pixel stride)"
        self.sn = (t.stride(0) if t.shape[0] > 1 else self.H * self.W * self.sp) if sn is None else sn

    @staticmethod
    def empty(N, H, W, C_, dtype=torch.float16, device="cuda"):
        return FM(torch.empty((N, H, W, C_), dtype=dtype, device=device))

    @staticmethod
    def zeros(N, H, W, C_, dtype=torch.float16, device="cuda"):
        return FM(torch.zeros((N, H, W, C_), dtype=dtype, device=device))

    @property
    def f32(self):
        return self.t.dtype == torch.float32

    def ch(self, c0, C_):
        """channel slice view"""
        assert 0 <= c0 and c0 + C_ <= self.C
        return FM(self.t, self.off + c0, self.N, C_, self.sn)

    def batch(self, n0, N):
        assert 0 <= n0 and n0 + N <= self.N
        return FM(self.t, self.off + n0 * self.sn, N, self.C, self.sn)

    def as_slices(self, b, T, Cs):
        """the T channel-slices (width Cs) of batch item b, viewed as a batch of T images"""
        assert self.C == T * Cs and 0 <= b < self.N
        return FM(self.t, self.off + b * self.sn, T, Cs, Cs)

    def desc(self) -> L.FMapDesc:
        d = self._d                          # an FM is immutable: built once (struct fields are copied on assignment into descriptors)
        if d is None:
            p = self.t.data_ptr() + self.off * self.t.element_size()
            d = self._d = L.FMapDesc(p, self.N, self.H, self.W, self.C, self.sn, self.sp, L.F32 if self.f32 else L.F16)
        return d

    def to_nchw(self, C_=None) -> torch.Tensor:
        C_ = self.C if C_ is None else C_
        out = torch.empty((self.N, C_, self.H, self.W), dtype=torch.float32, device=self.t.device)
        d = self.desc()
        L.check(L.lib().tdvc_fmap_to_nchw(C.byref(d), C_, out.data_ptr(), _stream()), "fmap_to_nchw")
        return out


_NULL_FM = L.FMapDesc(None, 0, 0, 0, 0, 0, 0, 0)


def from_nchw(x: torch.Tensor, Cpad: int | None = None, dtype=torch.float16, out: FM | None = None) -> FM:
    """(N,C,H,W) fp32 cuda tensor -> FM (channels zero-padded to Cpad)."""
    assert x.is_cuda and x.dtype == torch.float32
    x = x.contiguous()
    N, Cc, H, W = x.shape
    if out is None:
        out = FM.empty(N, H, W, pad8(Cc) if Cpad is None else Cpad, dtype=dtype, device=x.device)
    d = out.desc()
    L.check(L.lib().tdvc_nchw_to_fmap(x.data_ptr(), Cc, C.byref(d), _stream()), "nchw_to_fmap")
    return out


# ----------------------------------------------------------------------------- conv
@dataclass
class PackedConv:
    w: torch.Tensor          # device uint8 blob (fragment-ordered fp16)
    bias: torch.Tensor       # device fp32 [cout_pad]
    cout: int
    cin: int                 # padded input channels the kernel will see
    kh: int
    kw: int
    stride: int
    pad: int
    ck: int
    taps: list               # [(dy, dx)]
    shuffle: bool = False
    cin_real: int = 0        # un-padded input channels (algorithmic FLOP accounting)
    s2d: bool = False        # stride-2 3x3 conv executed as a 2x2 conv over a space-to-depth view
    flops_per_px: float = 0.0  # algorithmic FLOP per OUTPUT pixel of the ORIGINAL conv
    # --- provenance for re-packing after an optimizer step and for the backward pass (training path)
    wsrc: torch.Tensor | None = None       # the fp32 weight (nn.Parameter or tensor) on the device
    bsrc: torch.Tensor | None = None
    layout: convpack.WeightLayout | None = None
    tables: convpack.PackTables | None = None
    orig: dict | None = None               # geometry of the ORIGINAL conv: stride, pad, taps, cin_perm, kh, kw
    dgrad: "PackedConv | None" = None      # lazily built data-gradient conv (ops.conv_dgrad)
    param_w: torch.Tensor | None = None    # the nn.Parameters that own wsrc / bsrc (gradient accumulators live on them)
    param_b: torch.Tensor | None = None
    owner: object | None = None            # module that maps this layer's weight gradient to its own parameters (GDN)
    w32: torch.Tensor | None = None        # fp32 twin of `w` for the fp32 islands (conv_f32.hip), packed on first use

    def packed_f32(self) -> torch.Tensor:
        """the same fragment order as `w`, as floats (tdvc_pack_conv_weights_indexed_f32)"""
        if self.w32 is None:
            self.w32 = _pack_from_tables_f32(self.wsrc, self.tables)
        return self.w32

    def repack(self):
        """re-pack from the (updated) fp32 parameters: one kernel launch, plus the bias gather"""
        _pack_from_tables(self.wsrc, self.tables, self.w)
        self.w32 = None
        if self.bsrc is not None:
            b = self.bsrc.detach().float()
            self.bias[:self.cout] = b[torch.from_numpy(convpack.shuffle_perm(self.cout)).to(b.device)] if self.shuffle else b
        if self.dgrad is not None:
            self.dgrad.repack()
        for key in ("_colpc", "_plain"):           # the DCN weight as a 1x1 conv over sampled columns; the plain stride-2 form
            aux = self.__dict__.get(key)
            if aux is not None:
                aux.repack()


def _cout_pad(cout):
    return 32 if cout <= 32 else 64 * ((cout + 63) // 64)


def _s2d_weights(w: torch.Tensor) -> torch.Tensor:
    """(cout, C, 3, 3) stride-2 pad-1 kernel -> (cout, 4C, 2, 2) stride-1 kernel over the 2x2
    space-to-depth view: virtual channel q*C + c = parity (py, px) = (q>>1, q&1); virtual tap (dy, dx)
    covers original kernel row ky = 2*dy + py - 1 (zero weight when outside 0..2).  Reference form of
    `convpack.forward_tables_s2d` (tests)."""
    cout, C_, kh, kw = w.shape
    assert (kh, kw) == (3, 3)
    w2 = torch.zeros(cout, 4 * C_, 2, 2)
    for py in range(2):
        for px in range(2):
            q = py * 2 + px
            for dy in range(2):
                for dx in range(2):
                    ky, kx = 2 * dy + py - 1, 2 * dx + px - 1
                    if 0 <= ky <= 2 and 0 <= kx <= 2:
                        w2[:, q * C_:(q + 1) * C_, dy, dx] = w[:, :, ky, kx]
    return w2


def _pack_from_tables(wsrc: torch.Tensor, tb: convpack.PackTables, dst: torch.Tensor | None = None) -> torch.Tensor:
    lib = L.lib()
    nbytes = lib.tdvc_conv_packed_bytes(tb.cout, tb.cin, len(tb.taps), tb.ck)
    assert nbytes > 0
    w = wsrc.detach()
    assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
    if dst is None:
        dst = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    assert dst.numel() == nbytes
    L.check(lib.tdvc_pack_conv_weights_indexed(w.data_ptr(), tb.row_off.data_ptr(), tb.chan_off.data_ptr(), tb.tap_off.data_ptr(),
                                               tb.row_mask.data_ptr(), tb.chan_mask.data_ptr(), tb.tap_mask.data_ptr(),
                                               tb.cout, tb.cin, len(tb.taps), tb.ck, dst.data_ptr(), _stream()), "pack_conv_weights_indexed")
    return dst


def _pack_from_tables_f32(wsrc: torch.Tensor, tb: convpack.PackTables) -> torch.Tensor:
    lib = L.lib()
    nbytes = lib.tdvc_conv_packed_bytes(tb.cout, tb.cin, len(tb.taps), tb.ck)
    assert nbytes > 0
    w = wsrc.detach()
    assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
    dst = torch.empty(nbytes // 2, dtype=torch.float32, device=w.device)      # 8 floats where the fp16 blob has 8 halves
    L.check(lib.tdvc_pack_conv_weights_indexed_f32(w.data_ptr(), tb.row_off.data_ptr(), tb.chan_off.data_ptr(), tb.tap_off.data_ptr(),
                                                   tb.row_mask.data_ptr(), tb.chan_mask.data_ptr(), tb.tap_mask.data_ptr(),
                                                   tb.cout, tb.cin, len(tb.taps), tb.ck, dst.data_ptr(), _stream()), "pack_conv_weights_indexed_f32")
    return dst


class _PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("row_off", C.c_void_p), ("chan_off", C.c_void_p), ("tap_off", C.c_void_p),
                ("row_mask", C.c_void_p), ("chan_mask", C.c_void_p), ("tap_mask", C.c_void_p), ("dst", C.c_void_p),
                ("bias_src", C.c_void_p), ("bias_dst", C.c_void_p), ("bias_perm", C.c_void_p),
                ("cout", C.c_int32), ("cin", C.c_int32), ("ntaps", C.c_int32), ("ck", C.c_int32)]


class PackBatch:
    """Every PackedConv of a model (its forward, dgrad and column forms) re-packed from the fp32 parameters by ONE
    launch (`tdvc_pack_conv_weights_batch`): after an optimizer step ~400 small packing launches and ~200 bias copies
    become one kernel.  The job table lives on the device and is rebuilt when a tensor it points at has moved."""

    def __init__(self, pcs):
        self.pcs = []
        seen = set()

        def walk(pc):
            if pc is None or id(pc) in seen:
                return
            seen.add(id(pc))
            self.pcs.append(pc)
            walk(pc.dgrad)
            walk(pc.__dict__.get("_colpc"))
            walk(pc.__dict__.get("_plain"))
        for pc in pcs:
            walk(pc)
        self._build()

    def _tensors(self, pc):
        tb = pc.tables
        return [pc.wsrc, tb.row_off, tb.chan_off, tb.tap_off, tb.row_mask, tb.chan_mask, tb.tap_mask, pc.w] + \
               ([pc.bsrc, pc.bias] if pc.bsrc is not None else [])

    def _build(self):
        lib = L.lib()
        jobs = (_PackJob * len(self.pcs))()
        starts = [0]
        self._keep = []
        self._ptrs = []
        dev = self.pcs[0].w.device
        for j, pc in zip(jobs, self.pcs):
            tb = pc.tables
            w = pc.wsrc.detach()
            assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
            j.w, j.row_off, j.chan_off, j.tap_off = w.data_ptr(), tb.row_off.data_ptr(), tb.chan_off.data_ptr(), tb.tap_off.data_ptr()
            j.row_mask, j.chan_mask, j.tap_mask = tb.row_mask.data_ptr(), tb.chan_mask.data_ptr(), tb.tap_mask.data_ptr()
            j.dst = pc.w.data_ptr()
            j.cout, j.cin, j.ntaps, j.ck = tb.cout, tb.cin, len(tb.taps), tb.ck
            if pc.bsrc is not None:
                b = pc.bsrc.detach()
                assert b.is_cuda and b.dtype == torch.float32 and b.is_contiguous() and pc.bias.dtype == torch.float32
                j.bias_src, j.bias_dst = b.data_ptr(), pc.bias.data_ptr()
                if pc.shuffle:
                    perm = torch.from_numpy(convpack.shuffle_perm(pc.cout).astype(np.int32)).to(dev)
                    self._keep.append(perm)
                    j.bias_perm = perm.data_ptr()
            nb = lib.tdvc_pack_job_blocks(tb.cout, tb.cin, len(tb.taps), tb.ck)
            assert nb > 0
            starts.append(starts[-1] + nb)
            self._ptrs.append([t.data_ptr() for t in self._tensors(pc)])
        self.total_blocks = starts[-1]
        self.jobs = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        self.starts = torch.tensor(starts, dtype=torch.int32, device=dev)

    def run(self):
        # a moved tensor (module.to(), a re-assigned parameter) invalidates the device job table.  The full check reads ~10 pointers of ~470
        # layers (4 k `data_ptr()` calls: 2-3 ms of a host-bound step); the parameter and the blob of every layer each time, everything
        # every 64th run
        self._runs = getattr(self, "_runs", 0) + 1
        full = (self._runs & 63) == 1
        for pc, ptrs in zip(self.pcs, self._ptrs):
            if (([t.data_ptr() for t in self._tensors(pc)] != ptrs) if full else (pc.wsrc.data_ptr() != ptrs[0] or pc.w.data_ptr() != ptrs[7])):
                self._build()
                break
        L.check(L.lib().tdvc_pack_conv_weights_batch(self.jobs.data_ptr(), self.starts.data_ptr(), len(self.pcs), self.total_blocks, _stream()),
                "pack_conv_weights_batch")
        for pc in self.pcs:
            pc.w32 = None                # fp32 twins (fp32 islands, inference only) are rebuilt from the parameters on next use


def _pick_ck(cin, cout, kh, kw, stride, pad):
    if (stride == 1 or (stride == 2 and kh == 1 and kw == 1 and pad == 0)) and cin >= 32 and cout >= 64:
        return 32            # the weight-stationary / pipelined kernels stream 32-channel chunks
    ck = L.lib().tdvc_conv_plan(cin, kh, kw, stride)
    L.check(0 if ck > 0 else ck, "conv_plan")
    return ck


def pack_conv(weight: torch.Tensor, bias: torch.Tensor | None, stride=1, pad=0, cin_pad: int | None = None,
              taps=None, shuffle=False, cin_perm=None, device="cuda", ck=None, allow_s2d=True,
              layout: convpack.WeightLayout | None = None, param_w: torch.Tensor | None = None,
              param_b: torch.Tensor | None = None) -> PackedConv:
    """weight (cout, cin, kh, kw) fp32 (moved to `device` if it is not there; an nn.Parameter on the device is
    referenced, not copied, so `PackedConv.repack()` sees optimizer updates).  `taps`: list of (dy,dx) to keep
    (masked convs); `shuffle`: rows permuted for the PixelShuffle(2) store; `cin_perm`: index list applied to
    input channels (free re-ordering of concatenated inputs); `layout`: how the logical weight sits in the
    parameter's storage when it is not the dense (cout, cin, kh, kw) order (Conv3d holders)."""
    wsrc = weight if (weight.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous()) else \
        weight.detach().float().contiguous().to(device)
    dev = wsrc.device
    if layout is None:
        cout, cin_real, kh, kw = weight.shape
        layout = convpack.WeightLayout.dense(cout, cin_real, kh, kw)
    cout, kh, kw = layout.cout, layout.kh, layout.kw
    cin_real = layout.cin if cin_perm is None else len(cin_perm)
    bsrc = None if bias is None else (bias if bias.is_cuda else bias.detach().float().to(dev))
    orig = dict(stride=stride, pad=pad, taps=taps, cin_perm=cin_perm, kh=kh, kw=kw)
    s2d = (allow_s2d and stride == 2 and (kh, kw) == (3, 3) and pad == 1 and cin_real % 32 == 0 and cout >= 64
           and taps is None and not shuffle and ck is None and cin_perm is None and cin_pad is None)
    if shuffle:
        assert cout % 4 == 0
    if s2d:
        tb = convpack.forward_tables_s2d(layout, ck=32, device=dev)
        cin, k_stride, k_pad = cin_real, 1, 1
    else:
        cin = pad8(cin_real) if cin_pad is None else cin_pad
        if taps is None:
            taps = [(dy, dx) for dy in range(kh) for dx in range(kw)]
        if ck is None:
            ck = _pick_ck(cin, cout, kh, kw, stride, pad)
        tb = convpack.forward_tables(layout, cin_pad=cin, taps=taps, pad=pad, ck=ck, shuffle=shuffle, cin_perm=cin_perm, device=dev)
        k_stride, k_pad = stride, pad
    orig["taps"] = [(dy, dx) for dy in range(kh) for dx in range(kw)] if orig["taps"] is None else list(orig["taps"])
    bp = torch.zeros(_cout_pad(cout), dtype=torch.float32, device=dev)
    pc = PackedConv(_pack_from_tables(wsrc, tb), bp, cout, cin, tb.kh, tb.kw, k_stride, k_pad, tb.ck, tb.taps, shuffle, cin_real,
                    s2d=s2d, flops_per_px=2.0 * cout * cin_real * 9 if s2d else 0.0,
                    wsrc=wsrc, bsrc=bsrc, layout=layout, tables=tb, orig=orig,
                    param_w=param_w if param_w is not None else (weight if isinstance(weight, torch.nn.Parameter) else None),
                    param_b=param_b if param_b is not None else (bias if isinstance(bias, torch.nn.Parameter) else None))
    if bsrc is not None:
        b = bsrc.detach().float()
        bp[:cout] = b[torch.from_numpy(convpack.shuffle_perm(cout)).to(dev)] if shuffle else b
    return pc


SMALL_MAP_PIXELS = 8192      # conv_mfma_v9's range (csrc/conv_mfma_v9.hip)


def _plain_form(pc: PackedConv) -> PackedConv:
    """the un-transformed stride-2 3x3 packing of a layer whose main form is the space-to-depth view (built on first use)"""
    alt = pc.__dict__.get("_plain")
    if alt is None:
        alt = pack_conv(pc.wsrc, pc.bsrc, stride=2, pad=1, allow_s2d=False, layout=pc.layout, param_w=pc.param_w, param_b=pc.param_b)
        pc.__dict__["_plain"] = alt
    return alt


def conv_desc(x: FM, pc: PackedConv, out: FM | None = None, act=ACT_NONE, slope=0.0, res: FM | None = None,
              res2: FM | None = None, gdn=GDN_NONE, aux: FM | None = None, square=False, out_dtype=None,
              round16=False, nchw_out: torch.Tensor | None = None, bcast_T=0, bcast_slope=0.0):
    """the `tdvc_conv_desc` a conv() call would launch, without launching it (native loops re-launch fixed descriptors:
    `tdvc_ar_decode_serial`).  -> (descriptor, result FM / tensor, Ho, Wo, layer form used, layer as recorded)

    An fp32 input FM selects the fp32 form of the layer (fp32 weights, v_mfma_f32_32x32x2_f32: the fp32 islands of
    pnet.py:33,57); `out_dtype=None` allocates the output in the input's dtype."""
    assert x.C == pc.cin, f"conv: input has {x.C} channels, layer packed for {pc.cin}"
    rec_pc = pc                                   # the tape records the layer itself (its dgrad / wgrad forms hang off it)
    if out_dtype is None:
        out_dtype = torch.float32 if x.f32 else torch.float16
    if pc.s2d and (x.f32 or x.N * (x.H // 2) * (x.W // 2) <= SMALL_MAP_PIXELS):
        pc = _plain_form(pc)                      # small maps: the plain stride-2 form runs on the split-K kernel (fp32: always plain)
    if pc.s2d:
        if x.H % 2 or x.W % 2:
            raise L.TdvcHipError("conv: the space-to-depth stride-2 path needs even H, W")
        Ho, Wo = x.H // 2, x.W // 2
    else:
        Ho = (x.H + 2 * pc.pad - pc.kh) // pc.stride + 1
        Wo = (x.W + 2 * pc.pad - pc.kw) // pc.stride + 1
    # the layer's part of the descriptor (weights, bias, window, taps) is built once per layer and weight form and copied per call
    # (re-packing writes the same buffers in place; a re-allocated blob is caught by the pointer check)
    wt = pc.packed_f32() if x.f32 else pc.w
    key = "_tmpl32" if x.f32 else "_tmpl16"
    tm = pc.__dict__.get(key)
    if tm is None or tm[1] != wt.data_ptr():
        t_ = L.ConvDesc()
        t_.w = wt.data_ptr()
        t_.bias = pc.bias.data_ptr()
        t_.cout, t_.ntaps = pc.cout, len(pc.taps)
        for i, (dy, dx) in enumerate(pc.taps):
            t_.tap_dy[i], t_.tap_dx[i] = dy, dx
        t_.kh, t_.kw, t_.stride, t_.pad, t_.ck = pc.kh, pc.kw, pc.stride, pc.pad, pc.ck
        t_.s2d = int(pc.s2d)
        tm = pc.__dict__[key] = (t_, wt.data_ptr())
    d = L.ConvDesc.from_buffer_copy(tm[0])
    d.x = x.desc()
    d.square_input, d.gdn = int(square), gdn
    d.aux = aux.desc() if aux is not None else _NULL_FM
    d.act, d.slope, d.round_before_act = act, slope, int(round16)
    d.res = res.desc() if res is not None else _NULL_FM
    d.res2 = res2.desc() if res2 is not None else _NULL_FM
    d.bcast_T, d.bcast_slope = int(bcast_T), float(bcast_slope)
    if nchw_out is not None:
        assert nchw_out.shape == (x.N, pc.cout, Ho, Wo) and nchw_out.dtype == torch.float32 and nchw_out.is_contiguous()
        d.out_mode = OUT_NCHW_F32
        d.y = L.FMapDesc(nchw_out.data_ptr(), x.N, Ho, Wo, pc.cout, pc.cout * Ho * Wo, 1, L.F32)
        ret = nchw_out
    else:
        if pc.shuffle:
            d.out_mode = OUT_SHUFFLE2
            if out is None:
                out = FM.empty(x.N, 2 * Ho, 2 * Wo, pad8(pc.cout // 4), dtype=out_dtype, device=x.t.device)
        else:
            d.out_mode = OUT_NHWC
            if out is None:
                out = FM.empty(x.N, Ho, Wo, pc.cout if (out_dtype == torch.float32 and not x.f32) else pad8(pc.cout),
                               dtype=out_dtype, device=x.t.device)
        d.y = out.desc()
        ret = out
    return d, ret, Ho, Wo, pc, rec_pc


def conv(x: FM, pc: PackedConv, out: FM | None = None, act=ACT_NONE, slope=0.0, res: FM | None = None,
         res2: FM | None = None, gdn=GDN_NONE, aux: FM | None = None, square=False, out_dtype=None,
         round16=False, nchw_out: torch.Tensor | None = None, bcast_T=0, bcast_slope=0.0, chan_sum: list | None = None) -> FM | torch.Tensor:
    """`bcast_T` (inference only): the conv result is not stored but broadcast-added, with LeakyReLU(bcast_slope), over the
    bcast_T channel slices that start at `out` (tdvc_conv_desc::bcast_T).
    `chan_sum` (inference only): a list; when the kernel this conv dispatches to can sum the values it stores per channel
    (tdvc_conv_desc::chan_sum), (partial [N][rows][cout] fp32, rows) is appended -- the input of `se_gate(..., partial=)`;
    left empty otherwise (the caller then sums with tdvc_channel_sum)."""
    d, ret, Ho, Wo, pc, rec_pc = conv_desc(x, pc, out, act, slope, res, res2, gdn, aux, square, out_dtype, round16, nchw_out, bcast_T, bcast_slope)
    if chan_sum is not None and TAPE is None and FUSE_CHAN_SUM:
        rows = L.lib().tdvc_conv_chan_sum_rows(C.byref(d))
        if rows > 0:
            part = torch.empty((x.N, rows, pc.cout), dtype=torch.float32, device=x.t.device)
            d.chan_sum = part.data_ptr()
            chan_sum.append((part, rows))
    if x.f32 and TAPE is not None and not _IN_BACKWARD:
        raise L.TdvcHipError("conv: the fp32 form has no backward (the fp32 islands are an inference / coding mode)")
    if bcast_T and TAPE is not None and not _IN_BACKWARD:
        raise L.TdvcHipError("conv: bcast_T is an inference-only fusion (under the tape use conv + bcast_add_act)")
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib().tdvc_conv2d(C.byref(d), _stream()), "conv2d")
        e1.record()
        PROFILE.append(dict(kernel=L.lib().tdvc_last_conv_kernel().decode(),
                            shape=f"{pc.kh}x{pc.kw} s{pc.stride} {x.C}->{pc.cout} @{x.H}x{x.W}" + (" f32out" if out_dtype == torch.float32 else "") + (" nchw" if nchw_out is not None else ""),
                            e0=e0, e1=e1,
                            flops=2.0 * x.N * Ho * Wo * pc.cout * x.C * len(pc.taps),
                            flops_real=(x.N * Ho * Wo * pc.flops_per_px) if pc.s2d else 2.0 * x.N * Ho * Wo * pc.cout * pc.cin_real * len(pc.taps),
                            bytes=2.0 * x.N * (x.H * x.W * x.C + Ho * Wo * pc.cout * (4 if pc.shuffle else 1) / (4 if pc.shuffle else 1))))
        _rec("conv", x, rec_pc, None if nchw_out is not None else ret, act, slope, res, res2, gdn, aux, square, nchw_out)
        return ret
    L.check(L.lib().tdvc_conv2d(C.byref(d), _stream()), "conv2d")
    _rec("conv", x, rec_pc, None if nchw_out is not None else ret, act, slope, res, res2, gdn, aux, square, nchw_out)
    return ret


# ----------------------------------------------------------------------------- fused conv pair (inference)
class PackedConvPair:
    """two 3x3 64->64 convs packed for `tdvc_conv_pair` (per-wave v_mfma_f32_16x16x32_f16 A fragments + both biases)"""

    def __init__(self, w: torch.Tensor, bias: torch.Tensor):
        self.w, self.bias = w, bias


def pack_conv_pair(w1: torch.Tensor, b1: torch.Tensor | None, w2: torch.Tensor, b2: torch.Tensor | None) -> PackedConvPair:
    """weights (64, 64, 3, 3) on the device -> [conv][wave][tap*2 + chunk][lane][8] fp16 (the layout of
    tdvc_pack_conv_pair_weights: cout = 16 wave + (lane & 15), cin = 32 chunk + 8 (lane >> 4) + j)"""
    for w in (w1, w2):
        if tuple(w.shape) != (64, 64, 3, 3):
            raise L.TdvcHipError(f"pack_conv_pair: weights must be (64, 64, 3, 3), got {tuple(w.shape)}")
    frag = lambda w: w.detach().float().reshape(4, 16, 2, 4, 8, 9).permute(0, 5, 2, 3, 1, 4)
    w = torch.stack([frag(w1), frag(w2)]).contiguous().to(torch.float16)
    zb = lambda b_: torch.zeros(64, device=w1.device) if b_ is None else b_.detach().float()
    return PackedConvPair(w, torch.cat([zb(b1), zb(b2)]).contiguous())


def _pair_desc(x: FM, pp: PackedConvPair, out: FM, act1, slope1, act2, slope2, add_input, res2):
    d = L.ConvPairDesc()
    d.x, d.y = x.desc(), out.desc()
    d.w, d.bias = pp.w.data_ptr(), pp.bias.data_ptr()
    d.act1, d.slope1, d.act2, d.slope2 = act1, slope1, act2, slope2
    d.add_input = 1 if add_input else 0
    d.res2 = res2.desc() if res2 is not None else L.FMapDesc()
    return d


def conv_pair_supported(x: FM, out: FM | None = None, res2: FM | None = None) -> bool:
    """inference only (nothing is kept for a backward pass); geometry limits are the library's (tdvc_conv_pair_supported)"""
    if TAPE is not None or x.f32 or x.C != 64:
        return False
    d = L.ConvPairDesc()
    d.x = x.desc()
    d.y = out.desc() if out is not None else x.desc()
    d.res2 = res2.desc() if res2 is not None else L.FMapDesc()
    d.w = d.bias = 1            # presence only; never dereferenced by the query
    return bool(L.lib().tdvc_conv_pair_supported(C.byref(d)))


def conv_pair(x: FM, pp: PackedConvPair, out: FM | None = None, act1=ACT_RELU, slope1=0.0, act2=ACT_NONE, slope2=0.0,
              add_input=True, res2: FM | None = None) -> FM:
    """y = act2(conv2(act1(conv1(x)))) [+ x] [+ res2] in one launch (`tdvc_conv_pair`)"""
    if TAPE is not None:
        raise L.TdvcHipError("conv_pair: inference only (the intermediate map is not kept for the backward pass)")
    if out is None:
        out = FM.empty(x.N, x.H, x.W, 64, device=x.t.device)
    d = _pair_desc(x, pp, out, act1, slope1, act2, slope2, add_input, res2)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib().tdvc_conv_pair(C.byref(d), _stream()), "conv_pair")
        e1.record()
        fl = 2 * 2.0 * x.N * x.H * x.W * 64 * 64 * 9
        PROFILE.append(dict(kernel="conv_pair", shape=f"2x(3x3 s1 64->64) @{x.H}x{x.W}", e0=e0, e1=e1, flops=fl, flops_real=fl,
                            bytes=2.0 * x.N * x.H * x.W * 128))
        return out
    L.check(L.lib().tdvc_conv_pair(C.byref(d), _stream()), "conv_pair")
    return out


# ----------------------------------------------------------------------------- conv backward (training path)
def act_backward(g: FM, y: FM, act, slope=0.0, res: FM | None = None, out: FM | None = None) -> FM:
    """g * act'(z); the sign of the pre-activation comes from the stored output y (minus its residual)"""
    out = g if out is None else out
    dg, dy, do = g.desc(), y.desc(), out.desc()
    dr = res.desc() if res is not None else None
    L.check(L.lib().tdvc_act_backward(C.byref(dg), C.byref(dy), C.byref(dr) if dr is not None else None, act, slope, C.byref(do), _stream()),
            "act_backward")
    return out


def pixel_unshuffle(y: FM, out: FM | None = None) -> FM:
    """(N, 2H, 2W, C) -> (N, H, W, 4C), channel (i*2+j)*C + c: the gradient of a sub-pixel conv in packed-row order"""
    if out is None:
        out = FM.empty(y.N, y.H // 2, y.W // 2, 4 * y.C, dtype=y.t.dtype, device=y.t.device)
    dy, do = y.desc(), out.desc()
    L.check(L.lib().tdvc_pixel_unshuffle(C.byref(dy), C.byref(do), _stream()), "pixel_unshuffle")
    return out


def _dgrad_conv(pc: PackedConv, g_channels: int, x_channels: int) -> PackedConv:
    """the conv that maps dY to dX for the layer `pc` (built once, re-packed with it)"""
    if pc.dgrad is None:
        o = pc.orig
        tb0 = dict(g_channels=g_channels, x_channels=x_channels, taps=o["taps"], pad=o["pad"], stride=o["stride"], shuffle=pc.shuffle,
                   cin_perm=o["cin_perm"], device=pc.w.device)
        rows = x_channels * (4 if o["stride"] == 2 else 1)
        kk = 3 if (o["stride"] == 2 and o["kh"] == 3) else o["kh"]
        ck = _pick_ck(g_channels, rows, kk, kk, 1, 1)
        tb = convpack.dgrad_tables(pc.layout, ck=ck, **tb0)
        bias = torch.zeros(_cout_pad(tb.cout), dtype=torch.float32, device=pc.w.device)
        pc.dgrad = PackedConv(_pack_from_tables(pc.wsrc, tb), bias, tb.cout, tb.cin, tb.kh, tb.kw, 1, tb.pad, tb.ck, tb.taps, tb.shuffle,
                              tb.cin, wsrc=pc.wsrc, tables=tb)
    assert pc.dgrad.cin == g_channels
    return pc.dgrad


def conv_dgrad(pc: PackedConv, g: FM, dx: FM, accumulate=True, extra: FM | None = None) -> FM:
    """dX (+)= dL/dx of y = conv(x, W) given g = dL/dy (pre-activation).  Runs on the forward conv kernels with
    the weights packed transposed / mirrored (`convpack.dgrad_tables`); for a sub-pixel layer `g` is the
    un-shuffled gradient (`pixel_unshuffle`), for a stride-2 layer the result is stored through PixelShuffle."""
    dpc = _dgrad_conv(pc, g.C, dx.C)
    if accumulate:
        return conv(g, dpc, out=dx, res=dx, res2=extra)
    return conv(g, dpc, out=dx, res=extra)          # `extra`: one more gradient of the same geometry added in the epilogue (Tape.add_identity)


class WgradBatch:
    """Deferred second stages of conv weight gradients (training: ~220 layers per step, each order-fixed reduce a 15 us launch of its own,
    3.4 ms of the side stream).  `conv_wgrad(..., defer=batch)` runs the first stage only; `flush()` reduces everything collected so far
    in ONE launch of `tdvc_wgrad_reduce_batch` per round, on the current stream (the one the first stages ran on).  Jobs that add to the
    same parameter (shared layers, per-image launches of one layer) go to successive rounds in their order of arrival, so the sums and
    their order are those of the immediate form: bit-identical gradients."""

    def __init__(self):
        self.items = []

    def add(self, job, keep):
        self.items.append((job, keep))

    def flush(self):
        if not self.items:
            return
        rounds, nxt = [], {}
        for job, _ in self.items:
            r = nxt.get(job.dw, 0)
            nxt[job.dw] = r + 1
            if len(rounds) <= r:
                rounds.append([])
            rounds[r].append(job)
        lib = L.lib()
        for jobs in rounds:
            n = len(jobs)
            arr = (L.WgradReduceJob * n)(*jobs)
            starts = (C.c_int32 * (n + 1))()
            for i, j in enumerate(jobs):
                starts[i + 1] = starts[i] + j.nblocks
            nb_j, nb_s = C.sizeof(arr), C.sizeof(starts)
            host = torch.empty((nb_j + nb_s,), dtype=torch.uint8, pin_memory=True)
            C.memmove(host.data_ptr(), arr, nb_j)
            C.memmove(host.data_ptr() + nb_j, starts, nb_s)
            dev = host.to(jobs_device(jobs), non_blocking=True)
            if torch.cuda.is_current_stream_capturing():
                _GRAPH_KEEP.append((host, dev))          # a captured copy node re-reads the pinned table at every replay
            L.check(lib.tdvc_wgrad_reduce_batch(dev.data_ptr(), dev.data_ptr() + nb_j, n, int(starts[n]), _stream()), "wgrad_reduce_batch")
        self.items.clear()


_GRAPH_KEEP: list = []


def jobs_device(jobs):
    return torch.device("cuda", torch.cuda.current_device())


def conv_wgrad(pc: PackedConv, g: FM, x: FM, dw: torch.Tensor, scale=1.0, square_x=False, db: torch.Tensor | None = None,
               defer: WgradBatch | None = None) -> None:
    """dW += scale * dL/dW (fp32, the parameter's own layout); `g` as in conv_dgrad.  With `db` the bias gradient
    (what `conv_bgrad` computes) comes out of the same launch: the kernel already holds the dY tiles.  `pc.orig["wgrad_taps"]` widens
    the tap list beyond the forward's (masked convs: the reference's autograd also fills the masked taps)."""
    o = pc.orig
    taps = o.get("wgrad_taps") or o["taps"]
    tb = pc.__dict__.get("_wg_tables")
    if tb is None:                               # plain-geometry tables (also for layers that RUN in space-to-depth form)
        tb = convpack.forward_tables(pc.layout, cin_pad=x.C, taps=taps, pad=o["pad"], ck=8, shuffle=pc.shuffle, cin_perm=o["cin_perm"],
                                     device=pc.w.device)
        pc.__dict__["_wg_tables"] = tb
    assert dw.is_cuda and dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == pc.wsrc.numel()
    lib = L.lib()
    Ho, Wo = g.H, g.W
    wk = pc.__dict__.get("_wg_static")          # per layer and geometry: workspace size, tap arrays (built once)
    if wk is None or wk[0] != (x.C, x.N, Ho, Wo):
        wk = pc.__dict__["_wg_static"] = ((x.C, x.N, Ho, Wo), lib.tdvc_conv_wgrad_work_floats(pc.cout, x.C, len(taps), x.N, Ho, Wo),
                                          (C.c_int8 * len(taps))(*[t[0] for t in taps]), (C.c_int8 * len(taps))(*[t[1] for t in taps]))
    _, nwork, dy, dxs = wk
    work = torch.empty((nwork,), dtype=torch.float32, device=dw.device)
    if FORCE_KEEP is not None:
        FORCE_KEEP.append(work)
    dg, dxd = g.desc(), x.desc()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    bidx = None
    if db is not None:
        assert db.is_cuda and db.dtype == torch.float32 and db.is_contiguous() and db.numel() >= pc.cout
        bidx = _bias_index(pc, db.device)
    if defer is not None and PROFILE is None:
        job = L.WgradReduceJob()
        L.check(lib.tdvc_conv_wgrad_partials(C.byref(dg), C.byref(dxd), pc.cout, o["kh"], o["kw"], o["stride"], o["pad"], len(taps), dy, dxs,
                                             tb.row_off.data_ptr(), tb.chan_off.data_ptr(), tb.tap_off.data_ptr(), int(square_x), scale, dw.data_ptr(),
                                             bidx.data_ptr() if bidx is not None else None, db.data_ptr() if db is not None else None,
                                             work.data_ptr(), nwork, C.byref(job), _stream()), "conv_wgrad_partials")
        defer.add(job, (work, dw, db, tb, bidx))             # the workspace and the tables live until the flush
        return
    L.check(lib.tdvc_conv_wgrad_bias(C.byref(dg), C.byref(dxd), pc.cout, o["kh"], o["kw"], o["stride"], o["pad"], len(taps), dy, dxs,
                                tb.row_off.data_ptr(), tb.chan_off.data_ptr(), tb.tap_off.data_ptr(), int(square_x), scale, dw.data_ptr(),
                                bidx.data_ptr() if bidx is not None else None, db.data_ptr() if db is not None else None,
                                work.data_ptr(), nwork, _stream()), "conv_wgrad")
    if PROFILE is not None:
        e1.record()
        PROFILE.append(dict(kernel="conv_wgrad", shape=f"{o['kh']}x{o['kw']} s{o['stride']} {x.C}->{pc.cout} @{x.N}x{x.H}x{x.W}", e0=e0, e1=e1,
                            flops=2.0 * x.N * Ho * Wo * pc.cout * x.C * len(taps), flops_real=2.0 * x.N * Ho * Wo * pc.cout * pc.cin_real * len(taps),
                            bytes=0.0))


def _bias_index(pc: PackedConv, device):
    idx = pc.__dict__.get("_bg_index")
    if idx is None and pc.shuffle:
        idx = torch.from_numpy(convpack.shuffle_perm(pc.cout)).to(torch.int32).to(device)
        pc.__dict__["_bg_index"] = idx
    return idx


def conv_bgrad(pc: PackedConv, g: FM, db: torch.Tensor, scale=1.0) -> None:
    """db += scale * sum over batch and pixels of g (rows un-permuted for sub-pixel layers)"""
    lib = L.lib()
    idx = _bias_index(pc, db.device)
    nwork = lib.tdvc_bias_grad_work_floats(g.N, g.C)
    work = torch.empty((nwork,), dtype=torch.float32, device=db.device)
    dg = g.desc()
    L.check(lib.tdvc_bias_grad(C.byref(dg), pc.cout, idx.data_ptr() if idx is not None else None, scale, db.data_ptr(), work.data_ptr(), nwork,
                               _stream()), "bias_grad")


def gdn_backward(g: FM, x: FM, n32: FM, inverse: bool, dx: FM) -> FM:
    """-> dn (fp16); dx += g * n^(-+1/2)"""
    dn = FM.empty(g.N, g.H, g.W, g.C, device=g.t.device)
    d1, d2, d3, d4, d5 = g.desc(), x.desc(), n32.desc(), dn.desc(), dx.desc()
    L.check(L.lib().tdvc_gdn_backward(C.byref(d1), C.byref(d2), C.byref(d3), int(inverse), C.byref(d4), C.byref(d5), _stream()), "gdn_backward")
    return dn


def mul2_accumulate(dx: FM, x: FM, t: FM) -> None:
    d1, d2, d3 = dx.desc(), x.desc(), t.desc()
    L.check(L.lib().tdvc_mul2_accumulate(C.byref(d1), C.byref(d2), C.byref(d3), _stream()), "mul2_accumulate")


def copy_cast(src: FM, dst: FM) -> FM:
    ds, dd = src.desc(), dst.desc()
    L.check(L.lib().tdvc_copy_cast(C.byref(ds), C.byref(dd), _stream()), "copy_cast")
    return dst


def clamp01_backward(g: FM, y: FM) -> FM:
    dg, dy = g.desc(), y.desc()
    L.check(L.lib().tdvc_clamp01_backward(C.byref(dg), C.byref(dy), _stream()), "clamp01_backward")
    return g


def gate_backward(g: FM, a: FM, gate: torch.Tensor, da: FM | None, dgate: torch.Tensor) -> None:
    lib = L.lib()
    nwork = lib.tdvc_gate_backward_work_floats(g.N, g.C)
    work = torch.empty((nwork,), dtype=torch.float32, device=gate.device)
    dg, dad = g.desc(), a.desc()
    dda = da.desc() if da is not None else None
    L.check(lib.tdvc_gate_backward(C.byref(dg), C.byref(dad), gate.data_ptr(), C.byref(dda) if dda is not None else None, dgate.data_ptr(),
                                   work.data_ptr(), nwork, _stream()), "gate_backward")


def se_gate_backward(p: "SEParams", partial: torch.Tensor, nblocks: int, npix: int, gate: torch.Tensor, dgate: torch.Tensor, scale: float,
                     grads) -> torch.Tensor:
    """-> dmean [N][C]; parameter gradients accumulate into `grads` = (dw1, db1, dw2, db2)"""
    N = gate.shape[0]
    dmean = torch.empty_like(gate)
    L.check(L.lib().tdvc_se_gate_backward(partial.data_ptr(), nblocks, 1.0 / npix, N, p.C, p.Cmid, p.w1.data_ptr(), p.b1.data_ptr(),
                                          p.w2.data_ptr(), p.b2.data_ptr(), gate.data_ptr(), dgate.data_ptr(), scale, dmean.data_ptr(),
                                          grads[0].data_ptr(), grads[1].data_ptr(), grads[2].data_ptr(), grads[3].data_ptr(), _stream()),
            "se_gate_backward")
    return dmean


def bcast_channel_add(dx: FM, v: torch.Tensor, scale: float) -> None:
    d = dx.desc()
    L.check(L.lib().tdvc_bcast_channel_add(C.byref(d), v.data_ptr(), scale, _stream()), "bcast_channel_add")


def add_flow_backward(doff: FM, dflow: FM) -> None:
    do, df = doff.desc(), dflow.desc()
    L.check(L.lib().tdvc_add_flow_backward(C.byref(do), C.byref(df), _stream()), "add_flow_backward")


def bcast_add_act_backward(dx: FM, x: FM, db: FM, slope: float) -> None:
    d1, d2, d3 = dx.desc(), x.desc(), db.desc()
    L.check(L.lib().tdvc_bcast_add_act_backward(C.byref(d1), C.byref(d2), C.byref(d3), slope, _stream()), "bcast_add_act_backward")


def upsample2x_backward(dy: FM, dx: FM) -> None:
    d1, d2 = dy.desc(), dx.desc()
    L.check(L.lib().tdvc_upsample2x_backward(C.byref(d1), C.byref(d2), _stream()), "upsample2x_backward")


DCN_PLANAR_MIN_PIXELS = 1 << 16      # below this the map is L2-resident anyway and the extra pass does not pay
DCN_PLANAR = __import__("os").environ.get("TDVC_DCN_PLANAR", "0") == "1"


def dcn_fused(x: FM, om: FM, pc: PackedConv, out: FM, groups=8, act=ACT_NONE, slope=0.0, round16=True, planar: bool | None = None) -> FM:
    """`planar` (default: large maps): gather from a group-planar copy of x made by the same call (one extra pass over x;
    the 288 bilinear corner gathers per pixel then share 128-byte lines between neighbouring pixels of a group)"""
    d = L.DcnDesc()
    d.x, d.om, d.y = x.desc(), om.desc(), out.desc()
    d.w, d.bias = pc.w.data_ptr(), pc.bias.data_ptr()
    d.groups, d.act, d.slope, d.round_before_act = groups, act, slope, int(round16)
    if planar is None:
        planar = DCN_PLANAR and x.H * x.W >= DCN_PLANAR_MIN_PIXELS
    scratch = torch.empty((x.N, groups, x.H, x.W, 8), dtype=torch.float16, device=x.t.device) if planar else None
    d.x_planar = scratch.data_ptr() if scratch is not None else None
    L.check(L.lib().tdvc_dcn_fused(C.byref(d), _stream()), "dcn_fused")
    _rec("dcn_fused", x, om, pc, out, groups, act, slope)
    return out


def dcn_columns(x: FM, om: FM, groups: int) -> FM:
    col = FM.empty(x.N, x.H, x.W, 72 * groups, device=x.t.device)
    d1, d2, d3 = x.desc(), om.desc(), col.desc()
    L.check(L.lib().tdvc_dcn_columns(C.byref(d1), C.byref(d2), groups, C.byref(d3), _stream()), "dcn_columns")
    return col


# Run-to-run reproducible training (tests/test_model_gpu.py trains its operating-point model with it): the one operator whose
# result depends on arrival order -- the float atomics of the DCN's bilinear scatter for samples displaced out of their
# tile's window -- records those samples and adds them in sorted key order instead.  Costs a host synchronisation per step.
DETERMINISTIC = False


def dcn_col2im(x: FM, om: FM, dcol: FM, groups: int, dom: FM) -> FM:
    """-> dx as an fp32 FM (fresh; scatter through per-tile windows); offset / mask gradients accumulate into `dom`"""
    dx32 = FM.zeros(x.N, x.H, x.W, x.C, dtype=torch.float32, device=x.t.device)
    lib = L.lib()
    nwork = lib.tdvc_dcn_col2im_work_floats(x.N, x.H, x.W, groups)
    work = torch.empty((nwork,), dtype=torch.float32, device=x.t.device)
    d1, d2, d3, d4 = x.desc(), om.desc(), dcol.desc(), dom.desc()
    if DETERMINISTIC:
        dev = x.t.device
        cap = max(1 << 16, x.N * x.H * x.W * groups * 36 // 8)          # an eighth of all samples far out of their window
        keys = torch.empty((cap,), dtype=torch.int64, device=dev)
        vals = torch.empty((cap, 8), dtype=torch.float32, device=dev)
        cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
        L.check(lib.tdvc_dcn_col2im_det(C.byref(d1), C.byref(d2), C.byref(d3), groups, dx32.t.data_ptr(), C.byref(d4), work.data_ptr(), nwork,
                                        keys.data_ptr(), vals.data_ptr(), cnt.data_ptr(), cap, _stream()), "dcn_col2im_det")
        n = int(cnt.item())
        if n > cap:
            raise L.TdvcHipError(f"dcn_col2im (deterministic): {n} far samples exceed the record capacity {cap}")
        if n:
            ks, order = torch.sort(keys[:n], stable=True)
            L.check(lib.tdvc_dcn_far_apply(ks.data_ptr(), order.data_ptr(), vals.data_ptr(), n, dx32.t.data_ptr(), _stream()), "dcn_far_apply")
        return dx32
    L.check(lib.tdvc_dcn_col2im(C.byref(d1), C.byref(d2), C.byref(d3), groups, dx32.t.data_ptr(), C.byref(d4), work.data_ptr(), nwork, _stream()),
            "dcn_col2im")
    return dx32


def dcn_column_conv(pc: PackedConv, groups: int) -> PackedConv:
    """the DCN weight (cout, 8G, 3, 3) as a 1x1 conv over the 72G column channels k = g*72 + t*8 + j, input channel
    c = g*8 + j (weight offset c*9 + t)"""
    cp = pc.__dict__.get("_colpc")
    if cp is None:
        cin = 8 * groups
        chan = np.array([(g * 8 + j) * 9 + t for g in range(groups) for t in range(9) for j in range(8)], dtype=np.int64)
        lay = convpack.WeightLayout(pc.cout, 9 * cin, 1, 1, np.arange(pc.cout, dtype=np.int64) * cin * 9, chan, np.zeros(1, dtype=np.int64))
        cp = pack_conv(pc.wsrc, None, stride=1, pad=0, layout=lay, param_w=pc.param_w)
        pc.__dict__["_colpc"] = cp
    return cp


def sigmoid_f32(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(x)
    L.check(L.lib().tdvc_sigmoid_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "sigmoid_f32")
    return out


def sigmoid_backward_f32(g: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    L.check(L.lib().tdvc_sigmoid_backward_f32(g.data_ptr(), s.data_ptr(), g.numel(), _stream()), "sigmoid_backward_f32")
    return g


def axpy_f32(dst: torch.Tensor, src: torch.Tensor, scale: float) -> None:
    assert dst.numel() == src.numel() and dst.dtype == src.dtype == torch.float32 and dst.is_contiguous() and src.is_contiguous()
    L.check(L.lib().tdvc_axpy_f32(dst.data_ptr(), src.data_ptr(), scale, dst.numel(), _stream()), "axpy_f32")


# ----------------------------------------------------------------------------- elementwise / SE
def scale_act_res(a: FM, out: FM, gate: torch.Tensor | None = None, act=ACT_NONE, slope=0.0, res: FM | None = None,
                  res_sign=1.0, out2: FM | None = None) -> FM:
    da, dy = a.desc(), out.desc()
    dr = res.desc() if res is not None else None
    d2 = out2.desc() if out2 is not None else None
    L.check(L.lib().tdvc_scale_act_res(C.byref(da), gate.data_ptr() if gate is not None else None, act, slope,
                                       C.byref(dr) if dr is not None else None, res_sign, C.byref(dy),
                                       C.byref(d2) if d2 is not None else None, _stream()), "scale_act_res")
    _rec("scale_act_res", a, out, gate, act, slope, res, res_sign, out2)
    return out


def clone(x: FM) -> FM:
    """a recorded copy: the in-place kernels below run on it under the tape, so that the producer of `x` still finds
    its own output (activation sign, weight-gradient operand) in the backward pass"""
    out = copy_cast(x, FM.empty(x.N, x.H, x.W, x.C, dtype=x.t.dtype, device=x.t.device))
    _rec("clone", x, out)
    return out


def add_flow(off: FM, flow: FM):
    do, df = off.desc(), flow.desc()
    L.check(L.lib().tdvc_add_flow(C.byref(do), C.byref(df), _stream()), "add_flow")
    _rec("add_flow", off, flow)


def bcast_add_act(x: FM, b: FM, T: int, slope: float):
    dx, db = x.desc(), b.desc()
    L.check(L.lib().tdvc_bcast_add_act(C.byref(dx), C.byref(db), T, slope, _stream()), "bcast_add_act")
    _rec("bcast_add_act", x, b, slope)


@dataclass
class SEParams:
    w1: torch.Tensor
    b1: torch.Tensor
    w2: torch.Tensor
    b2: torch.Tensor
    C: int
    Cmid: int
    params: tuple = ()       # the four nn.Parameters (gradient accumulators), same storage as w1 / b1 / w2 / b2


FUSE_CHAN_SUM = True           # A/B switch: the SELayer's channel sums from the producing conv's epilogue (conv(..., chan_sum=[]))


def se_gate(x: FM, p: SEParams, partial=None) -> torch.Tensor:
    """gate[N][C] fp32 (`main/model/inflate.py:204-208` without the final multiply).  `partial` = (tensor [N][rows][C], rows): the
    channel sums of `x` that its producer already wrote (conv(..., chan_sum=)); otherwise one pass over x (tdvc_channel_sum)."""
    npix = x.H * x.W
    gate = torch.empty((x.N, x.C), dtype=torch.float32, device=x.t.device)
    lib = L.lib()
    if partial is not None:
        partial, nblocks = partial
        assert tuple(partial.shape) == (x.N, nblocks, x.C) and partial.dtype == torch.float32
    else:
        nblocks = max(1, min(1024, npix // 256))      # 4 workgroups per CU at full resolution: 64 KB of loads in flight per CU
        partial = torch.empty((x.N, nblocks, x.C), dtype=torch.float32, device=x.t.device)
        dx = x.desc()
        L.check(lib.tdvc_channel_sum(C.byref(dx), partial.data_ptr(), nblocks, _stream()), "channel_sum")
    L.check(lib.tdvc_se_gate(partial.data_ptr(), nblocks, 1.0 / npix, x.N, p.C, p.Cmid, p.w1.data_ptr(),
                             p.b1.data_ptr(), p.w2.data_ptr(), p.b2.data_ptr(), gate.data_ptr(), _stream()), "se_gate")
    _rec("se_gate", x, p, partial, nblocks, gate)
    return gate


# ----------------------------------------------------------------------------- resampling
def upsample2x(x: FM, out: FM | None = None) -> FM:
    if out is None:
        out = FM.empty(x.N, 2 * x.H, 2 * x.W, x.C, device=x.t.device)
    dx, dy = x.desc(), out.desc()
    L.check(L.lib().tdvc_upsample2x(C.byref(dx), C.byref(dy), _stream()), "upsample2x")
    _rec("upsample2x", x, out)
    return out


def avgpool2(x: FM) -> FM:
    out = FM.empty(x.N, x.H // 2, x.W // 2, x.C, dtype=torch.float32, device=x.t.device)
    dx, dy = x.desc(), out.desc()
    L.check(L.lib().tdvc_avgpool2(C.byref(dx), C.byref(dy), _stream()), "avgpool2")
    return out


def spynet_level_input(ref: FM, supp: FM, flow_lo: FM | None, flow_up: FM, cat8: FM):
    dr, ds, du, dc = ref.desc(), supp.desc(), flow_up.desc(), cat8.desc()
    dl = flow_lo.desc() if flow_lo is not None else None
    L.check(L.lib().tdvc_spynet_level_input(C.byref(dr), C.byref(ds), C.byref(dl) if dl is not None else None,
                                            C.byref(du), C.byref(dc), _stream()), "spynet_level_input")
    _rec("spynet_level_input", supp, flow_lo, flow_up, cat8)


def spynet_level_input_backward(supp: FM, flow_up: FM, dcat8: FM, dflow_up: FM, dflow_lo: FM | None) -> None:
    d1, d2, d3, d4 = supp.desc(), flow_up.desc(), dcat8.desc(), dflow_up.desc()
    d5 = dflow_lo.desc() if dflow_lo is not None else None
    L.check(L.lib().tdvc_spynet_level_input_backward(C.byref(d1), C.byref(d2), C.byref(d3), C.byref(d4),
                                                     C.byref(d5) if d5 is not None else None, _stream()), "spynet_level_input_backward")


def resize_bilinear(x: FM, H: int, W: int, chscale: torch.Tensor | None = None) -> FM:
    out = FM.empty(x.N, H, W, x.C, dtype=torch.float32, device=x.t.device)
    dx, dy = x.desc(), out.desc()
    L.check(L.lib().tdvc_resize_bilinear(C.byref(dx), C.byref(dy), chscale.data_ptr() if chscale is not None else None,
                                         _stream()), "resize_bilinear")
    _rec("resize_bilinear", x, out, chscale)
    return out


def resize_bilinear_backward(dy: FM, dx: FM, chscale: torch.Tensor | None = None) -> None:
    d1, d2 = dy.desc(), dx.desc()
    L.check(L.lib().tdvc_resize_bilinear_backward(C.byref(d1), C.byref(d2), chscale.data_ptr() if chscale is not None else None, _stream()),
            "resize_bilinear_backward")


# ----------------------------------------------------------------------------- in-loop filter matching
def avgpool_k(x: FM, scale: int) -> torch.Tensor:
    hp, wp = x.H // scale, x.W // scale
    pooled = torch.empty((x.N, hp, wp, x.C), dtype=torch.float32, device=x.t.device)
    nwork = L.lib().tdvc_avgpool_k_work_floats(x.N, hp, wp, x.C, scale)
    work = torch.empty((nwork,), dtype=torch.float32, device=x.t.device)
    dx = x.desc()
    L.check(L.lib().tdvc_avgpool_k(C.byref(dx), scale, pooled.data_ptr(), hp, wp, work.data_ptr(), nwork, _stream()), "avgpool_k")
    return pooled


def patch_match(pin: torch.Tensor, pref: torch.Tensor) -> torch.Tensor:
    N, hp, wp, Cc = pin.shape
    Lp = ((hp + 3) // 3 + 1) * ((wp + 3) // 3 + 1)
    idx = torch.empty((N, Lp), dtype=torch.int32, device=pin.device)
    L.check(L.lib().tdvc_patch_match(pin.data_ptr(), pref.data_ptr(), N, hp, wp, Cc, idx.data_ptr(), _stream()),
            "patch_match")
    return idx


def match_gather(fin: FM, fref: FM, idx: torch.Tensor, scale: int, cat: FM):
    df, dr, dc = fin.desc(), fref.desc(), cat.desc()
    L.check(L.lib().tdvc_match_gather(C.byref(df), C.byref(dr), idx.data_ptr(), scale, fin.H // scale, fin.W // scale,
                                      C.byref(dc), _stream()), "match_gather")
    _rec("match_gather", fin, fref, idx, scale, cat)


def match_gather_backward(fin: FM, fref: FM, idx: torch.Tensor, scale: int, dcat: FM, dfin: FM, dfref: FM) -> None:
    d1, d2, d3, d4, d5 = fin.desc(), fref.desc(), dcat.desc(), dfin.desc(), dfref.desc()
    L.check(L.lib().tdvc_match_gather_backward(C.byref(d1), C.byref(d2), idx.data_ptr(), scale, fin.H // scale, fin.W // scale,
                                               C.byref(d3), C.byref(d4), C.byref(d5), _stream()), "match_gather_backward")


# ----------------------------------------------------------------------------- entropy model
def eb_forward(z: FM, params: torch.Tensor, z_hat: FM, bits_out: torch.Tensor, noise: FM | None = None):
    numel = z.N * z.H * z.W * z.C
    cap = (numel + 255) // 256
    partial = torch.empty(cap, dtype=torch.float32, device=z.t.device)
    dz, dh = z.desc(), z_hat.desc()
    dn = noise.desc() if noise is not None else None
    L.check(L.lib().tdvc_eb_forward(C.byref(dz), params.data_ptr(), C.byref(dn) if dn is not None else None, C.byref(dh),
                                    bits_out.data_ptr(), partial.data_ptr(), cap, _stream()), "eb_forward")
    _rec("eb_forward", z, params, z_hat, noise)


def eb_pack(raw_table: torch.Tensor, quantiles: torch.Tensor, packed: torch.Tensor, Cn: int) -> None:
    L.check(L.lib().tdvc_eb_pack(raw_table.data_ptr(), quantiles.data_ptr(), packed.data_ptr(), Cn, _stream()), "eb_pack")


def eb_param_chain(dpacked: torch.Tensor, raw_table: torch.Tensor, grad_table: torch.Tensor, scale: float, Cn: int) -> None:
    assert dpacked.is_contiguous() and dpacked.dtype == torch.float32 and tuple(dpacked.shape) == (Cn, 59)
    L.check(L.lib().tdvc_eb_param_chain(dpacked.data_ptr(), raw_table.data_ptr(), grad_table.data_ptr(), scale, Cn, _stream()), "eb_param_chain")


def eb_aux(params: torch.Tensor, quantiles: torch.Tensor, target: float, dq: torch.Tensor, loss: torch.Tensor, Cn: int) -> None:
    assert quantiles.is_contiguous() and dq.is_contiguous() and quantiles.numel() == 3 * Cn == dq.numel() and dq.dtype == torch.float32
    L.check(L.lib().tdvc_eb_aux(params.data_ptr(), quantiles.data_ptr(), target, dq.data_ptr(), loss.data_ptr(), Cn, _stream()), "eb_aux")


def eb_backward(z: FM, params: torch.Tensor, noise: FM, gscale: float, dz: FM, dparams: torch.Tensor) -> None:
    d1, d2, d3 = z.desc(), noise.desc(), dz.desc()
    L.check(L.lib().tdvc_eb_backward(C.byref(d1), params.data_ptr(), C.byref(d2), gscale, C.byref(d3), dparams.data_ptr(), _stream()), "eb_backward")


def gc_backward(y: FM, gp: FM, noise: FM, gscale: float, dy: FM, dgp: FM) -> None:
    d1, d2, d3, d4, d5 = y.desc(), gp.desc(), noise.desc(), dy.desc(), dgp.desc()
    L.check(L.lib().tdvc_gc_backward(C.byref(d1), C.byref(d2), C.byref(d3), gscale, C.byref(d4), C.byref(d5), _stream()), "gc_backward")


def gc_forward(y: FM, gp: FM, bits_out: torch.Tensor, noise: FM | None = None):
    numel = y.N * y.H * y.W * y.C
    cap = (numel + 255) // 256
    partial = torch.empty(cap, dtype=torch.float32, device=y.t.device)
    dy, dg = y.desc(), gp.desc()
    dn = noise.desc() if noise is not None else None
    L.check(L.lib().tdvc_gc_forward(C.byref(dy), C.byref(dg), C.byref(dn) if dn is not None else None,
                                    bits_out.data_ptr(), partial.data_ptr(), cap, _stream()), "gc_forward")
    _rec("gc_forward", y, gp, noise)


def quantize(y: FM, out: FM, noise: FM | None = None) -> FM:
    dy, do = y.desc(), out.desc()
    dn = noise.desc() if noise is not None else None
    L.check(L.lib().tdvc_quantize(C.byref(dy), C.byref(dn) if dn is not None else None, C.byref(do), _stream()), "quantize")
    _rec("quantize", y, out, noise)
    return out


# ----------------------------------------------------------------------------- entropy coding (host + AR kernels)
def pmf_to_quantized_cdf(pmf: np.ndarray, precision: int = 16) -> np.ndarray:
    p = np.ascontiguousarray(pmf, dtype=np.float32)
    out = np.zeros(p.size + 1, dtype=np.int32)
    L.check(L.lib().tdvc_pmf_to_quantized_cdf(p.ctypes.data, p.size, precision, out.ctypes.data), "pmf_to_quantized_cdf")
    return out


class CdfTables:
    """host copies of an entropy model's quantised CDFs in the layout the range coder takes"""

    def __init__(self, cdf: torch.Tensor, lengths: torch.Tensor, offsets: torch.Tensor):
        self.cdf = np.ascontiguousarray(cdf.detach().cpu().numpy(), dtype=np.int32)
        self.sizes = np.ascontiguousarray(lengths.detach().cpu().numpy().reshape(-1), dtype=np.int32)
        self.offsets = np.ascontiguousarray(offsets.detach().cpu().numpy().reshape(-1), dtype=np.int32)
        self.stride = self.cdf.shape[1]


def rans_encode(symbols: np.ndarray, indexes: np.ndarray, t: CdfTables) -> bytes:
    s = np.ascontiguousarray(symbols.reshape(-1), dtype=np.int32)
    i = np.ascontiguousarray(indexes.reshape(-1), dtype=np.int32)
    out = np.empty(4 * s.size + 64, dtype=np.uint8)
    n = L.lib().tdvc_rans_encode(s.ctypes.data, i.ctypes.data, s.size, t.cdf.ctypes.data, t.stride, t.sizes.ctypes.data,
                                 t.offsets.ctypes.data, out.ctypes.data, out.size)
    if n < 0:
        L.check(int(n), "rans_encode")
    return out[:n].tobytes()


class RansDecoder:
    def __init__(self, data: bytes):
        self._buf = np.frombuffer(data, dtype=np.uint8)
        self._h = L.lib().tdvc_rans_decoder_create(self._buf.ctypes.data, self._buf.size)
        if not self._h:
            L.check(-1, "rans_decoder_create")

    def decode(self, indexes: np.ndarray, t: CdfTables) -> np.ndarray:
        i = np.ascontiguousarray(indexes.reshape(-1), dtype=np.int32)
        out = np.empty(i.size, dtype=np.int32)
        L.check(L.lib().tdvc_rans_decoder_decode(self._h, i.ctypes.data, i.size, t.cdf.ctypes.data, t.stride,
                                                 t.sizes.ctypes.data, t.offsets.ctypes.data, out.ctypes.data), "rans_decode")
        return out

    def close(self):
        if self._h:
            L.lib().tdvc_rans_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def ar_decode_serial(data: bytes, t: CdfTables, y_hat: FM, params: FM, x1: FM, pc: FM, descs: list, gp: FM, pos_table: torch.Tensor,
                     M: int, W: int, scale_table: torch.Tensor, idx: torch.Tensor, sym: torch.Tensor) -> None:
    """the decoder's serial context loop of one image (`tdvc_ar_decode_serial`); fills y_hat, sym, idx"""
    buf = np.frombuffer(data, dtype=np.uint8)
    arr = (L.ConvDesc * len(descs))(*descs)
    dy, dp, dx, dc, dg = y_hat.desc(), params.desc(), x1.desc(), pc.desc(), gp.desc()
    L.check(L.lib().tdvc_ar_decode_serial(buf.ctypes.data, buf.size, t.cdf.ctypes.data, t.stride, t.sizes.ctypes.data, t.offsets.ctypes.data,
                                          C.byref(dy), C.byref(dp), C.byref(dx), C.byref(dc), arr, len(descs), C.byref(dg),
                                          pos_table.data_ptr(), pos_table.shape[0], M, W, scale_table.data_ptr(), scale_table.numel(),
                                          idx.data_ptr(), sym.data_ptr(), _stream()), "ar_decode_serial")


def ar_wavefront(data: bytes | None, t: CdfTables | None, y: FM | None, y_hat: FM, params: FM, x1: FM, pc: FM, descs: list, gp: FM,
                 pos: torch.Tensor, step_sizes: np.ndarray, M: int, W: int, scale_table: torch.Tensor, idx: torch.Tensor, sym: torch.Tensor) -> None:
    """the context loop of one image over anti-diagonals (`tdvc_ar_wavefront`): encoder with `y`, decoder with `data` + `t`"""
    buf = np.frombuffer(data, dtype=np.uint8) if data is not None else None
    arr = (L.ConvDesc * len(descs))(*descs)
    ss = np.ascontiguousarray(step_sizes, dtype=np.int32)
    dyh, dp, dx, dc, dg = y_hat.desc(), params.desc(), x1.desc(), pc.desc(), gp.desc()
    dy = y.desc() if y is not None else None
    L.check(L.lib().tdvc_ar_wavefront(buf.ctypes.data if buf is not None else None, buf.size if buf is not None else 0,
                                      t.cdf.ctypes.data if t else None, t.stride if t else 0, t.sizes.ctypes.data if t else None,
                                      t.offsets.ctypes.data if t else None, C.byref(dy) if dy is not None else None, C.byref(dyh), C.byref(dp),
                                      C.byref(dx), C.byref(dc), arr, len(descs), C.byref(dg), pos.data_ptr(), ss.ctypes.data, ss.size, M, W,
                                      scale_table.data_ptr(), scale_table.numel(), idx.data_ptr(), sym.data_ptr(), _stream()), "ar_wavefront")


def ar_gather(y_hat: FM, params: FM, pos: torch.Tensor, npos: int, x1: FM, pc: FM):
    dy, dp, dx, dc = y_hat.desc(), params.desc(), x1.desc(), pc.desc()
    L.check(L.lib().tdvc_ar_gather(C.byref(dy), C.byref(dp), pos.data_ptr(), npos, C.byref(dx), C.byref(dc), _stream()), "ar_gather")


def ar_quantize(y: FM | None, gp: FM, pos: torch.Tensor, npos: int, table: torch.Tensor, y_hat: FM, symbols: torch.Tensor,
                indexes: torch.Tensor, symbols_in: torch.Tensor | None = None):
    dg, dh = gp.desc(), y_hat.desc()
    dy = y.desc() if y is not None else None
    L.check(L.lib().tdvc_ar_quantize(C.byref(dy) if dy is not None else None, C.byref(dg), pos.data_ptr(), npos, table.data_ptr(),
                                     table.numel(), symbols_in.data_ptr() if symbols_in is not None else None, C.byref(dh),
                                     symbols.data_ptr(), indexes.data_ptr(), _stream()), "ar_quantize")


def ar_indexes(gp: FM, pos: torch.Tensor, npos: int, table: torch.Tensor, M: int, W: int, indexes: torch.Tensor):
    dg = gp.desc()
    L.check(L.lib().tdvc_ar_indexes(C.byref(dg), pos.data_ptr(), npos, table.data_ptr(), table.numel(), M, W,
                                    indexes.data_ptr(), _stream()), "ar_indexes")


def round_symbols(z: FM, median: torch.Tensor) -> torch.Tensor:
    out = torch.empty((z.N, z.H, z.W, z.C), dtype=torch.int32, device=z.t.device)
    dz = z.desc()
    L.check(L.lib().tdvc_round_symbols(C.byref(dz), median.data_ptr(), out.data_ptr(), _stream()), "round_symbols")
    return out


# ----------------------------------------------------------------------------- `_ext` operator
def dcn_v2_forward(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, deformable_group):
    """Same arity / semantics as `_ext.dcn_v2_forward` (src/vision.cpp:3-8): fp32 contiguous
    NCHW cuda tensors in, freshly allocated output out; raises RuntimeError on violations
    (the reference raises through AT_ASSERTM, src/cuda/dcn_v2_cuda.cu:38-62)."""
    ts = (input, weight, bias, offset, mask)
    for t in ts:
        if not (torch.is_tensor(t) and t.is_cuda):
            raise RuntimeError("dcn_v2_forward: all tensors must be CUDA/HIP tensors (Not compiled with CPU support)")
        if t.dtype != torch.float32:
            raise RuntimeError("dcn_v2_forward: fp32 tensors only")
    input, weight, bias, offset, mask = [t.contiguous() for t in ts]
    B, Cc, H, W = input.shape
    Cout, Ck, kh_, kw_ = weight.shape
    if (kh_, kw_) != (kh, kw):
        raise RuntimeError(f"Input shape and kernel shape wont match: ({kh} x {kw} vs {kh_} x {kw_}).")
    if Cc != Ck:
        raise RuntimeError(f"Input shape and kernel channels wont match: ({Cc} vs {Ck}).")
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    if offset.shape != (B, 2 * deformable_group * kh * kw, Ho, Wo) or mask.shape != (B, deformable_group * kh * kw, Ho, Wo):
        raise RuntimeError("dcn_v2_forward: offset/mask shape mismatch")
    out = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=input.device)
    with torch.cuda.device(input.device):
        L.check(L.lib().tdvc_dcn_v2_forward_f32(input.data_ptr(), weight.data_ptr(), bias.data_ptr(), offset.data_ptr(),
                                                mask.data_ptr(), out.data_ptr(), B, Cc, H, W, Cout, kh, kw, sh, sw, ph, pw,
                                                dh, dw, deformable_group, _stream()), "dcn_v2_forward")
    return out
