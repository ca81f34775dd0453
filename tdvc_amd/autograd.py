"""Reverse-mode differentiation of the TDVC forward on MI355X (training path).

The forward is a fixed sequence of `ops.*` calls on `FM` views (channel slices of shared buffers stand in for the
reference's `torch.cat`, in-place kernels for its `x += ...`).  torch.autograd cannot follow writes into views of
pre-allocated buffers, so the build records its own tape: while `autograd.record()` is active every differentiable
op appends a closure that, given the gradient of its outputs, ACCUMULATES into the gradients of its inputs.

Gradient storage mirrors activation storage: one gradient buffer per activation buffer (same shape and dtype,
zero-initialised on first use), addressed through the same view geometry — so a conv that wrote channels 64..127
of a buffer reads its output gradient from channels 64..127 of the mirror, and the data gradient of a
`cat`-consumer lands in the producers' slices for free.  Parameter gradients are fp32 `param.grad` tensors; the
loss scale of mixed-precision training is removed inside the weight-gradient kernels (`scale = 1 / loss_scale`).
"""
from __future__ import annotations

import contextlib

import torch

from . import ops
from .ops import FM


LAZY_IDENTITY = True           # A/B switch: identity gradients of residual connections as the next dgrad conv's residual operand (Tape.add_identity)
BATCH_WGRAD_REDUCE = True        # A/B switch (tools/ab_train.py): False = one reduce launch per layer, right behind its first stage


class MirrorPool:
    """Zeroed gradient-mirror buffers that outlive a step.  A tape takes its mirrors from the pool instead of
    `torch.zeros_like` (211 fills, 1.5 ms of a 35 ms step); after the sweep `recycle()` re-zeroes them on the side stream,
    where the fills run under the optimizer and the next forward instead of on the backward's critical path.  The mirrors are
    carved out of a few large arenas, so the re-zeroing is one fill per arena (a step's ~280 mirrors of two dtypes took torch's
    per-tensor path: ~280 launches from the Python thread).  ~5 GB stay allocated at 4 x 256 x 256 (of 288 GB)."""
    ARENA_BYTES = 1 << 30

    def __init__(self):
        self.free: dict = {}
        self.used: list = []
        self.arenas: list = []                    # [uint8 buffer, bytes handed out]
        self.loose: list = []                     # mirrors that did not come from an arena (non-contiguous originals)

    def _carve(self, like: torch.Tensor) -> torch.Tensor:
        nbytes = like.numel() * like.element_size()
        if not like.is_contiguous() or nbytes == 0:
            t = torch.zeros_like(like)
            self.loose.append(t)
            return t
        need = (nbytes + 255) & ~255
        if not self.arenas or self.arenas[-1][1] + need > self.arenas[-1][0].numel():
            self.arenas.append([torch.zeros(max(self.ARENA_BYTES, need), dtype=torch.uint8, device=like.device), 0])
        ar = self.arenas[-1]
        t = ar[0][ar[1]:ar[1] + nbytes].view(like.dtype).view(like.shape)
        ar[1] += need
        return t

    def get(self, like: torch.Tensor) -> torch.Tensor:
        key = (tuple(like.shape), like.dtype, like.device)
        lst = self.free.get(key)
        t = lst.pop() if lst else self._carve(like)
        self.used.append((key, t))
        return t

    def recycle(self, side):
        """everything issued so far (main and side stream) precedes the fills; the next step waits for `side`"""
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            if self.used:
                for buf, n in self.arenas:
                    buf[:n].zero_()
                for t in self.loose:
                    t.zero_()
            for key, t in self.used:
                self.free.setdefault(key, []).append(t)
        self.used = []


class Tape:
    def __init__(self, loss_scale: float = 1.0):
        self.nodes: list = []
        self.gbuf: dict[int, torch.Tensor] = {}
        self.keep: list = []                      # activation tensors referenced by the closures stay alive
        self.nograd: set[int] = set()             # base buffers that need no gradient (network inputs)
        self.loss_scale = float(loss_scale)
        self.rate_grad = 0.0                      # dLoss/d(bits) of every rate term, set before backward() (1 / pixels for bpp)
        # data-parallel overlap (train.GradBuckets): which backward node writes a parameter's gradient last, and a hook
        # after every node so a finished bucket's all-reduce starts under the rest of the sweep
        self.touch_log: dict[int, int] | None = None
        self.on_node_done = None
        self._cur = -1
        self.n_backward_nodes = 0
        # weight / bias gradients are off the critical path of the sweep (nothing in the backward reads them): with a side
        # stream they run next to the dgrad chain and fill the CUs the small layers leave idle
        self.side = None
        self.pool: MirrorPool | None = None       # persistent zeroed mirrors (TrainStep, eager mode)
        # second stages of the weight gradients, reduced in one launch per join() instead of one per layer (ops.WgradBatch)
        self.wbatch = ops.WgradBatch() if BATCH_WGRAD_REDUCE else None
        # parameter-space chains that run as torch kernels (the entropy bottleneck's softplus / tanh chain, the GDN reparametrisation):
        # executed after the sweep AND after join(), when no MFMA kernel runs on the side stream any more -- torch's own device code is
        # built WITH packed-FP32 instructions, and its softplus kernels contain the `v_pk_*_f32 ... op_sel:[0,1]` form that returned
        # wrong values next to a concurrent MFMA stream (profiles/r04_torch_packed_fp32_scan.txt)
        self.fused_aux = False                    # the owner takes the auxiliary loss's gradient from EntropyBottleneck.loss_fused (coder.aux_loss)
        self.touched: set = set()                 # activation buffers whose gradient mirror has been handed out
        self.pending: dict = {}                   # buffer -> (view, identity gradient) not yet added to its mirror (add_identity)
        self.deferred: list = []
        self.gdn_items: list = []                 # (GDN module, d gamma_eff, d beta_eff) of this sweep: one batched chain (coder.gdn_chain_batched)
        self._ev, self._side_raw = None, None

    # ------------------------------------------------------------------ gradient views
    def _mirror(self, t: torch.Tensor) -> torch.Tensor:
        key = t.data_ptr()
        g = self.gbuf.get(key)
        if g is None:
            g = self.pool.get(t) if self.pool is not None else torch.zeros_like(t)
            self.gbuf[key] = g
            self.keep.append(t)
        return g

    def _base(self, t: torch.Tensor) -> torch.Tensor:
        """the mirror of `t`, handed out: from now on it may hold a partial sum, and an identity gradient still waiting for it is added first"""
        key = t.data_ptr()
        g = self._mirror(t)
        self.touched.add(key)
        p = self.pending.pop(key, None)
        if p is not None:
            pf, src = p
            accumulate(FM(g, pf.off, pf.N, pf.C, pf.sn), src)
        return g

    # Identity gradients of residual connections (y = conv(...) + x: grad(x) += grad(y)) were one add kernel each, 63 launches per step, in front
    # of the dgrad conv that then accumulates into the same mirror (reading it as its residual operand).  Deferred instead: when the next
    # writer of that mirror is a dgrad conv and nothing else has touched the mirror, the identity gradient IS the conv's residual operand and
    # the mirror is overwritten -- the same fp16 sum of the same two values, bit for bit, without the add and without reading zeros.
    def add_identity(self, fm: FM, src: FM):
        """grad(fm) += src, possibly later (src is final: a mirror whose producers have all run, or a temporary the tape keeps).
        -> True when the add was deferred: `src` is then still to be read and must not be overwritten (in-place act_backward)"""
        key = fm.t.data_ptr()
        if not LAZY_IDENTITY or key in self.touched or key in self.pending or src.f32 or fm.f32 or src.C != fm.C:
            accumulate(self.grad(fm), src)
            return False
        self.pending[key] = (fm, src)
        return True                               # the caller must leave `src` untouched from here on

    def grad_for_write(self, fm: FM):
        """for a kernel that can overwrite: -> (gradient view of `fm`, accumulate into it?, identity gradient to add as a residual operand or None)"""
        key = fm.t.data_ptr()
        if not LAZY_IDENTITY or key in self.touched:
            return self.grad(fm), True, None
        p = self.pending.get(key)
        if p is not None and (p[0].off, p[0].N, p[0].C, p[0].sn) != (fm.off, fm.N, fm.C, fm.sn):
            return self.grad(fm), True, None      # the waiting identity gradient covers another view of the buffer: add it the plain way
        self.pending.pop(key, None)
        self.touched.add(key)
        return FM(self._mirror(fm.t), fm.off, fm.N, fm.C, fm.sn), False, (p[1] if p is not None else None)

    def grad(self, fm: FM) -> FM:
        """the gradient view that mirrors `fm`"""
        return FM(self._base(fm.t), fm.off, fm.N, fm.C, fm.sn)

    def grad_tensor(self, t: torch.Tensor) -> torch.Tensor:
        return self._base(t)

    def zeros_like(self, t: torch.Tensor) -> torch.Tensor:
        """a zeroed scratch accumulator for this sweep (parameter-space gradients that go through a chain before they reach `.grad`): from the
        mirror pool when there is one (re-zeroed with the arenas, no fill launch of its own), else a fresh tensor"""
        if self.pool is not None and t.is_contiguous() and t.dtype in (torch.float32, torch.float16):
            return self.pool.get(t.detach())
        return torch.zeros_like(t)

    def needs_grad(self, fm: FM) -> bool:
        return fm.t.data_ptr() not in self.nograd

    def mark_input(self, fm: FM):
        self.nograd.add(fm.t.data_ptr())

    def add(self, fn):
        self.nodes.append(fn)

    def defer(self, fn):
        """run `fn` at the end of backward(), behind join(); it counts as one more backward node (index n_backward_nodes) for touch()"""
        self.deferred.append(fn)

    def off_path(self, fn, *tensors):
        """run `fn` (parameter-gradient kernels reading `tensors`) on the side stream, ordered after everything issued so far"""
        if self.side is None:
            return fn()
        # one event object per tape, re-recorded per call (a wait captures the event's state at the time of the wait call); the launches
        # inside `fn` go to the side stream by handle (ops.FORCE_STREAM) instead of through a `with torch.cuda.stream(side)` block: the
        # event allocation and the context switch were ~15 us of Python per weight-gradient launch of a step that is bound by host time
        ev = self._ev
        if ev is None:
            ev = self._ev = torch.cuda.Event()
            self._side_raw = self.side.cuda_stream
        ev.record()
        self.side.wait_event(ev)
        # the operands stay referenced until the tape dies, i.e. until after join(): the caching allocator cannot hand
        # them to the main stream meanwhile (record_stream would do too, but its deferred frees made the allocator grow
        # and stall erratically: 36 -> 60-110 ms per step in some runs); temporaries `fn` allocates (from the main stream's pool) likewise
        self.keep.extend(tensors)
        ops.FORCE_STREAM, ops.FORCE_KEEP = self._side_raw, self.keep
        try:
            fn()
        finally:
            ops.FORCE_STREAM, ops.FORCE_KEEP = None, None

    def release(self):
        """drop every reference the tape holds (closures, mirrors, kept operands, hooks): the hook closures refer back to
        the tape, so without this a finished tape — and the ~4 GB of activations it pins — waits for the cyclic GC"""
        self.nodes.clear()
        self.deferred.clear()
        self.gbuf.clear()
        self.keep.clear()
        self.pending.clear()
        self.touched.clear()
        self.on_node_done = None
        self.touch_log = None
        self.pool = None

    def join(self):
        """the main stream waits for the parameter-gradient kernels (before an all-reduce, the optimizer, the tape's end); the deferred
        reduce stages of the weight gradients collected so far go out first, on the stream their first stages ran on"""
        if self.wbatch is not None and self.wbatch.items:
            if self.side is not None:
                with torch.cuda.stream(self.side):            # the job table's host-to-device copy is a torch op: a real stream context
                    self.wbatch.flush()
            else:
                self.wbatch.flush()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def touch(self, *params):
        """the running backward node writes these parameters' gradients"""
        if self.touch_log is not None:
            for q in params:
                self.touch_log[id(q)] = self._cur

    @property
    def inv_scale(self) -> float:
        return 1.0 / self.loss_scale

    # ------------------------------------------------------------------ the backward sweep
    def backward(self):
        prev, ops._IN_BACKWARD = ops._IN_BACKWARD, True
        try:
            self.n_backward_nodes = len(self.nodes)
            for k, fn in enumerate(reversed(self.nodes)):
                self._cur = k
                fn()
                if self.on_node_done is not None:
                    self.on_node_done(k)
            self.join()
            self._cur = self.n_backward_nodes                 # the deferred chains: one virtual node behind the sweep
            for fn in self.deferred:
                fn()
        finally:
            self.join()
            ops._IN_BACKWARD = prev
            self._cur = -1
        self.nodes.clear()
        self.deferred.clear()


def param_grad(p: torch.Tensor) -> torch.Tensor:
    """fp32 gradient accumulator of a parameter (created zeroed)"""
    if p.grad is None:
        p.grad = torch.zeros_like(p, dtype=torch.float32)
    t = ops.TAPE
    if t is not None:
        t.touch(p)
    return p.grad


@contextlib.contextmanager
def record(loss_scale: float = 1.0, side_stream=None, pool: MirrorPool | None = None):
    tape = Tape(loss_scale)
    tape.side = side_stream
    tape.pool = pool
    prev, ops.TAPE = ops.TAPE, tape
    try:
        yield tape
    finally:
        ops.TAPE = prev


def accumulate(dst: FM, src: FM, sign: float = 1.0):
    """dst += sign * src (same geometry)"""
    ops.scale_act_res(dst, dst, res=src, res_sign=sign)


# ---------------------------------------------------------------------- op records (called from ops.* when taping)
def record_conv(tape: Tape, x: FM, pc, y, act, slope, res, res2, gdn, aux, square, nchw_out):
    if gdn != ops.GDN_NONE:
        return record_gdn(tape, x, pc, y, res, gdn)
    assert not square
    assert not (act not in (ops.ACT_NONE,) and res2 is not None), "autograd: activation + two residuals is not on the path"
    weight, bias = pc.param_w, pc.param_b

    def bwd():
        if nchw_out is not None:                  # planar fp32 output (the RGB reconstruction): gradient arrives planar
            g = ops.from_nchw(tape.grad_tensor(nchw_out), Cpad=ops.pad8(pc.cout))
            yv = ops.from_nchw(nchw_out, Cpad=ops.pad8(pc.cout))
        else:
            g, yv = tape.grad(y), y
        held = False                              # an identity gradient still to be read from g: the activation derivative goes to a new buffer
        if res is not None and tape.needs_grad(res):
            held |= tape.add_identity(res, g)
        if res2 is not None and tape.needs_grad(res2):
            held |= tape.add_identity(res2, g)
        if g.f32:                                 # fp32 outputs (flow, latents): the MFMA backward kernels take fp16
            g = ops.copy_cast(g, FM.empty(g.N, g.H, g.W, ops.pad8(g.C), device=g.t.device))
            held = False
        fresh = (lambda: FM.empty(g.N, g.H, g.W, g.C, dtype=g.t.dtype, device=g.t.device)) if held else (lambda: None)
        if act in (ops.ACT_RELU, ops.ACT_LRELU):
            g = ops.act_backward(g, yv, act, slope, res=res, out=fresh())
        elif act == ops.ACT_CLAMP01:
            g = ops.clamp01_backward(g, yv) if not held else ops.clamp01_backward(ops.copy_cast(g, fresh()), yv)
        elif act != ops.ACT_NONE:
            raise NotImplementedError(f"autograd: activation {act}")
        gq = ops.pixel_unshuffle(g) if pc.shuffle else g
        tape.off_path(lambda: ops.conv_wgrad(pc, gq, x, param_grad(weight).view(-1), scale=tape.inv_scale,
                                             db=param_grad(bias) if bias is not None else None,      # bias gradient from the same launch
                                             defer=tape.wbatch),                                     # its reduce stage: batched at join()
                      gq.t, x.t)
        if tape.needs_grad(x):
            gx, acc, extra = tape.grad_for_write(x)
            ops.conv_dgrad(pc, gq, gx, accumulate=acc, extra=extra)

    tape.add(bwd)


def record_scale_act_res(tape: Tape, a: FM, out: FM, gate, act, slope, res, res_sign, out2):
    def bwd():
        g = tape.grad(out)
        if out2 is not None:
            accumulate(g, tape.grad(out2))
        if res is not None and tape.needs_grad(res):
            if res_sign == 1.0:
                tape.add_identity(res, g)                 # (g is not written below: act_backward goes to a fresh buffer)
            else:
                accumulate(tape.grad(res), g, res_sign)
        if act in (ops.ACT_RELU, ops.ACT_LRELU):          # y = act(a * gate) + res: the sign of the argument is that of y - res
            g = ops.act_backward(g, out, act, slope, res=res, out=FM.empty(g.N, g.H, g.W, g.C, dtype=g.t.dtype, device=g.t.device))
        elif act != ops.ACT_NONE:
            raise NotImplementedError(f"autograd: scale_act_res with activation {act}")
        if gate is not None:
            ops.gate_backward(g, a, gate, tape.grad(a) if tape.needs_grad(a) else None, tape.grad_tensor(gate))
        elif tape.needs_grad(a):
            tape.add_identity(a, g)

    tape.add(bwd)


def record_se_gate(tape: Tape, x: FM, p, partial, nblocks, gate):
    params = p.params                             # (conv1.weight, conv1.bias, conv2.weight, conv2.bias) nn.Parameters

    def bwd():
        dgate = tape.grad_tensor(gate)
        grads = [param_grad(q) for q in params]
        dmean = ops.se_gate_backward(p, partial, nblocks, x.H * x.W, gate, dgate, tape.inv_scale, grads)
        if tape.needs_grad(x):
            ops.bcast_channel_add(tape.grad(x), dmean, 1.0 / (x.H * x.W))

    tape.add(bwd)


def record_clone(tape: Tape, x: FM, out: FM):
    def bwd():
        if tape.needs_grad(x):
            tape.add_identity(x, tape.grad(out))

    tape.add(bwd)


def record_add_flow(tape: Tape, off: FM, flow: FM):
    tape.add(lambda: ops.add_flow_backward(tape.grad(off), tape.grad(flow)))


def record_bcast_add_act(tape: Tape, x: FM, b: FM, slope):
    tape.add(lambda: ops.bcast_add_act_backward(tape.grad(x), x, tape.grad(b), slope))


def record_upsample2x(tape: Tape, x: FM, out: FM):
    def bwd():
        if tape.needs_grad(x):
            ops.upsample2x_backward(tape.grad(out), tape.grad(x))

    tape.add(bwd)


def record_resize_bilinear(tape: Tape, x: FM, out: FM, chscale):
    """flownet.py:153-173: only the flow's resize back from the x32-padded size carries a gradient (the images are inputs)"""
    def bwd():
        if tape.needs_grad(x):
            ops.resize_bilinear_backward(tape.grad(out), tape.grad(x), chscale)

    tape.add(bwd)


def record_spynet_level_input(tape: Tape, supp: FM, flow_lo, flow_up: FM, cat8: FM):
    tape.add(lambda: ops.spynet_level_input_backward(supp, flow_up, tape.grad(cat8), tape.grad(flow_up),
                                                     tape.grad(flow_lo) if flow_lo is not None else None))


def record_dcn_fused(tape: Tape, x: FM, om: FM, pc, out: FM, groups, act, slope):
    """DCNv2 backward (src/cuda/dcn_v2_cuda.cu:97-216 restructured for channel-innermost fp16): the sampled columns are
    rebuilt once (the fused forward keeps none); dW is a 1x1 weight-gradient over them, d(columns) a 1x1 conv of dY,
    and one kernel turns d(columns) into the offset / mask gradients and the scatter to the sampled features."""
    weight, bias = pc.param_w, pc.param_b

    def bwd():
        g = tape.grad(out)
        if act in (ops.ACT_RELU, ops.ACT_LRELU):
            g = ops.act_backward(g, out, act, slope)
        cpc = ops.dcn_column_conv(pc, groups)
        col = ops.dcn_columns(x, om, groups)
        ops.conv_wgrad(cpc, g, col, param_grad(weight).view(-1), scale=tape.inv_scale, db=param_grad(bias))
        dcol = ops.conv_dgrad(cpc, g, col, accumulate=False)            # overwrites the column buffer
        dx32 = ops.dcn_col2im(x, om, dcol, groups, tape.grad(om))
        if tape.needs_grad(x):
            accumulate(tape.grad(x), dx32)

    tape.add(bwd)


def record_match_gather(tape: Tape, fin: FM, fref: FM, idx, scale, cat: FM):
    tape.add(lambda: ops.match_gather_backward(fin, fref, idx, scale, tape.grad(cat), tape.grad(fin), tape.grad(fref)))


def record_gdn(tape: Tape, x: FM, pc, y: FM, res, gdn):
    """y = x * n^(-1/2) (GDN) | x * n^(+1/2) (inverse GDN) [+ res], n = beta + gamma . x^2 (a 1x1 conv on x^2).
    The fused forward keeps no `n`: it is recomputed in fp32 by the same conv kernel."""
    owner = pc.owner                              # the GDN module: maps (dgamma_eff, dbeta_eff) to its raw parameters

    def bwd():
        g = tape.grad(y)
        if res is not None and tape.needs_grad(res):
            accumulate(tape.grad(res), g)
        n32 = ops.conv(x, pc, square=True, out_dtype=torch.float32)
        dx = tape.grad(x)
        dn = ops.gdn_backward(g, x, n32, gdn == ops.GDN_INV, dx)
        t = ops.conv_dgrad(pc, dn, FM.empty(x.N, x.H, x.W, x.C, device=x.t.device), accumulate=False)
        ops.mul2_accumulate(dx, x, t)
        dgamma, dbeta = tape.zeros_like(pc.wsrc), tape.zeros_like(pc.bsrc)
        ops.conv_wgrad(pc, dn, x, dgamma.view(-1), scale=tape.inv_scale, square_x=True, db=dbeta)

        # the chain through the reparametrisation runs behind the sweep (Tape.defer), for all GDN layers of the step at once
        if not tape.gdn_items:
            def chain():
                from .model.coder import gdn_chain_batched
                gdn_chain_batched(tape.gdn_items, param_grad)      # param_grad() also logs the touch (the virtual node behind the sweep)
                tape.gdn_items = []
            tape.defer(chain)
        tape.gdn_items.append((owner, dgamma.view(dgamma.shape[0], dgamma.shape[1]), dbeta))

    tape.add(bwd)


def record_eb_forward(tape: Tape, z: FM, params: torch.Tensor, z_hat: FM, noise):
    if noise is None:
        raise NotImplementedError("autograd: the factorised prior is differentiable in training mode (additive noise) only")
    owner = params.owner                          # the EntropyBottleneck module

    def bwd():
        dz = tape.grad(z)
        accumulate(dz, tape.grad(z_hat))          # z_hat = z + noise
        dp = tape.zeros_like(params)
        ops.eb_backward(z, params, noise, tape.rate_grad * tape.loss_scale, dz, dp)

        def chain():                              # softplus / tanh chain to the raw parameters as torch kernels: behind the sweep (Tape.defer)
            owner.accumulate_param_grads(dp, tape.inv_scale)
            tape.touch(*owner.parameters())
        tape.defer(chain)

    tape.add(bwd)


def record_gc_forward(tape: Tape, y: FM, gp: FM, noise):
    if noise is None:
        raise NotImplementedError("autograd: the Gaussian conditional is differentiable in training mode (additive noise) only")
    tape.add(lambda: ops.gc_backward(y, gp, noise, tape.rate_grad * tape.loss_scale, tape.grad(y), tape.grad(gp)))


def record_quantize(tape: Tape, y: FM, out: FM, noise):
    if noise is None:
        raise NotImplementedError("autograd: rounding has no gradient; training uses additive noise")
    tape.add(lambda: accumulate(tape.grad(y), tape.grad(out)))
