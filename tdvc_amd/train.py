"""One optimisation step of the TDVC P-frame model on MI355X (counterpart of `tools/train.py:122-159`).

    rd_loss = lambda * MSE(recon, input) + bpp_res + bpp_mv          (train.py:136-140)
    optimizer.zero_grad(); aux_optimizer.zero_grad()
    scaler.scale(rd_loss).backward(); scaler.unscale_(optimizer)
    clip_grad_norm_(model.parameters(), 2); scaler.step(optimizer); scaler.update()
    aux_loss.backward(); aux_optimizer.step()                        (train.py:142-152)

The forward and backward run in the HIP kernels (forward under `autograd.record`, backward = the tape); the loss
scale follows GradScaler's skip / halve / grow policy (the weight-gradient kernels un-scale, so there is no separate
`unscale_` pass), gradient
clipping and Adam are torch's device-side optimizer utilities, the auxiliary (quantile) loss is parameter-space
autograd.  With several ranks the parameter gradients are averaged by RCCL before clipping, so every rank clips
and steps identically; `GradBuckets` packs them into a few flat buffers.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import autograd, ops
from .synth import split_optim_params


def refresh_packed(model: torch.nn.Module) -> None:
    """after an optimizer step: every packed layer follows its (in-place updated) parameters"""
    pcs = []
    mods = model.__dict__.get("_mods_list")          # the module tree does not change between steps: walk it once
    if mods is None:
        mods = model.__dict__["_mods_list"] = list(model.modules())
    from .model.coder import GDN, gdn_refresh_batched
    pcs += gdn_refresh_batched([m for m in mods if isinstance(m, GDN)])      # effective gamma / beta of all GDN layers, re-packed with the rest
    for m in mods:
        m.__dict__.pop("_ar_cache", None)            # coder: cached wavefront-loop descriptors (their fp32 weight twins are re-created)
        if isinstance(m, GDN):
            continue
        if hasattr(m, "refresh_packed"):
            m.refresh_packed()                       # EntropyBottleneck (packed table)
            continue
        pk = m.__dict__.get("_packed", {})
        pcs += [pc for pc in pk.values() if isinstance(pc, ops.PackedConv)]
        for key in [k for k, v in pk.items() if isinstance(v, ops.PackedConvPair)]:
            del pk[key]                              # inference-only fused form (Res_Block): rebuilt from the live weights on next use
    batch = model.__dict__.get("_pack_batch")
    if batch is None or [id(pc) for pc in batch.roots] != [id(pc) for pc in pcs]:
        batch = ops.PackBatch(pcs)                   # one launch for every conv layer (forward, dgrad and column forms)
        batch.roots = pcs
        model.__dict__["_pack_batch"] = batch
    batch.run()


class GradBuckets:
    """Flat fp32 buckets over the parameter gradients: `param.grad` tensors become views into a few large buffers,
    so the data-parallel exchange is a handful of large all-reduces (xGMI rings are per-link bound: few, large
    messages) and `zero_grad` is one fill per bucket.

    Overlap with the backward sweep: `reorder(ready)` re-lays the buckets in the order the tape finishes
    the gradients (`ready[id(p)]` = index of the last backward node that writes p's gradient, `Tape.touch_log`);
    `node_done(k)` — the tape's per-node hook — then starts the async all-reduce of every bucket whose last writer was
    node k, so the exchange of the late layers runs under the backward of the early ones; `all_reduce_mean()` starts
    whatever has not been started, waits, and divides by the world size."""

    def __init__(self, params, bucket_bytes: int = 16 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.bucket_bytes = int(bucket_bytes)
        self._layout([(p, -1) for p in self.params], keep=False)

    def _layout(self, order, keep: bool):
        """order: [(param, ready index)] in bucket order; keep: carry the current gradient values over"""
        old = {id(p): p.grad for p, _ in order} if keep else {}
        self.buckets, self.ready_at, self._works = [], [], {}
        cur, cur_n = [], 0
        for p, r in order:
            if cur and (cur_n + p.numel()) * 4 > self.bucket_bytes:
                self._close(cur, cur_n, old)
                cur, cur_n = [], 0
            cur.append((p, r))
            cur_n += p.numel()
        if cur:
            self._close(cur, cur_n, old)
        self._by_ready = {}
        for i, r in enumerate(self.ready_at):
            self._by_ready.setdefault(r, []).append(i)

    def _close(self, ps, n, old):
        flat = torch.zeros(n, dtype=torch.float32, device=ps[0][0].device)
        off = 0
        for p, _ in ps:
            v = flat[off:off + p.numel()].view_as(p)
            if old.get(id(p)) is not None:
                v.copy_(old[id(p)])
            p.grad = v
            off += p.numel()
        self.buckets.append(flat)
        self.ready_at.append(max(r for _, r in ps))

    def reorder(self, ready: dict, n_nodes: int):
        """buckets in gradient-completion order; parameters the tape never touched (zero gradient) go first"""
        self.n_nodes = int(n_nodes)
        order = sorted(((p, ready.get(id(p), -1)) for p in self.params), key=lambda t: t[1])
        self._layout(order, keep=True)

    def zero(self):
        for b in self.buckets:
            b.zero_()

    # test hook (tests/test_train_dist_gpu.py): take the asynchronous all-reduce path in a process group of ONE rank, so that
    # RCCL's stream ordering against the side-stream weight-gradient kernels can be exercised on a single GPU
    force_async = False

    def _distributed(self) -> bool:
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force_async)

    def node_done(self, k: int):
        for i in self._by_ready.get(k, ()):
            if self._distributed() and i not in self._works:
                self._works[i] = dist.all_reduce(self.buckets[i], op=dist.ReduceOp.SUM, async_op=True)

    def all_reduce_mean(self):
        if not self._distributed():
            return
        world = dist.get_world_size()
        for i, b in enumerate(self.buckets):
            if i not in self._works:
                self._works[i] = dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True)
        for i, b in enumerate(self.buckets):
            self._works[i].wait()
            b.mul_(1.0 / world)
        self._works = {}


class TrainStep:
    def __init__(self, model, train_lambda: float = 2048.0, lr: float = 1e-4, loss_scale: float = 1024.0, clip: float = 2.0,
                 dynamic_scale: bool = True, growth_interval: int = 2000, graph: bool = False, graph_warmup: int = 2,
                 side_stream: bool = True):
        """graph=True: after `graph_warmup` eager steps (they build every lazily packed form) the forward + backward of
        one step is captured into a HIP graph and replayed: ~2300 launches leave the Python interpreter's critical
        path.  Input shapes are then fixed; a loss-scale change re-captures."""
        self.model = model
        self.use_graph, self.graph_warmup, self._eager_steps = bool(graph), int(graph_warmup), 0
        self._graph = None
        self._ready, self._n_nodes = None, None      # gradient-completion order, learnt on step 0
        # weight-gradient kernels run on a side stream next to the dgrad chain (autograd.Tape.off_path)
        self._side = torch.cuda.Stream() if (side_stream and torch.cuda.is_available()) else None
        self._pool = autograd.MirrorPool()            # zeroed gradient mirrors, re-zeroed off the critical path
        self.use_pool = True
        self.lam = float(train_lambda)
        self.loss_scale = float(loss_scale)
        self.clip = float(clip)
        # torch.cuda.amp.GradScaler's policy (train.py:101,146-149): skip the step and halve the scale when a gradient is
        # not finite, double it after `growth_interval` clean steps
        self.dynamic_scale, self.growth_interval, self._clean_steps = bool(dynamic_scale), int(growth_interval), 0
        main, aux = split_optim_params(model)
        named = dict(model.named_parameters())
        self.main_params = [named[n] for n in main]
        self.aux_params = [named[n] for n in aux]
        self.buckets = GradBuckets(self.main_params)
        fused = all(p.is_cuda for p in self.main_params + self.aux_params)          # one multi-tensor kernel per step
        self.optimizer = torch.optim.Adam(self.main_params, lr=lr, fused=fused)
        self.aux_optimizer = torch.optim.Adam(self.aux_params, lr=10 * lr, fused=fused)          # utils.py:110-112

    def _node_done(self, tape, k):
        if tape.n_backward_nodes == self._n_nodes and k in self.buckets._by_ready and self.buckets._distributed():
            tape.join()                                  # the side stream's weight gradients of this bucket are complete
            self.buckets.node_done(k)

    def _forward_backward(self, input_image, refer_frames, capturing: bool = False):
        """forward, loss seeds, backward: gradients accumulate into the (zeroed) buckets"""
        B, _, H, W = input_image.shape
        self.buckets.zero()
        pool = self._pool if (self.use_pool and self._side is not None and not capturing) else None
        if pool is not None:
            torch.cuda.current_stream().wait_stream(self._side)          # last step's re-zeroing of the mirrors
        with autograd.record(self.loss_scale, side_stream=self._side, pool=pool) as tape:
            if self._ready is None:
                tape.touch_log = {}                                  # first step: learn when each gradient is final
            elif not capturing:
                # later steps: all-reduce finished buckets under the rest of the sweep (same tape shape as the logged step)
                tape.on_node_done = lambda k: self._node_done(tape, k)
            recon, bpp_res, bpp_mv, _, _ = self.model(input_image, refer_frames, True)
            diff = recon - input_image.float()
            # d(lambda * MSE)/d recon, scaled; the rate terms are seeded through tape.rate_grad
            tape.grad_tensor(recon).copy_(diff * (2.0 * self.lam * self.loss_scale / diff.numel()))
            tape.rate_grad = 1.0 / float(B * H * W)
            tape.backward()
        if pool is not None:
            pool.recycle(self._side)
        if self._ready is None:
            self._ready, self._n_nodes = tape.touch_log, tape.n_backward_nodes
            self.buckets.reorder(self._ready, self._n_nodes)
        tape.release()
        # MSE itself is reduced by the caller, outside a captured graph: torch's multi-block reduction zeroes its
        # semaphores with a memset node, and on this ROCm build the first replay after other work on the stream returned
        # partial sums (tools/graph_reduce_repro.py); the gradients never depended on that scalar
        return diff, bpp_res.mean(), bpp_mv.mean()

    def _capture(self, input_image, refer_frames):
        self._static_in = (input_image.clone(), refer_frames.clone())
        self._graph = torch.cuda.CUDAGraph()
        self._graph_scale = self.loss_scale
        torch.cuda.synchronize()
        with torch.cuda.graph(self._graph):
            self._static_out = self._forward_backward(*self._static_in, capturing=True)

    def __call__(self, input_image: torch.Tensor, refer_frames: torch.Tensor) -> dict:
        model = self.model
        if not (model.training and model.mvCoder.training and model.resCoder.training):
            model.train()                              # a recursive walk over ~480 modules: only when the mode actually changes
        for p in self.aux_params:
            p.grad = None
        if self.use_graph and self._eager_steps >= self.graph_warmup:
            if (self._graph is None or self._graph_scale != self.loss_scale or self._static_in[0].shape != input_image.shape
                    or self._static_in[1].shape != refer_frames.shape):
                self._capture(input_image, refer_frames)
            self._static_in[0].copy_(input_image)
            self._static_in[1].copy_(refer_frames)
            self._graph.replay()
            diff, bpp_res, bpp_mv = self._static_out
        else:
            diff, bpp_res, bpp_mv = self._forward_backward(input_image, refer_frames)
            self._eager_steps += 1
        mse = (diff * diff).mean()
        self.buckets.all_reduce_mean()
        # clip_grad_norm_(main params, clip) on the flat buckets: the global 2-norm is also the finiteness test (after the
        # mean, so every rank agrees)
        gnorm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(b) for b in self.buckets.buckets]))
        finite = bool(torch.isfinite(gnorm))
        if finite:
            coef = torch.clamp(self.clip / (gnorm + 1e-6), max=1.0)
            for b in self.buckets.buckets:
                b.mul_(coef)
            self.optimizer.step()
            self._clean_steps += 1
            if self.dynamic_scale and self._clean_steps % self.growth_interval == 0:
                self.loss_scale *= 2.0
        else:
            gnorm = torch.tensor(float("nan"))
            if self.dynamic_scale:
                self.loss_scale *= 0.5
                self._clean_steps = 0
        aux = model.mvCoder.aux_loss() + model.resCoder.aux_loss()      # pnet.py's forward returns exactly these two
        aux.backward()
        self.aux_optimizer.step()
        refresh_packed(model)
        return dict(rd_loss=float(self.lam * mse + bpp_res + bpp_mv), mse=float(mse), bpp_res=float(bpp_res),
                    bpp_mv=float(bpp_mv), aux_loss=float(aux.detach()), grad_norm=float(gnorm), loss_scale=self.loss_scale,
                    skipped=not finite)
