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

import gc
from collections.abc import Mapping

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


class StepLog(Mapping):
    """What one TrainStep call reports: rd_loss, mse, bpp_res, bpp_mv, aux_loss, grad_norm, loss_scale (after this step's
    update), skipped.  The step itself never waits for the GPU: the seven scalars travel to pinned host memory behind the
    step's kernels and the first read of any key waits for that copy (the reference reads its losses every 10th iteration,
    tools/train.py:159-163; GradScaler.step's own host read of found_inf has gone into the fused Adam kernel)."""
    KEYS = ("rd_loss", "mse", "bpp_res", "bpp_mv", "aux_loss", "grad_norm", "loss_scale", "skipped")

    def __init__(self, owner, vals):
        self._owner, self._d, self._scale_after = owner, None, None
        if vals.is_cuda:
            self._host = torch.empty(7, dtype=torch.float32, pin_memory=True)
            self._host.copy_(vals, non_blocking=True)
            self._ev = torch.cuda.Event()
            self._ev.record()
        else:
            self._host, self._ev = vals.detach().float(), None

    def _arrived(self, block):
        if self._ev is not None:
            if not block and not self._ev.query():
                return False
            self._ev.synchronize()
            self._ev = None
        return True

    def _vals(self):
        if self._d is None:
            if self._scale_after is None:
                self._owner._settle(self)
            v = self._host.tolist()
            skipped = v[6] != 0.0
            self._d = dict(rd_loss=v[0], mse=v[1], bpp_res=v[2], bpp_mv=v[3], aux_loss=v[4],
                           grad_norm=float("nan") if skipped else v[5], loss_scale=self._scale_after, skipped=skipped)
            self._owner = None
        return self._d

    def __getitem__(self, k):
        return self._vals()[k]

    def __iter__(self):
        return iter(self.KEYS)

    def __len__(self):
        return len(self.KEYS)

    def __repr__(self):
        return f"StepLog({self._vals()!r})"


class TrainStep:
    def __init__(self, model, train_lambda: float = 2048.0, lr: float = 1e-4, loss_scale: float = 1024.0, clip: float = 2.0,
                 dynamic_scale: bool = True, growth_interval: int = 2000, graph: bool = False, graph_warmup: int = 2,
                 side_stream: bool = True, scale_update: str = "exact", freeze_gc: bool = True):
        """freeze_gc: after the second step (model, packed weights, descriptor caches and pools exist by then; bench.py's three warm-up steps
        include the collection) everything alive
        moves to the garbage collector's permanent generation (gc.freeze): a step allocates ~50 k short-lived containers (tape
        closures, descriptors), which triggers full collections, and each of those walked the whole heap of the process --
        40-75 ms once or twice per 30 steps (+2.3 ms per step on average, tools/train_back_to_back.py).
        scale_update: "exact" = GradScaler.update()'s timing: a step that overflowed halves the scale before the NEXT
        forward (the host waits, at the start of a step, for the previous step's flag -- by then the GPU still holds that
        step's optimizer / refresh kernels, so the wait does not drain it); "lagged" = never wait: the update lands on
        the first step that starts after the flag has arrived (one or two steps late; the overflowed steps in between are
        still skipped exactly, by the flag on the device).  With dynamic_scale=False nothing ever waits.
        graph=True: after `graph_warmup` eager steps (they build every lazily packed form) the forward + backward of
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
        assert scale_update in ("exact", "lagged")
        self.scale_update, self._pending = scale_update, []
        self.freeze_gc, self._calls = bool(freeze_gc), 0
        main, aux = split_optim_params(model)
        named = dict(model.named_parameters())
        self.main_params = [named[n] for n in main]
        self.aux_params = [named[n] for n in aux]
        self.buckets = GradBuckets(self.main_params)
        fused = all(p.is_cuda for p in self.main_params + self.aux_params)          # one multi-tensor kernel per step
        self._fused = fused
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
        with autograd.record(self.loss_scale, side_stream=self._side, pool=pool) as tape:
            tape.fused_aux = self._fused                             # the forward's aux_loss() values: no torch graph needed (loss_fused below)
            if self._ready is None:
                tape.touch_log = {}                                  # first step: learn when each gradient is final
            elif not capturing:
                # later steps: all-reduce finished buckets under the rest of the sweep (same tape shape as the logged step)
                tape.on_node_done = lambda k: self._node_done(tape, k)
            recon, bpp_res, bpp_mv, _, _ = self.model(input_image, refer_frames, True)
            diff = recon - input_image.float()
            if pool is not None:
                # last step's re-zeroing of the mirrors (5 GB of fills on the side stream) has the optimizer, the re-packing and this
                # forward to finish under: nothing before this line writes a mirror
                torch.cuda.current_stream().wait_stream(self._side)
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
        self._calls += 1
        if self._calls == 3 and self.freeze_gc:
            gc.collect()
            gc.freeze()
        self._settle(block=self.dynamic_scale and self.scale_update == "exact")
        if not (model.training and model.mvCoder.training and model.resCoder.training):
            model.train()                              # a recursive walk over ~480 modules: only when the mode actually changes
        if not self._fused:
            for p in self.aux_params:
                p.grad = None                          # (the fused auxiliary loss overwrites its gradients)
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
        coef = torch.clamp(self.clip / (gnorm + 1e-6), max=1.0)
        if self._fused:
            # no host round trip: the fused Adam kernel itself skips the update (and its step counter) when `found_inf` is
            # set -- what GradScaler.step does for a fused optimizer; the host learns about it from the StepLog's copy
            found = torch.logical_not(torch.isfinite(gnorm)).float()
            torch._foreach_mul_(self.buckets.buckets, coef)
            self.optimizer.found_inf = found
            self.optimizer.step()
        else:
            found = torch.logical_not(torch.isfinite(gnorm)).float()
            if not bool(found):
                for b in self.buckets.buckets:
                    b.mul_(coef)
                self.optimizer.step()
        if self._fused:
            # the auxiliary quantile loss of both coders and its gradient: one launch each, on the forward's parameter values (the packed
            # tables are re-packed below) -- what `aux_loss.backward()` differentiates in tools/train.py:150, whose graph dates from the forward
            a1, a2 = model.mvCoder.entropy_bottleneck.loss_fused(), model.resCoder.entropy_bottleneck.loss_fused()
            aux = (a1 + a2).reshape(())
        else:
            aux = model.mvCoder.aux_loss() + model.resCoder.aux_loss()      # pnet.py's forward returns exactly these two
            aux.backward()
        self.aux_optimizer.step()
        refresh_packed(model)
        vals = torch.stack([self.lam * mse + bpp_res + bpp_mv, mse, bpp_res, bpp_mv, aux.detach().float(), gnorm, found])
        log = StepLog(self, vals)
        self._pending.append(log)
        if not vals.is_cuda:
            self._settle(log)
        return log

    def _settle(self, upto=None, block: bool = True):
        """apply GradScaler.update()'s bookkeeping for the steps whose finiteness flag has reached the host, oldest first:
        up to `upto` (all of them when None); block=False stops at the first step the GPU has not finished"""
        while self._pending:
            log = self._pending[0]
            if not log._arrived(block):
                return
            self._pending.pop(0)
            if log._host[6] == 0.0:
                self._clean_steps += 1
                if self.dynamic_scale and self._clean_steps % self.growth_interval == 0:
                    self.loss_scale *= 2.0
            elif self.dynamic_scale:
                self.loss_scale *= 0.5
                self._clean_steps = 0
            log._scale_after = self.loss_scale
            if log is upto:
                return
