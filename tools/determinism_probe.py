"""Are two identical TrainStep runs bitwise identical?  Runs the tests' operating-point training recipe twice from the same
seed for a few steps and lists the parameters whose first-step gradients / final values differ.
python tools/determinism_probe.py [steps] [side_stream 0|1] [pool 0|1] [mode]
mode "sync": the side-stream work is followed by a device synchronisation (a race between the streams disappears, an
             uninitialised read does not);
mode "poison" / "poison1": every FM.empty buffer is filled with NaN / with 1.0 (a read of memory nothing wrote shows up as NaN
             or as a changed result)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops  # noqa: E402
from tdvc_amd.model import VideoCompressor  # noqa: E402
from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402
from tdvc_amd.train import TrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
side = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
use_pool = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
mode = sys.argv[4] if len(sys.argv) > 4 else ""
if mode == "sync":
    from tdvc_amd import autograd
    _off = autograd.Tape.off_path

    def off_sync(self, fn, *tensors):
        _off(self, fn, *tensors)
        torch.cuda.synchronize()
    autograd.Tape.off_path = off_sync
elif mode in ("spynet_main", "spynet_side"):
    # the SPyNet weight gradients (7x7 convs: (Cin, dY channels) in the set below) on the main stream and the rest on the side
    # stream, or the other way round
    from tdvc_amd import autograd
    _off = autograd.Tape.off_path
    SPY = {(8, 32), (32, 64), (64, 32), (32, 16), (16, 8)}

    def off_sel(self, fn, *tensors):
        spy = (tensors[1].shape[-1], tensors[0].shape[-1]) in SPY
        if spy == (mode == "spynet_main"):
            return fn()
        return _off(self, fn, *tensors)
    autograd.Tape.off_path = off_sel
elif mode == "mirrors":
    # no kernel is added to the sweep: every gradient mirror and every kept operand is still alive when the tape is released
    # (after the join); their checksums, in the order the sweep first touched them, show where two runs part
    from tdvc_amd import autograd
    MIR = {"on": False, "log": []}
    _rel = autograd.Tape.release

    def bits(t):
        t = t.contiguous()
        return int(t.view(torch.int32 if t.dtype == torch.float32 else torch.int16).sum(dtype=torch.int64))

    def release(self):
        if MIR["on"]:
            torch.cuda.synchronize()
            for i, g in enumerate(self.gbuf.values()):
                MIR["log"].append((f"mirror {i}", tuple(g.shape), str(g.dtype), bits(g)))
            for i, t in enumerate(self.keep):
                if torch.is_tensor(t) and t.dtype in (torch.float16, torch.float32):
                    MIR["log"].append((f"kept {i}", tuple(t.shape), str(t.dtype), bits(t)))
        return _rel(self)
    autograd.Tape.release = release
elif mode == "trace":
    # checksum (sum of bit patterns: exact, order-free) of what the main-stream backward ops of step 1 produce; the first
    # entry that differs between the two runs names the operator whose result depends on timing
    TRACE = {"on": False, "log": []}

    def bits(t):
        t = t.contiguous()
        return t.view(torch.int32 if t.dtype == torch.float32 else torch.int16).sum(dtype=torch.int64)

    def wrap(name, pick):
        orig = getattr(ops, name)

        def f(*a, **k):
            if TRACE["on"] and ops._IN_BACKWARD:
                pre = [(nm, tuple(t.t.shape), bits(t.t)) for nm, t in pick(a, k, None) if t is not None]
            r = orig(*a, **k)
            if TRACE["on"] and ops._IN_BACKWARD:
                for nm, shp, c in pre:
                    TRACE["log"].append((name + ":in:" + nm, shp, c))
                for nm, t in pick(a, k, r):
                    if t is not None:
                        TRACE["log"].append((name + ":out:" + nm, tuple(t.t.shape), bits(t.t)))
            return r
        setattr(ops, name, f)
    wrap("conv_dgrad", lambda a, k, r: [("g", a[1]), ("dx", a[2])])
    wrap("act_backward", lambda a, k, r: [("g", a[0])] if r is None else [("out", r)])
    wrap("copy_cast", lambda a, k, r: [("src", a[0])] if r is None else [("dst", r)])
    wrap("spynet_level_input_backward", lambda a, k, r: [("dcat8", a[2]), ("dflow_up", a[3]), ("dflow_lo", a[4])])
    wrap("add_flow_backward", lambda a, k, r: [("doff", a[0]), ("dflow", a[1])])
    wrap("scale_act_res", lambda a, k, r: [("a", a[0]), ("res", k.get("res"))] if r is None else [("out", a[1])])
elif mode.startswith("no_"):                   # no_n16 / no_c8 / no_gdn128 / no_dcn_lds: switch one kernel family off
    import ctypes
    from tdvc_amd import _lib
    name = {"no_n16": "tdvc_debug_enable_conv_n16", "no_c8": "tdvc_debug_enable_conv_c8", "no_gdn128": "tdvc_debug_enable_gdn128",
            "no_dcn_lds": "tdvc_debug_enable_dcn_lds"}[mode]
    fn = getattr(_lib.lib(), name)
    fn.argtypes = [ctypes.c_int]
    fn.restype = None
    fn(0)
elif mode.startswith("poison"):
    fillv = 1.0 if mode == "poison1" else float("nan")

    def empty_poison(N, H, W, C_, dtype=torch.float16, device="cuda"):
        return ops.FM(torch.full((N, H, W, C_), fillv, dtype=dtype, device=device))
    ops.FM.empty = staticmethod(empty_poison)


def batches(n):
    pool, cursor, out = [], 0, []
    while len(out) < n:
        while len(pool) < 4:
            gop = make_gop(5000 + cursor, 7, 256, 256)
            cursor += 1
            for t in range(1, 7):
                pool.append((gop[t:t + 1], ref_list([gop[k:k + 1] for k in range(0, t)][-4:] if t > 3 else [gop[k:k + 1] for k in range(0, t)])))
        b, pool = pool[:4], pool[4:]
        out.append((torch.cat([q[0] for q in b]).cuda(), torch.cat([q[1] for q in b]).cuda()))
    return out


def run(data):
    torch.manual_seed(1111)
    ops.DETERMINISTIC = True
    net = VideoCompressor()
    fill_parameters(net)
    net = net.cuda().train()
    step = TrainStep(net, train_lambda=256.0, lr=2e-4, loss_scale=128.0, side_stream=side)
    step.use_pool = use_pool
    grads, states, logs = [], [], []
    for it, (x, refs) in enumerate(data):
        if mode == "trace":
            TRACE["on"] = it == 1
        if mode == "mirrors":
            MIR["on"] = it == 1
        logs.append(step(x, refs))
        torch.cuda.synchronize()
        grads.append({n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
        states.append({k: v.detach().clone() for k, v in net.state_dict().items()})
    if mode == "mirrors":
        MIR["on"] = False
        logs.append(list(MIR["log"]))
        MIR["log"] = []
    if mode == "trace":
        TRACE["on"] = False
        logs.append([(n, shp, int(c)) for n, shp, c in TRACE["log"]])
        TRACE["log"] = []
    return grads, states, logs


data = batches(steps)
ga, sa, la = run(data)
gb, sb, lb = run(data)
print(f"side_stream={side} pool={use_pool} steps={steps} mode={mode!r}")
if mode == "mirrors":
    ta, tb = la.pop(), lb.pop()
    bad = [i for i, (ea, eb) in enumerate(zip(ta, tb)) if ea != eb]
    print(f" mirrors / kept operands of step 1: {len(ta)} / {len(tb)} entries, {len(bad)} differ")
    for i in bad[:40]:
        print(f"   entry {i}: {ta[i][0]} {ta[i][1]} {ta[i][2]}: {ta[i][3]} vs {tb[i][3]}")
    if bad:
        for j in range(max(0, bad[0] - 6), bad[0]):
            print(f"   (before) entry {j}: {ta[j][0]} {ta[j][1]} {ta[j][2]}")
if mode == "trace":
    ta, tb = la.pop(), lb.pop()
    print(f" trace of step 1: {len(ta)} / {len(tb)} entries")
    shown = 0
    for i, (ea, eb) in enumerate(zip(ta, tb)):
        if ea != eb:
            print(f"   entry {i}: {ea[0]} {ea[1]}: {ea[2]} vs {eb[2]}" + ("" if ea[:2] == eb[:2] else f"  (B: {eb[0]} {eb[1]})"))
            shown += 1
            if shown >= 12:
                break
    if not shown:
        print("   all entries equal")
    else:
        first = next(i for i, (ea, eb) in enumerate(zip(ta, tb)) if ea != eb)
        for j in range(max(0, first - 8), first):
            print(f"   (before) entry {j}: {ta[j][0]} {ta[j][1]}")
for it in range(steps):
    bad_g = [n for n in ga[it] if not torch.equal(ga[it][n], gb[it][n])]
    bad_s = [n for n in sa[it] if not torch.equal(sa[it][n], sb[it][n])]
    print(f" step {it}: gradients (after clipping) differ in {len(bad_g)}/{len(ga[it])} tensors, state after the step in {len(bad_s)}/{len(sa[it])}, "
          f"log equal {la[it] == lb[it]}: rd_loss {la[it]['rd_loss']!r} vs {lb[it]['rd_loss']!r}, grad_norm {la[it]['grad_norm']!r} vs {lb[it]['grad_norm']!r}")
    nf = [n for n in ga[it] if not bool(torch.isfinite(ga[it][n]).all())]
    if nf:
        print(f"     non-finite gradients in {len(nf)} tensors: {nf[:10]}")
    if bad_g or bad_s:
        mods = sorted({".".join(n.split(".")[:4]) for n in bad_g})
        print(f"     modules with differing gradients: {mods}")
        for n in bad_g[:12]:
            d = (ga[it][n].float() - gb[it][n].float()).abs()
            print(f"     grad {n}: {int((d > 0).sum())}/{d.numel()} elements, max |d| {float(d.max()):.3e} (max |g| {float(ga[it][n].float().abs().max()):.3e})")
        for n in bad_s[:12]:
            d = (sa[it][n].float() - sb[it][n].float()).abs()
            print(f"     state {n}: {int((d > 0).sum())}/{d.numel()} elements, max |d| {float(d.max()):.3e}")
        break
