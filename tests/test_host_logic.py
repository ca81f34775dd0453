"""CPU: host-side logic of the product (no GPU compute)."""
import numpy as np
import pytest
import torch

from tdvc_amd import ops
from tdvc_amd.synth import fill_parameters, hash_uniform, make_gop, ref_list, split_optim_params


def test_filler_is_a_pure_function_of_name_and_index():
    u = hash_uniform(5, 12345)
    np.testing.assert_allclose(u, hash_uniform(5, 12345))
    assert 0.0 <= u.min() and u.max() < 1.0 and len(set(np.round(u, 9))) == 5
    a, b = torch.nn.Conv2d(8, 8, 3), torch.nn.Conv2d(8, 8, 3)
    ha, hb = torch.nn.Module(), torch.nn.Module()
    ha.add_module("x", a); hb.add_module("x", b)
    fill_parameters(ha); fill_parameters(hb)
    assert torch.equal(a.weight, b.weight) and torch.equal(a.bias, b.bias)
    assert abs(float(a.weight.detach().mean())) < 0.02 and 0.05 < float(a.weight.detach().std()) < 0.2


def test_make_gop_and_ref_list():
    g = make_gop(1234, 7, 64, 96)
    assert g.shape == (7, 3, 64, 96) and g.dtype == torch.float32
    assert torch.equal(g, make_gop(1234, 7, 64, 96)) and not torch.equal(g, make_gop(1235, 7, 64, 96))
    q = g * 255.0
    assert float((q - q.round()).abs().max()) < 1e-4 and 0.0 <= float(g.min()) and float(g.max()) <= 1.0
    # frame t is frame 0 shifted by (t, 2t) px up to noise
    d = (g[1, :, :-1, :-2] - g[0, :, 1:, 2:]).abs().mean()
    assert float(d) < 0.03
    I, r1, r2, r3 = [torch.full((1, 3, 2, 2), float(v)) for v in range(4)]
    f = lambda refs: ref_list(refs)[0, :, 0, 0, 0].tolist()
    assert f([I]) == [0, 0, 0, 0]
    assert f([I, r1]) == [0, 0, 1, 1]
    assert f([I, r1, r2]) == [0, 0, 1, 2]
    assert f([I, r1, r2, r3]) == [0, 1, 2, 3]


def test_state_dict_is_reference_compatible():
    from oracle.tdvc_ref import VideoCompressor as Ref
    from tdvc_amd.model import VideoCompressor
    m, r = VideoCompressor(), Ref()
    a, b = m.state_dict(), r.state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(a[k].shape == b[k].shape for k in a)
    m.load_state_dict(b, strict=True)
    main, aux = split_optim_params(m)
    assert len(aux) == 2 and len(main) + len(aux) == len(list(m.parameters()))
    for k in ("motion_est.spynet.basic_module.5.basic_module.4.conv.weight", "mcnet.dconv.conv_offset_mask.bias",
              "mvCoder.entropy_bottleneck._matrix0", "resCoder.g_a.0.gdn.beta_reparam.lower_bound.bound",
              "mcfilter.layer1.temporal_conv3d.weight", "loopfilter.conv_13.weight", "motion_est.offset_conv12.l1.weight"):
        assert k in a


def test_fm_views_address_arithmetic():
    t = torch.zeros(2, 4, 6, 256, dtype=torch.float16)
    f = ops.FM(t)
    base = t.data_ptr()
    d = f.ch(64, 64).desc()
    assert d.p == base + 64 * 2 and d.C == 64 and d.sp == 256 and d.sn == 4 * 6 * 256
    d = f.batch(1, 1).ch(128, 128).desc()
    assert d.p == base + (4 * 6 * 256 + 128) * 2 and d.N == 1
    s = f.as_slices(1, 4, 64)
    d = s.desc()
    assert d.p == base + 4 * 6 * 256 * 2 and d.N == 4 and d.sn == 64 and d.C == 64 and d.sp == 256
    d = s.batch(1, 3).desc()
    assert d.p == base + (4 * 6 * 256 + 64) * 2 and d.N == 3
    # size-1 dimensions carry arbitrary torch strides (e.g. after permute): FM must not trust them
    odd = torch.zeros(1, 128, 1, 1, dtype=torch.float16).permute(0, 2, 3, 1)
    d = ops.FM(odd).desc()
    assert d.sp == 128 and d.sn == 128 and d.C == 128
    nar = ops.FM(torch.zeros(1, 1, 8, 64, dtype=torch.float16)[:, :, :3])
    assert nar.W == 3 and nar.desc().sp == 64
    f32 = ops.FM(torch.zeros(1, 2, 2, 4))
    assert f32.f32 and f32.desc().dtype == 1


def _emulate_indexed_pack(w_flat, tb):
    """numpy restatement of pack_indexed_kernel: (row, channel, tap) -> value"""
    ro, co_, to = tb.row_off.numpy(), tb.chan_off.numpy(), tb.tap_off.numpy()
    rm, cm, tm = tb.row_mask.numpy(), tb.chan_mask.numpy(), tb.tap_mask.numpy()
    out = np.zeros((tb.cout, tb.cin, len(tb.taps)), dtype=np.float32)
    for r in range(tb.cout):
        for t in range(len(tb.taps)):
            if ro[r] < 0 or (tm[t] & rm[r]):
                continue
            ok = (co_ >= 0) & ((tm[t] & cm) == 0)
            out[r, ok, t] = w_flat[ro[r] + co_[ok] + to[t]]
    return out


def _dense_rct(w, taps):
    """(cout, cin, kh, kw) -> (cout, cin, ntaps) in tap-list order"""
    return np.stack([w[:, :, dy, dx] for dy, dx in taps], axis=2)


def test_pack_tables_forward_forms():
    from tdvc_amd import convpack as cp
    rng = np.random.default_rng(0)
    w = rng.standard_normal((16, 6, 3, 3)).astype(np.float32)
    full = [(dy, dx) for dy in range(3) for dx in range(3)]
    lay = cp.WeightLayout.dense(16, 6, 3, 3)
    # zero-padded channels
    tb = cp.forward_tables(lay, cin_pad=8, taps=full, pad=1, ck=8, device="cpu")
    got = _emulate_indexed_pack(w.reshape(-1), tb)
    assert np.array_equal(got[:, :6], _dense_rct(w, full)) and not got[:, 6:].any()
    assert tb.tap_lin.tolist() == list(range(9))
    # masked tap list + channel permutation (concatenation order)
    taps = [(0, 0), (0, 1), (1, 0)]
    perm = [3, 4, 5, 0, 1, 2]
    tb = cp.forward_tables(lay, cin_pad=8, taps=taps, pad=1, ck=8, cin_perm=perm, device="cpu")
    assert np.array_equal(_emulate_indexed_pack(w.reshape(-1), tb)[:, :6], _dense_rct(w[:, perm], taps))
    # PixelShuffle row order
    w4 = rng.standard_normal((8, 4, 3, 3)).astype(np.float32)
    tb = cp.forward_tables(cp.WeightLayout.dense(8, 4, 3, 3), cin_pad=8, taps=full, pad=1, ck=8, shuffle=True, device="cpu")
    rows = torch.arange(8).view(2, 4).t().reshape(-1).numpy()
    assert np.array_equal(_emulate_indexed_pack(w4.reshape(-1), tb)[:, :4], _dense_rct(w4[rows], full))
    # space-to-depth form of a 3x3 stride-2 conv == ops._s2d_weights
    ws = rng.standard_normal((64, 32, 3, 3)).astype(np.float32)
    tb = cp.forward_tables_s2d(cp.WeightLayout.dense(64, 32, 3, 3), ck=32, device="cpu")
    ref = ops._s2d_weights(torch.from_numpy(ws)).numpy()
    assert (tb.kh, tb.kw, tb.pad, tb.cin) == (2, 2, 1, 128)
    assert np.array_equal(_emulate_indexed_pack(ws.reshape(-1), tb), _dense_rct(ref, tb.taps))
    # Conv3d (3,1,1) holder gathered from the 5-D parameter
    w5 = rng.standard_normal((8, 4, 3, 1, 1)).astype(np.float32)
    lay5 = cp.WeightLayout.conv3d_temporal(8, 4, 3)
    tb = cp.forward_tables(lay5, cin_pad=16, taps=[(0, 0)], pad=0, ck=8, device="cpu")
    ref5 = np.transpose(w5, (0, 2, 1, 3, 4)).reshape(8, 12, 1, 1)
    assert np.array_equal(_emulate_indexed_pack(w5.reshape(-1), tb)[:, :12], _dense_rct(ref5, [(0, 0)]))


def test_pack_tables_dgrad_forms():
    """the data-gradient conv built from the tables == autograd's input gradient (dense evaluation on the CPU)"""
    import torch.nn.functional as F
    from tdvc_amd import convpack as cp
    g = torch.Generator().manual_seed(3)

    def run_packed(tb, w_flat, gy):
        """evaluate the conv the tables describe: input gy (N, tb.cin, H, W) -> (N, tb.cout, H, W) (+ PixelShuffle)"""
        wd = torch.from_numpy(_emulate_indexed_pack(w_flat, tb))              # (rows, chans, ntaps)
        wk = torch.zeros(tb.cout, tb.cin, tb.kh, tb.kw)
        for t, (dy, dx) in enumerate(tb.taps):
            wk[:, :, dy, dx] = wd[:, :, t]
        y = F.conv2d(gy, wk, None, padding=tb.pad)
        if tb.shuffle:                          # packed row (i*2+j)*cq + c -> pixel (2y+i, 2x+j), channel c
            N, C4, H, W = y.shape
            y = y.view(N, 2, 2, C4 // 4, H, W).permute(0, 3, 4, 1, 5, 2).reshape(N, C4 // 4, 2 * H, 2 * W)
        return y

    for (cout, cin, k, stride, pad, H, W) in [(8, 6, 3, 1, 1, 7, 9), (8, 6, 1, 1, 0, 5, 6), (16, 8, 3, 2, 1, 8, 12), (16, 8, 1, 2, 0, 8, 10),
                                              (8, 4, 7, 1, 3, 9, 9)]:
        w = torch.randn(cout, cin, k, k, generator=g)
        x = torch.randn(2, cin, H, W, generator=g, requires_grad=True)
        y = F.conv2d(x, w, None, stride=stride, padding=pad)
        gy = torch.randn(y.shape, generator=g)
        (gx,) = torch.autograd.grad(y, x, gy)
        taps = [(dy, dx) for dy in range(k) for dx in range(k)]
        tb = cp.dgrad_tables(cp.WeightLayout.dense(cout, cin, k, k), g_channels=cout, x_channels=cin, taps=taps, pad=pad, stride=stride,
                             ck=8, device="cpu")
        got = run_packed(tb, w.numpy().reshape(-1), gy)
        assert got.shape == gx.shape, (got.shape, gx.shape)
        assert torch.allclose(got, gx, atol=1e-4, rtol=1e-4), f"dgrad tables k={k} stride={stride}: {float((got - gx).abs().max())}"
    # sub-pixel forward conv (conv + PixelShuffle): dX = dgrad conv over the un-shuffled dY in packed-row order
    cout, cin, H, W = 16, 6, 5, 7
    w = torch.randn(cout, cin, 3, 3, generator=g)
    x = torch.randn(1, cin, H, W, generator=g, requires_grad=True)
    y = F.pixel_shuffle(F.conv2d(x, w, None, padding=1), 2)
    gy = torch.randn(y.shape, generator=g)
    (gx,) = torch.autograd.grad(y, x, gy)
    taps = [(dy, dx) for dy in range(3) for dx in range(3)]
    tb = cp.dgrad_tables(cp.WeightLayout.dense(cout, cin, 3, 3), g_channels=cout, x_channels=cin, taps=taps, pad=1, stride=1, ck=8,
                         shuffle=True, device="cpu")
    cq = cout // 4
    gun = gy.view(1, cq, H, 2, W, 2).permute(0, 3, 5, 1, 2, 4).reshape(1, cout, H, W)      # channel (i*2+j)*cq + c
    assert torch.allclose(run_packed(tb, w.numpy().reshape(-1), gun), gx, atol=1e-4, rtol=1e-4)


def test_masked_conv_taps():
    from tdvc_amd.model.coder import MaskedConv2d
    mc = MaskedConv2d(8, 16, kernel_size=5, padding=2)
    taps = mc.live_taps()
    assert len(taps) == 12 and all(mc.mask[0, 0, dy, dx] == 1 for dy, dx in taps) and int(mc.mask[0, 0].sum()) == 12


def test_pad_crop_match_oracle():
    from oracle.tdvc_ref.blocks import crop_to, pad_to
    from tdvc_amd.codec_utils import crop, pad
    x = torch.randn(2, 3, 1080, 1920)
    p = pad(x, 64)
    assert p.shape[-2:] == (1088, 1920) and torch.equal(p, pad_to(x, 64))
    assert torch.equal(p[..., :4, :], torch.zeros(2, 3, 4, 1920)) and torch.equal(p[..., -4:, :], torch.zeros(2, 3, 4, 1920))
    assert torch.equal(crop(p, (1080, 1920)), x) and torch.equal(crop(p, (1080, 1920)), crop_to(p, (1080, 1920)))
    y = torch.randn(1, 3, 59, 67)
    assert torch.equal(pad(y, 64), pad_to(y, 64)) and torch.equal(crop(pad(y, 64), (59, 67)), y)


def test_training_sampler_matches_reference_rule():
    """dataset.py:211-247: input im{t}, references [im1, t-3 .. t-1] padded by repeating the last, plus im7 <- [1,1,3,5]"""
    from tdvc_amd.tools.train import septuplet_samples
    frames = torch.arange(7).float().view(7, 1, 1, 1).expand(7, 3, 2, 2).contiguous() + 1.0      # frame value = 1-based index
    got = [(int(x[0, 0, 0]), [int(r[0, 0, 0]) for r in refs]) for x, refs in septuplet_samples(frames)]
    assert got == [(2, [1, 1, 1, 1]), (3, [1, 1, 2, 2]), (4, [1, 1, 2, 3]), (5, [1, 2, 3, 4]), (6, [1, 3, 4, 5]), (7, [1, 4, 5, 6]),
                   (7, [1, 1, 3, 5])]


def test_bitstream_container_layout():
    """record layout of tools/utils/encoder.py:61-68: `>4I` shape, native uint16 byte count, payload; long-payload escape"""
    import io
    import struct

    from tdvc_amd import bitstream
    strings = [b"abc", b"", bytes(range(256)) * 3]
    shapes = [(0, 128, 4, 6), (0, 128, 1, 2), (1, 128, 17, 30)]
    buf = io.BytesIO()
    n = bitstream.write_records(buf, strings, shapes)
    raw = buf.getvalue()
    assert n == len(raw) == sum(16 + 2 + len(s) for s in strings)
    assert struct.unpack(">4I", raw[:16]) == shapes[0] and int(np.frombuffer(raw[16:18], dtype=np.uint16)[0]) == 3 and raw[18:21] == b"abc"
    buf.seek(0)
    s2, sh2 = bitstream.read_records(buf, 3)
    assert s2 == strings and [tuple(s) for s in sh2] == shapes
    big = io.BytesIO()
    bitstream.write_records(big, [b"x" * 65535], [(1, 2, 3, 4)])
    assert len(big.getvalue()) == 16 + 2 + 4 + 65535
    big.seek(0)
    assert bitstream.read_records(big, 1)[0][0] == b"x" * 65535
    with pytest.raises(ValueError):
        bitstream.read_records(io.BytesIO(raw[:-1]), 3)


def test_checkpoint_with_filled_coder_tables_loads_strict():
    """a checkpoint saved after `update()` carries the CDF buffers filled; like compressai's load_state_dict the module
    resizes its (empty) buffers, so `load_state_dict(strict=True)` (tools/predict.py:150) works both ways"""
    from oracle.tdvc_ref import coder as oc
    from tdvc_amd.model import coder as dc
    a = oc.MVCoder(N=128)
    a.update(force=True)
    sd = a.state_dict()
    assert sd["gaussian_conditional._quantized_cdf"].numel() > 0
    for cls in (oc.MVCoder, dc.MVCoder):
        b = cls(N=128)
        b.load_state_dict(sd, strict=True)
        assert torch.equal(b.gaussian_conditional._quantized_cdf, sd["gaussian_conditional._quantized_cdf"])
        assert torch.equal(b.entropy_bottleneck._offset, sd["entropy_bottleneck._offset"])
        c = cls(N=128)                                     # and back: an un-updated checkpoint into an updated module
        b.load_state_dict(c.state_dict(), strict=True)
        assert b.gaussian_conditional._offset.numel() == 0


def test_bench_self_launch_forwards_arguments(monkeypatch):
    """`python bench.py --gpus 2` without a launcher: the ranks are started as fresh children under torch.distributed.run
    (one per GPU, 127.0.0.1 rendezvous) with the caller's arguments, and the launcher's exit code is the process's."""
    import importlib.util
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--mode", "train"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7                                # a failing child makes bench.py fail
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1", "--mode", "train"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher whose world size disagrees with --gpus the run refuses instead of measuring something else
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert "WORLD_SIZE=4" in str(ex.value.code)


def test_tape_deferred_chains_run_behind_the_sweep():
    """`Tape.defer`: parameter-space chains that run as torch kernels (entropy-bottleneck softplus / tanh chain, GDN reparametrisation) are
    executed after the last backward node and after join() -- no MFMA kernel of the side stream is in flight then (torch's softplus kernels
    contain the packed-FP32 form of profiles/r03_packed_fp32_race.txt) -- and count as one more node for the bucket bookkeeping."""
    from tdvc_amd import autograd
    tape = autograd.Tape()
    log = []
    tape.touch_log = {}
    p1, p2 = object(), object()
    tape.join = lambda: log.append("join")
    tape.add(lambda: (log.append("node_a"), tape.touch(p1)))                 # recorded first: runs LAST in the sweep

    def node_b():
        log.append("node_b")
        tape.defer(lambda: (log.append("chain"), tape.touch(p2)))
    tape.add(node_b)
    tape.on_node_done = lambda k: log.append(f"done{k}")
    tape.backward()
    assert log == ["node_b", "done0", "node_a", "done1", "join", "chain", "join"], log
    assert tape.touch_log[id(p1)] == 1 and tape.touch_log[id(p2)] == 2 == tape.n_backward_nodes
    assert not tape.deferred and not tape.nodes


def test_step_log_is_lazy_and_settles_the_loss_scale_in_order():
    """train.StepLog / TrainStep._settle (host logic only, CPU tensors): GradScaler.update()'s bookkeeping -- halve on a non-finite
    gradient norm, double after `growth_interval` clean steps -- is applied to the pending steps oldest first, exactly once, whether the
    log is read or the next step settles it; a skipped step reports grad_norm NaN and the scale AFTER its update"""
    from tdvc_amd.train import StepLog, TrainStep
    ts = TrainStep.__new__(TrainStep)
    ts.dynamic_scale, ts.growth_interval, ts._clean_steps, ts.loss_scale, ts._pending = True, 2, 0, 64.0, []

    def step(found):
        log = StepLog(ts, torch.tensor([9.5, 0.1, 1.0, 2.0, 30.0, 4.0, float(found)]))
        ts._pending.append(log)
        return log

    a, b, c = step(0), step(1), step(0)
    assert ts.loss_scale == 64.0 and len(ts._pending) == 3             # nothing settled yet: nobody read a log
    assert c["loss_scale"] == 32.0 and not c["skipped"]                 # reading the LAST log settles all three, in order
    assert ts._pending == [] and ts.loss_scale == 32.0 and ts._clean_steps == 1
    assert b["skipped"] and b["loss_scale"] == 32.0 and b["grad_norm"] != b["grad_norm"]
    assert a["loss_scale"] == 64.0 and a["rd_loss"] == 9.5 and abs(a["grad_norm"] - 4.0) < 1e-6 and set(a) == set(StepLog.KEYS)
    d = step(0)
    ts._settle()                                                        # what the next step's start does
    assert ts.loss_scale == 64.0 and d["loss_scale"] == 64.0            # second clean step in a row: growth_interval = 2
    ts.dynamic_scale = False
    e = step(1)
    assert e["skipped"] and ts.loss_scale == 64.0                       # a static scale never moves; the step is still reported as skipped


def test_mirror_pool_carves_zeroed_aligned_mirrors_from_arenas():
    """autograd.MirrorPool.get: mirrors are views of a few large zeroed arenas (256-byte aligned, the original's shape and dtype), a
    non-contiguous original gets its own buffer, and a recycled mirror is handed out again for the same key"""
    from tdvc_amd.autograd import MirrorPool
    pool = MirrorPool()
    pool.ARENA_BYTES = 1 << 16
    x16, x32 = torch.empty(2, 5, 7, 8, dtype=torch.float16), torch.empty(3, 9, dtype=torch.float32)
    m16, m32 = pool.get(x16), pool.get(x32)
    assert m16.shape == x16.shape and m16.dtype == torch.float16 and m32.shape == x32.shape and m32.dtype == torch.float32
    assert float(m16.abs().sum()) == 0.0 and float(m32.abs().sum()) == 0.0
    base = pool.arenas[0][0]
    # offsets inside an arena are multiples of 256 bytes (the arena itself is as aligned as the device allocator makes it: 512 B on the GPU)
    assert (m16.data_ptr() - base.data_ptr()) % 256 == 0 and (m32.data_ptr() - base.data_ptr()) % 256 == 0 and len(pool.arenas) == 1
    assert base.data_ptr() <= m32.data_ptr() < base.data_ptr() + base.numel()
    big = pool.get(torch.empty(40000, dtype=torch.float16))             # does not fit the rest of the first arena: a second one
    assert len(pool.arenas) == 2 and big.numel() == 40000
    nc = torch.empty(4, 6, dtype=torch.float16).t()
    loose = pool.get(nc)
    assert loose.shape == nc.shape and len(pool.loose) == 1
    pool.free.setdefault((tuple(x16.shape), x16.dtype, x16.device), []).append(m16)      # what recycle() does after the fills
    assert pool.get(x16) is m16


def test_tape_lazy_identity_gradients(monkeypatch):
    """Tape.add_identity / grad_for_write (host logic, CPU tensors, the add kernel replaced by a recorder): an identity gradient waits for the
    next writer of the mirror; a dgrad conv that is the first to touch the mirror takes it as its residual operand and overwrites; anything
    else -- a second writer, a plain grad(), another view of the buffer, an fp32 map -- gets the add the plain way, exactly once"""
    from tdvc_amd import autograd
    calls = []
    monkeypatch.setattr(autograd, "accumulate", lambda dst, src, sign=1.0: calls.append((dst.t.data_ptr(), dst.off, dst.C, src)))
    h = lambda c=16, dt=torch.float16: ops.FM(torch.zeros(1, 4, 4, c, dtype=dt))
    tape = autograd.Tape()
    x, src = h(), h()
    assert tape.add_identity(x, src) is True and calls == []                 # deferred
    gx, acc, extra = tape.grad_for_write(x)
    assert acc is False and extra is src and calls == [] and gx.t.data_ptr() == tape.gbuf[x.t.data_ptr()].data_ptr()
    _, acc2, extra2 = tape.grad_for_write(x)                                  # a second writer accumulates, nothing waits any more
    assert acc2 is True and extra2 is None and calls == []
    assert tape.add_identity(x, src) is False and len(calls) == 1             # the mirror holds a partial sum now: immediate add
    y = h()
    assert tape.add_identity(y, src) is True
    gy = tape.grad(y)                                                         # a plain reader first: the add happens in front of it
    assert len(calls) == 2 and calls[-1][0] == gy.t.data_ptr() and calls[-1][3] is src
    z, s16 = h(32), h(16)
    assert tape.add_identity(z.ch(0, 16), s16) is True
    _, acc3, extra3 = tape.grad_for_write(z.ch(16, 16))                       # another view of the same buffer: flush, then accumulate
    assert acc3 is True and extra3 is None and len(calls) == 3 and calls[-1][1:3] == (0, 16)
    w = h()
    _, acc4, extra4 = tape.grad_for_write(w)                                  # untouched, nothing waiting: plain overwrite
    assert acc4 is False and extra4 is None
    f = h(16, torch.float32)
    assert tape.add_identity(f, h(16, torch.float32)) is False and len(calls) == 4      # fp32 maps: immediate
    prev, autograd.LAZY_IDENTITY = autograd.LAZY_IDENTITY, False
    try:
        v = h()
        assert tape.add_identity(v, src) is False and len(calls) == 5
        assert tape.grad_for_write(h())[1] is True
    finally:
        autograd.LAZY_IDENTITY = prev
    tape.release()
    assert not tape.pending and not tape.touched
