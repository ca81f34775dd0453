"""End-to-end: MI355X `VideoCompressor` vs the CPU oracle on identical weights and inputs.

Gates: |dPSNR| <= 0.02 dB and |dbpp| <= 0.001 against the oracle; stage-wise
drift is reported (fp16 activations vs the fp32 oracle) and bounded loosely per stage."""
import math

import pytest
import torch

from util import fm_to_cpu

pytestmark = pytest.mark.gpu


def _build():
    from oracle.tdvc_ref import VideoCompressor as Ref
    from tdvc_amd.model import VideoCompressor
    from tdvc_amd.synth import fill_parameters
    ref = Ref().eval()
    fill_parameters(ref)
    m = VideoCompressor()
    m.load_state_dict(ref.state_dict(), strict=True)
    return ref, m.cuda().eval()


def psnr(a, b):
    return 10 * math.log10(1.0 / float(((a - b) ** 2).mean()))


@pytest.fixture(scope="module")
def models():
    return _build()


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-12))


def _stages(tr_g, tr_o):
    st = {}
    for k in ("f_cur", "f_ref", "estmv", "mv_x_hat", "pred1", "pred", "resid", "recon_f"):
        st[k] = rel(fm_to_cpu(tr_g[k]), tr_o[k].float())
    for c in ("mv", "res"):
        dbg = tr_o[c + "_dbg"]
        st[c + ".y"] = rel(fm_to_cpu(tr_g[c]["y"]), dbg["y"])
        st[c + ".z"] = rel(fm_to_cpu(tr_g[c]["z"]), dbg["z"])
        st[c + ".y_hat_flips"] = float((fm_to_cpu(tr_g[c]["y_hat"]) != dbg["y_hat"]).float().mean())
        gp = fm_to_cpu(tr_g[c]["gp"])
        st[c + ".scales"] = rel(gp[:, :128], dbg["scales"])
        st[c + ".means"] = rel(gp[:, 128:], dbg["means"])
    return st


# 192x320: FeatureFix scale = int(192 / 8) = 24 and 320 / 24 floors to 13 pooled columns (8 px dropped), the gather blocks
# (72 px) overhang the right and bottom edges: the non-divisible geometry of the 1080p frame (scale 136) at test size
@pytest.mark.parametrize("H,W", [(64, 64), (128, 192), (192, 320), (256, 256)])
def test_forward_vs_oracle(models, H, W, report):
    """Per-frame parity: both paths code frame t from the SAME reference list (the oracle's
    reconstructions), so the deltas measure this frame's arithmetic only.  The closed-loop run
    (each path continuing from its own output, tools/predict.py:68) is reported as drift: the
    untrained synthetic network amplifies every flipped quantiser symbol, so drift is gated
    loosely and the per-frame numbers are the parity statement."""
    from tdvc_amd.synth import make_gop, ref_list
    ref, m = models
    nfr = 4 if H <= 128 else 3
    g = make_gop(1234, nfr, H, W)
    refs_o, refs_g = [g[0:1]], [g[0:1].cuda()]
    for t in range(1, nfr):
        tr_o, tr_g = {}, {}
        with torch.no_grad():
            ro, bro, bmo = ref(g[t:t + 1], ref_list(refs_o), False, trace=tr_o)
            rg, brg, bmg = m(g[t:t + 1].cuda(), ref_list(refs_o).cuda(), True, trace=tr_g)      # open loop
            rc, brc, bmc = m(g[t:t + 1].cuda(), ref_list(refs_g), True)                           # closed loop
        rg_c, rc_c = rg.cpu(), rc.cpu()
        st = _stages(tr_g, tr_o)
        assert torch.equal(tr_g["ff_idx"].cpu().long(), ref.loopfilter.last_match_index), "in-loop filter patch argmax differs"
        p_o, p_g, p_c = psnr(ro, g[t:t + 1]), psnr(rg_c, g[t:t + 1]), psnr(rc_c, g[t:t + 1])
        report(f"[{H}x{W} frame {t}] PSNR oracle {p_o:.4f} gpu {p_g:.4f} (closed-loop {p_c:.4f}) | bpp_res "
               f"{float(bro):.5f}/{float(brg):.5f} ({float(brc):.5f}) bpp_mv {float(bmo):.5f}/{float(bmg):.5f} ({float(bmc):.5f})"
               f" | max|recon diff| {float((ro - rg_c).abs().max()):.4f} PSNR(gpu,oracle) {psnr(rg_c, ro):.2f} dB")
        report("   stage rel-L2: " + " ".join(f"{k}={v:.2e}" for k, v in st.items()))
        # 0.02 dB is the gate at sizes where symbol flips average out; a 64x64 frame has 2048 latents
        # per coder and ONE flipped symbol rewrites ~6 % of the picture, so the smallest case gets 0.05
        assert abs(p_o - p_g) <= (0.02 if H * W >= 128 * 192 else 0.05), f"PSNR delta {p_o - p_g}"
        # 0.001 bpp at the codec's trained operating point (~0.1 bpp at lambda=2048) is ~1 % of the
        # rate; the untrained synthetic weights run at several bpp, so the gate is 0.001 + 0.5 %.
        assert abs(float(bro) - float(brg)) <= 1e-3 + 5e-3 * float(bro), "bpp_res delta"
        assert abs(float(bmo) - float(bmg)) <= 1e-3 + 5e-3 * float(bmo), "bpp_mv delta"
        for k in ("f_cur", "f_ref", "estmv"):                 # upstream of any quantiser: pure fp16-vs-fp32 drift
            assert st[k] < 1e-2, (k, st[k])
        for k in ("pred1", "pred", "resid"):                  # downstream of round(): includes flipped symbols
            assert st[k] < 3e-2, (k, st[k])
        # quantiser-symbol flips vs the fp32 oracle (diagnostic): fp16 activations upstream of round();
        # the residual coder's input already carries the motion coder's flips, hence the wider bound
        assert st["mv.y_hat_flips"] < 5e-3 and st["res.y_hat_flips"] < 2e-2
        assert abs(p_o - p_c) <= 0.15, f"closed-loop PSNR drift {p_o - p_c}"
        assert abs(float(bro + bmo) - float(brc + bmc)) <= 2e-2 * float(bro + bmo), "closed-loop bpp drift"
        refs_o.append(ro)
        refs_g.append(rc)


def test_module_api(models):
    """drop-in surface used by tools/train.py / tools/predict.py"""
    from tdvc_amd.synth import split_optim_params
    ref, m = models
    assert type(m).__name__ == "VideoCompressor"
    main, aux = split_optim_params(m)
    assert len(aux) == 2 and all(n.endswith(".quantiles") for n in aux)
    assert set(m.state_dict().keys()) == set(ref.state_dict().keys())
    x = torch.zeros(1, 3, 60, 64).cuda()
    with pytest.raises(RuntimeError):
        m(x, torch.zeros(1, 4, 3, 60, 64).cuda(), True)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 4, 3, 64, 64), True)


@pytest.mark.parametrize("H,W", [(64, 64)])
def test_batched_small_frame_inference(models, H, W, report):
    """B = 2 at sizes where B*H*W >= 8192 > H*W: the fused temporal conv + broadcast add (conv_mfma_v5's bcast mode) takes
    maps of >= 8192 pixels PER IMAGE; below that the fusion block must fall back to conv + bcast_add_act (r03 gated on the
    batch total and raised TdvcHipError here).  Batched output == the two single-image runs."""
    from tdvc_amd.synth import make_gop, ref_list
    _, m = models
    gs = [make_gop(1234 + i, 2, H, W) for i in range(2)]
    x = torch.cat([g[1:2] for g in gs]).cuda()
    refs = torch.cat([ref_list([g[0:1]]) for g in gs]).cuda()
    with torch.no_grad():
        rb, brb, bmb = m(x, refs, True)
        singles = [m(x[i:i + 1], refs[i:i + 1], True) for i in range(2)]
    # a batch takes other kernels than a single image at these sizes (the small-map split-K kernel counts pixels over the batch),
    # so sums round differently and the untrained filler network turns a flipped quantiser symbol into a visible patch: the
    # two runs agree like two arithmetic orders of one frame (PSNR between them), not bit for bit
    for i in range(2):
        d = rb[i:i + 1] - singles[i][0]
        agree = 10 * math.log10(1.0 / max(float((d ** 2).mean()), 1e-12))
        report(f"[B=2 {H}x{W}] item {i}: max|batched - single| {float(d.abs().max()):.2e}, PSNR(batched, single) {agree:.1f} dB")
        assert agree >= 45.0
    bs = sum(float(s_[1] + s_[2]) for s_ in singles) / 2
    assert abs(float(brb + bmb) - bs) <= 5e-3 * bs


def test_training_mode_forward(models, report):
    """`.train()` forward: additive-noise quantisation, FeatureFix scale 8, 5-tuple return
    (pnet.py:80-83).  With the noise drawn from U(-0.5, 0.5) both paths are stochastic, so the
    comparison is statistical: rate and distortion of the GPU path must sit within the spread of the
    oracle's own seeds.  (No autograd graph yet: backward kernels are a later round.)"""
    from tdvc_amd.synth import make_gop, ref_list
    ref, m = models
    g = make_gop(1234, 2, 128, 192)
    refs = ref_list([g[0:1]])
    ref.train(); m.train()
    try:
        o_bpp, o_mse = [], []
        with torch.no_grad():
            for seed in range(3):
                torch.manual_seed(seed)
                ro, bro, bmo, a1, a2 = ref(g[1:2], refs, False)
                o_bpp.append(float(bro + bmo)); o_mse.append(float(((ro - g[1:2]) ** 2).mean()))
            torch.manual_seed(7)
            out = m(g[1:2].cuda(), refs.cuda(), True)
        assert len(out) == 5
        rg, brg, bmg, ag1, ag2 = out
        g_bpp, g_mse = float(brg + bmg), float(((rg.cpu() - g[1:2]) ** 2).mean())
        report(f"train-mode forward: bpp oracle seeds {o_bpp} gpu {g_bpp:.4f}; mse oracle {o_mse} gpu {g_mse:.5f}; "
               f"aux {float(a1):.3f}/{float(ag1):.3f}")
        mb, mm = sum(o_bpp) / 3, sum(o_mse) / 3
        assert abs(g_bpp - mb) < 0.03 * mb and abs(g_mse - mm) < 0.05 * mm
        assert abs(float(ag1) - float(a1)) < 1e-3 * abs(float(a1)) + 1e-3 and abs(float(ag2) - float(a2)) < 1e-3 * abs(float(a2)) + 1e-3
    finally:
        ref.eval(); m.eval()


def test_forward_vs_oracle_1080p(models, report):
    """The headline configuration itself (BASELINE.json configs[1], SURVEY 8d: seed 1234, 1080x1920 padded to 1088x1920 by
    `pad(., 64)`): ONE P-frame on the HIP path and on the fp32 CPU oracle (about a minute of CPU).  This is where the
    geometry differs from every smaller case: FeatureFix scale = int(1088 / 8) = 136 with floor pooling (1920 / 136 = 14.1,
    the last 16 px dropped), 408-px gather blocks overhanging the frame, centred 4 + 4 row padding, > 256 tiles per
    persistent workgroup in every conv kernel.  Gates: the north_star's own, absolute: |dPSNR| <= 0.02 dB and |dbpp| <= 0.001
    on the frame's total rate; patch-match indices bit-equal."""
    import time
    import torch.nn.functional as F
    from tdvc_amd.synth import make_gop, ref_list
    ref, m = models
    g = F.pad(make_gop(1234, 2, 1080, 1920), (0, 0, 4, 4))              # utils.pad(.., 64): 1080 -> 1088, centred
    refs = ref_list([g[0:1]])
    tr_o, tr_g = {}, {}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        t0 = time.time()
        ro, bro, bmo = ref(g[1:2], refs, False, trace=tr_o)
        t_cpu = time.time() - t0
        rg, brg, bmg = m(g[1:2].cuda(), refs.cuda(), True, trace=tr_g)
    rg_c = rg.cpu()
    idx_g, idx_o = tr_g["ff_idx"].cpu().long(), ref.loopfilter.last_match_index
    st = _stages(tr_g, tr_o)
    crop = lambda t: t[:, :, 4:-4]                                       # utils.crop: PSNR on the 1080 visible rows (predict.py:69-70,87)
    p_o, p_g = psnr(crop(ro), crop(g[1:2])), psnr(crop(rg_c), crop(g[1:2]))
    report(f"[1088x1920] oracle {t_cpu:.1f} s on {torch.get_num_threads()} threads | PSNR oracle {p_o:.4f} gpu {p_g:.4f} | bpp_res "
           f"{float(bro):.5f}/{float(brg):.5f} bpp_mv {float(bmo):.5f}/{float(bmg):.5f} (d total {float(brg + bmg) - float(bro + bmo):+.5f}) | "
           f"max|recon diff| {float((ro - rg_c).abs().max()):.4f} PSNR(gpu,oracle) {psnr(rg_c, ro):.2f} dB | patch indices {idx_g.tolist()}")
    report("   stage rel-L2: " + " ".join(f"{k}={v:.2e}" for k, v in st.items()))
    assert idx_o.shape == (1, 24) and torch.equal(idx_g, idx_o), f"patch argmax differs: {idx_g.tolist()} vs {idx_o.tolist()}"
    assert abs(p_o - p_g) <= 0.02, f"PSNR delta {p_o - p_g}"
    # SURVEY 8d / north_star, absolute, at the size they are stated for: |d bpp| <= 0.001 on the frame's total rate (a 4.66 bpp
    # frame with the filler weights: 0.02 % of the rate; measured +0.0008)
    assert abs(float(brg + bmg) - float(bro + bmo)) <= 1e-3, f"bpp delta {float(brg + bmg) - float(bro + bmo):+.5f} misses the absolute 0.001 gate"
    for k in ("f_cur", "f_ref", "estmv"):
        assert st[k] < 1e-2, (k, st[k])
    for k in ("pred1", "pred", "resid"):
        assert st[k] < 3e-2, (k, st[k])
    assert st["mv.y_hat_flips"] < 5e-3 and st["res.y_hat_flips"] < 2e-2


@pytest.mark.parametrize("H,W", [(128, 192), (256, 256)])
def test_forward_fp32_islands(models, H, W, report):
    """`enabled_amp=False`: both coders run as fp32 islands (the reference's pnet.py:33-49,57-73 under any setting).  The
    AMP regions around them stay fp16-in / fp32-accumulate, so the coders' INPUTS still differ from the fp32 oracle's in
    the last fp16 bit; what the mode removes is the coders' own rounding: fewer flipped symbols and a smaller rate delta
    than the default mode on the same frame."""
    from tdvc_amd.synth import make_gop, ref_list
    ref, m = models
    g = make_gop(1234, 2, H, W)
    refs = ref_list([g[0:1]])
    tr_o, tr_16, tr_32 = {}, {}, {}
    with torch.no_grad():
        ro, bro, bmo = ref(g[1:2], refs, False, trace=tr_o)
        r16, br16, bm16 = m(g[1:2].cuda(), refs.cuda(), True, trace=tr_16)
        r32, br32, bm32 = m(g[1:2].cuda(), refs.cuda(), False, trace=tr_32)
    s16, s32 = _stages(tr_16, tr_o), _stages(tr_32, tr_o)
    d16 = abs(float(br16 + bm16) - float(bro + bmo))
    d32 = abs(float(br32 + bm32) - float(bro + bmo))
    p_o, p_16, p_32 = psnr(ro, g[1:2]), psnr(r16.cpu(), g[1:2]), psnr(r32.cpu(), g[1:2])
    report(f"[{H}x{W}] fp32 islands vs default: |dbpp| {d32:.5f} vs {d16:.5f} (oracle {float(bro + bmo):.4f} bpp); PSNR oracle {p_o:.4f} "
           f"fp32-islands {p_32:.4f} default {p_16:.4f}; y_hat flips mv {s32['mv.y_hat_flips']:.2e} vs {s16['mv.y_hat_flips']:.2e}, "
           f"res {s32['res.y_hat_flips']:.2e} vs {s16['res.y_hat_flips']:.2e}; mv.y rel-L2 {s32['mv.y']:.2e} vs {s16['mv.y']:.2e}")
    assert abs(p_o - p_32) <= 0.02
    assert d32 <= 1e-3 + 2.5e-3 * float(bro + bmo)
    assert s32["mv.y_hat_flips"] <= s16["mv.y_hat_flips"] + 1e-4 and s32["mv.y"] <= s16["mv.y"]


SWA_ITERS = 100


def _train_to_operating_point(report, max_iters=600, target_bpp=0.30, lam=256.0):
    """deterministic TrainSteps on synthetic septuplets (batch 4, 256x256, the tools/train.py sample rule) from the filler
    initialisation until the training-mode rate is under `target_bpp` (or `max_iters`), then SWA_ITERS more whose iterates are
    averaged; -> (model in eval mode holding the LAST iterate, rate ema, state-dict of the averaged iterates)"""
    from tdvc_amd.model import VideoCompressor
    from tdvc_amd.synth import fill_parameters, make_gop, ref_list
    from tdvc_amd import ops
    from tdvc_amd.train import TrainStep
    torch.manual_seed(1111)                                   # tools/train.py:253-256
    prev_det, ops.DETERMINISTIC = ops.DETERMINISTIC, True   # run-to-run reproducible steps: every suite run tests the SAME model
    try:
        net = VideoCompressor()
        fill_parameters(net)
        net = net.cuda().train()
        step = TrainStep(net, train_lambda=lam, lr=2e-4, loss_scale=128.0)
        pool, cursor, ema, log = [], 0, None, None
        swa, swa_left = None, None
        for it in range(max_iters + SWA_ITERS):
            while len(pool) < 4:
                gop = make_gop(5000 + cursor, 7, 256, 256)
                cursor += 1
                for t in range(1, 7):                             # dataset.py:211-232: refs [I, x(t-3), x(t-2), x(t-1)] with the repeat rules
                    pool.append((gop[t:t + 1], ref_list([gop[k:k + 1] for k in range(0, t)][-4:] if t > 3 else [gop[k:k + 1] for k in range(0, t)])))
            batch, pool = pool[:4], pool[4:]
            x = torch.cat([b[0] for b in batch]).cuda()
            refs = torch.cat([b[1] for b in batch]).cuda()
            log = step(x, refs)
            bpp = log["bpp_res"] + log["bpp_mv"]
            ema = bpp if ema is None else 0.9 * ema + 0.1 * bpp
            if (it + 1) % 50 == 0:
                report(f"   train-to-operating-point it {it + 1}: rd_loss {log['rd_loss']:.4f} bpp {bpp:.4f} (ema {ema:.4f}) mse {log['mse']:.2e}")
            if swa_left is None and ((it >= 100 and ema <= target_bpp) or it + 1 >= max_iters):
                swa_left = SWA_ITERS                          # the operating point is reached: average the next SWA_ITERS iterates
            elif swa_left is not None:
                sd_ = {k: v.detach().double() for k, v in net.state_dict().items() if v.dtype.is_floating_point}
                swa = {k: v.clone() for k, v in sd_.items()} if swa is None else {k: swa[k] + sd_[k] for k in sd_}
                swa_left -= 1
                if swa_left == 0:
                    break
    finally:
        ops.DETERMINISTIC = prev_det                         # an exception during training must not leave the global on for later tests
    import hashlib
    hsh = hashlib.sha256()
    for k, v in sorted(net.state_dict().items()):
        hsh.update(k.encode()); hsh.update(v.detach().cpu().contiguous().numpy().tobytes())
    report(f"   trained {it + 1} iterations (lambda {lam:g}): training-mode bpp ema {ema:.4f}; state-dict sha256 {hsh.hexdigest()[:16]} "
           f"(deterministic TrainSteps from seed 1111: the same digest on every run of this build)")
    swa_sd = {k: ((swa[k] / SWA_ITERS).float().cpu() if k in swa else v.detach().cpu().clone()) for k, v in net.state_dict().items()}
    return net.eval(), ema, swa_sd


def _fp16_exact(sd):
    """the state-dict with every conv / DCN weight rounded to an fp16-representable value (biases, SE MLPs' use, entropy
    parameters, GDN parameters untouched): a checkpoint whose weights the HIP path's fp16 packing reproduces EXACTLY"""
    out = {}
    for k, v in sd.items():
        conv_w = k.endswith(".weight") and v.dim() >= 4
        out[k] = v.detach().half().float() if conv_w else v.detach().clone()
    return out


class _Trained:
    pass


@pytest.fixture(scope="module")
def trained(report):
    """Models at a trained operating point, built once per module run (deterministic training: the same weights every run).

    `last`: the LAST training iterate.  Round 3 found the fp32 oracle on its raw fp32 master weights 0.02-0.035 dB below the HIP
    path, one sign on all frames; round 4 measured why (tools/parity_diag.py): the last iterate of a constant
    learning-rate run is HYPERSENSITIVE -- the HIP path itself loses 0.008-0.031 dB when every conv weight is multiplied by
    (1 + 2^-12 xi) (half an fp16 ulp, a second independent "rounding"), and on it the reference's own two paths (fp32 CPU
    restatement / AMP emulation) differ by 0.04 dB, twice the north_star gate.  No arithmetic can sit within 0.02 dB of both.
    `net` / `ref` keep round 3's statement on that iterate: an fp16-EXACT copy of it (`_fp16_exact`) on the HIP path and on the
    fp32 oracle, i.e. one function evaluated by two arithmetics.

    `swa_*`: the mean of SWA_ITERS further iterates (stochastic weight averaging: what a converged, learning-rate-decayed run
    ends on), as a RAW fp32 checkpoint -- not fp16-exact, the form a user's checkpoint has -- on the HIP path, on the oracle in
    AMP emulation (`amp_emulation = True`: what the reference computes on a GPU with `enable_amp: True`, fp16 convs in the
    three autocast regions around fp32 coders) and on the fp32 oracle.  On it the perturbation costs the HIP path < 0.005 dB and
    all paths agree: this is the checkpoint the north_star gates are asserted on."""
    from oracle.tdvc_ref import VideoCompressor as Ref
    from tdvc_amd.model import VideoCompressor

    def hip(sd):
        m = VideoCompressor()
        m.load_state_dict(sd, strict=True)
        return m.cuda().eval()

    def oracle(sd, amp):
        r = Ref().eval()
        r.load_state_dict(sd, strict=True)
        r.amp_emulation = amp
        return r

    T = _Trained()
    T.last, ema, swa_sd = _train_to_operating_point(report)
    T.last_sd = {k: v.detach().cpu() for k, v in T.last.state_dict().items()}
    sd = _fp16_exact(T.last_sd)
    T.net, T.ref = hip(sd), oracle(sd, False)
    T.last_amp = oracle(T.last_sd, True)
    T.swa_sd = swa_sd
    T.swa_hip, T.swa_amp, T.swa_fp32 = hip(swa_sd), oracle(swa_sd, True), oracle(swa_sd, False)
    return T


def test_trained_master_weights_effect(trained, report):
    """what the precision of the CHECKPOINT alone does at the trained operating point: the fp32 CPU oracle with the raw fp32
    master weights of the training run against the same oracle with the fp16-exact checkpoint (two checkpoints 2^-11 apart,
    one arithmetic), next to the HIP path (which packs both to the same fp16 conv weights).  Reported, and bounded at three
    times the north_star gate: this is the deviation a per-frame 0.02 dB gate against master weights would measure before
    any kernel arithmetic enters."""
    from oracle.tdvc_ref import VideoCompressor as Ref
    from tdvc_amd.synth import make_gop, ref_list
    net, ref, raw = trained.net, trained.ref, trained.last
    ref_raw = Ref().eval()
    ref_raw.load_state_dict({k: v.detach().cpu() for k, v in raw.state_dict().items()}, strict=True)
    g = make_gop(1234, 3, 256, 256)
    refs_l = [g[0:1]]
    for t in (1, 2):
        refs = ref_list(refs_l)
        with torch.no_grad():
            r_e, br_e, bm_e = ref(g[t:t + 1], refs, False)
            r_m, br_m, bm_m = ref_raw(g[t:t + 1], refs, False)
            r_g, br_g, bm_g = net(g[t:t + 1].cuda(), refs.cuda(), True)
        pe, pm, pg = psnr(r_e, g[t:t + 1]), psnr(r_m, g[t:t + 1]), psnr(r_g.cpu(), g[t:t + 1])
        report(f"[trained, 256x256 frame {t}] oracle, fp16-exact checkpoint {pe:.4f} dB {float(br_e + bm_e):.5f} bpp | oracle, fp32 master weights "
               f"{pm:.4f} dB ({pm - pe:+.4f}) {float(br_m + bm_m):.5f} bpp ({float(br_m + bm_m) - float(br_e + bm_e):+.5f}) | HIP path {pg:.4f} dB ({pg - pe:+.4f})")
        assert abs(pm - pe) <= 0.06 and abs(float(br_m + bm_m) - float(br_e + bm_e)) <= 3e-3
        refs_l.append(r_e)


def test_trained_operating_point_parity(trained, report):
    """SURVEY 8d's absolute gates (|dbpp| <= 0.001, |dPSNR| <= 0.02 dB) at a TRAINED-like operating point instead of the
    several-bpp filler weights: <= 600 deterministic TrainSteps (ops.DETERMINISTIC: the same weights on every run, digest in
    the report) bring the rate to a few tenths of a bpp, the state-dict moves to the fp32 CPU oracle (strict=True), and
    P-frames at 256x256 and 512x768 are coded by both, in the default (fp16 coders) mode and in the fp32-island mode.  Both
    differences are sums of rare discrete events (a quantiser symbol that falls on the other side of .5 because the features
    upstream are fp16; in the motion latents it also changes the prediction) and a 65 k-pixel frame does not average such
    events out: from 512x768 up the gates themselves are asserted per frame (the headline size: the next test); at 256x256
    the PSNR gate holds for the median with three times the gate on every single frame, the rate as a distribution over 18
    frames (below), next to the direct statement that the two reconstructions agree to > 65 dB."""
    from tdvc_amd.synth import make_gop, ref_list
    net, ref = trained.net, trained.ref
    worst, d256, dps = 0.0, [], []
    for (H, W, seeds) in ((256, 256, (1234, 1235, 1236)), (512, 768, (1234,))):
        big = H * W >= 512 * 768
        for seed in seeds:
            g = make_gop(seed, 3, H, W)
            refs_l = [g[0:1]]
            for t in (1, 2):
                refs = ref_list(refs_l)
                with torch.no_grad():
                    ro, bro, bmo = ref(g[t:t + 1], refs, False)
                    r16, br16, bm16 = net(g[t:t + 1].cuda(), refs.cuda(), True)
                    r32, br32, bm32 = net(g[t:t + 1].cuda(), refs.cuda(), False)
                bo = float(bro + bmo)
                d16, d32 = float(br16 + bm16) - bo, float(br32 + bm32) - bo
                p_o, p_16, p_32 = psnr(ro, g[t:t + 1]), psnr(r16.cpu(), g[t:t + 1]), psnr(r32.cpu(), g[t:t + 1])
                report(f"[trained, {H}x{W} seed {seed} frame {t}] oracle {bo:.5f} bpp {p_o:.4f} dB | default mode dbpp {d16:+.5f} dPSNR {p_16 - p_o:+.4f} | "
                       f"fp32 islands dbpp {d32:+.5f} dPSNR {p_32 - p_o:+.4f} | PSNR(gpu, oracle) {psnr(r16.cpu(), ro):.1f} dB")
                assert bo < 1.0, "not a trained-like operating point"      # eval mode (rounding, int(H/8) matching) on the first frames of a GOP: under 1 bpp
                # the reconstructions themselves agree to > 65 dB (measured 77-82 dB, no flipped symbol on most frames); against the
                # SOURCE that deviation is partly correlated with the coding error (fp16 weights are one fixed perturbation of the
                # model): at 33-36 dB a deviation of 8e-5 RMS can move the PSNR by up to 0.04 dB, measured 0.002-0.022 dB
                agree16, agree32 = psnr(r16.cpu(), ro), psnr(r32.cpu(), ro)
                assert agree16 >= 65.0 and agree32 >= 65.0, f"reconstructions differ: PSNR(gpu, oracle) {agree16:.1f} / {agree32:.1f} dB"
                gate_p = 0.02                            # r03: 0.06 at 256x256; measured 0.0013-0.0057 dB on the fp16-exact checkpoint
                assert abs(p_16 - p_o) <= gate_p and abs(p_32 - p_o) <= gate_p, f"a {H}x{W} frame misses the {gate_p} dB PSNR bound"
                assert abs(d32) <= (1e-3 if big else 3e-3), "fp32-island mode misses the rate bound at the trained operating point"
                if big:
                    assert abs(d16) <= 1e-3, "default mode misses the SURVEY 8d rate gate at the trained operating point"
                else:
                    d256.append(d16)
                dps.append((abs(p_16 - p_o), abs(p_32 - p_o), abs(d32)))
                worst = max(worst, abs(d16))
                refs_l.append(ro)
    med = lambda vals: sorted(vals)[len(vals) // 2]
    report(f"[trained, {len(dps)} oracle frames] median |dPSNR| default {med([v[0] for v in dps]):.4f} islands {med([v[1] for v in dps]):.4f} dB; "
           f"median |dbpp| islands {med([v[2] for v in dps]):.5f}")
    assert med([v[0] for v in dps]) <= 0.02 and med([v[1] for v in dps]) <= 0.02 and med([v[2] for v in dps]) <= 1e-3
    # The default mode's rate at 256x256 over 6 + 12 frames; the fp32-island mode (= the oracle's bits on identical coder
    # inputs, asserted above) is the reference of the extra 12.  tools/trained_point_sweep.py: median 2-3e-5 bpp, 1-2 of 32
    # frames over 1e-3 (up to 2.4e-3); none of 32 at 512x768; at 1088x1920 the same events are 30x smaller per pixel.
    for s_ in range(6):
        g = make_gop(7000 + s_, 3, 256, 256).cuda()
        refs_l = [g[0:1]]
        for t in (1, 2):
            refs = ref_list(refs_l)
            with torch.no_grad():
                r32, br32, bm32 = net(g[t:t + 1], refs, False)
                r16, br16, bm16 = net(g[t:t + 1], refs, True)
            d256.append(float(br16 + bm16) - float(br32 + bm32))
            refs_l.append(r32)
    a256 = sorted(abs(v) for v in d256)
    report(f"[trained, 256x256, {len(a256)} frames] default mode |dbpp|: median {a256[len(a256) // 2]:.5f} max {a256[-1]:.5f}, over 0.001: {sum(v > 1e-3 for v in a256)}")
    assert a256[len(a256) // 2] <= 5e-4 and sum(v > 1e-3 for v in a256) <= 5 and a256[-1] <= 5e-3


def test_trained_operating_point_parity_1080p(trained, report):
    """The north_star gates at the size AND in the regime they are stated for: one P-frame of the cfg-2 GOP at 1088x1920 coded
    with the trained weights by the HIP path (default mode and fp32 islands) and by the fp32 CPU oracle (about a minute):
    |dPSNR| <= 0.02 dB (on the 1080 visible rows, predict.py:69-70,87) and |dbpp| <= 0.001, absolute, per frame, both modes;
    patch-match indices bit-equal."""
    import time
    import torch.nn.functional as F
    from tdvc_amd.synth import make_gop, ref_list
    net, ref = trained.net, trained.ref
    g = F.pad(make_gop(1234, 2, 1080, 1920), (0, 0, 4, 4))
    refs = ref_list([g[0:1]])
    torch.set_num_threads(min(16, torch.get_num_threads()))
    tr16 = {}
    with torch.no_grad():
        t0 = time.time()
        ro, bro, bmo = ref(g[1:2], refs, False)
        t_cpu = time.time() - t0
        r16, br16, bm16 = net(g[1:2].cuda(), refs.cuda(), True, trace=tr16)
        r32, br32, bm32 = net(g[1:2].cuda(), refs.cuda(), False)
    crop = lambda t: t[:, :, 4:-4]
    bo = float(bro + bmo)
    p_o, p_16, p_32 = psnr(crop(ro), crop(g[1:2])), psnr(crop(r16.cpu()), crop(g[1:2])), psnr(crop(r32.cpu()), crop(g[1:2]))
    d16, d32 = float(br16 + bm16) - bo, float(br32 + bm32) - bo
    report(f"[trained, 1088x1920] oracle {t_cpu:.1f} s | oracle {bo:.5f} bpp {p_o:.4f} dB | default mode dbpp {d16:+.6f} dPSNR {p_16 - p_o:+.5f} | "
           f"fp32 islands dbpp {d32:+.6f} dPSNR {p_32 - p_o:+.5f} | PSNR(gpu, oracle) {psnr(r16.cpu(), ro):.1f} / {psnr(r32.cpu(), ro):.1f} dB")
    assert bo < 1.0, "not a trained-like operating point"
    assert torch.equal(tr16["ff_idx"].cpu().long(), ref.loopfilter.last_match_index), "in-loop filter patch argmax differs"
    assert abs(p_16 - p_o) <= 0.02 and abs(p_32 - p_o) <= 0.02, "PSNR gate (0.02 dB) missed at 1088x1920"
    assert abs(d16) <= 1e-3 and abs(d32) <= 1e-3, "rate gate (0.001 bpp) missed at 1088x1920"


def _code_frame(model, x, refs, amp):
    with torch.no_grad():
        r, br, bm = model(x, refs, amp)
    return r, float(br + bm)


def _within_reference_spread(d_hip_amp, d_f32_amp, gate):
    """the criterion against the AMP-emulating oracle: inside the gate -- or, where the reference's OWN two paths (fp32 CPU / AMP
    emulation) are further apart than half the gate, no further from the AMP path than the reference's CPU path is, plus half the gate"""
    return abs(d_hip_amp) <= max(gate, abs(d_f32_amp) + 0.5 * gate)


def _idx_note(tr, amp, f32):
    i_h = tr["ff_idx"].cpu().long()
    i_a, i_f = amp.loopfilter.last_match_index, f32.loopfilter.last_match_index
    return f"patch indices hip/fp32 {int((i_h != i_f).sum())} hip/amp {int((i_h != i_a).sum())} fp32/amp {int((i_f != i_a).sum())} of {i_h.numel()} differ"


def test_trained_raw_checkpoint_parity_amp_oracle(trained, report):
    """RAW fp32 weights, no `_fp16_exact`: the averaged checkpoint (fixture) on the HIP path against BOTH paths of the reference:
    the fp32 oracle (its CPU path, the one north_star names) and the oracle in AMP emulation (`amp_emulation = True`: what it
    computes on a GPU with `enable_amp: True`, pnet.py:27-78 -- fp16 convs around fp32 coders).  Per frame, open loop (all paths
    code from the fp32 oracle's reference list), 256x256 and 512x768.
    Against the fp32 oracle: the north_star gates themselves, |dPSNR| <= 0.02 dB and |dbpp| <= 0.001, in the fp32-island mode AND
    in the default mode (fp16 coder weights, which the reference does not have); 0.003 bpp at 256x256, where one flipped motion
    symbol is 1-2.4e-3 bpp of a 65 k-pixel frame.
    Against the AMP emulation: the same gates wherever the reference's own two paths agree with each other; they do not always
    (r04: 0.027 dB apart on the 512x768 frame at 38 dB, where the HIP path sat 0.006 dB from the fp32 path and 0.0215 dB from
    the AMP path), and no implementation can be within 0.02 dB of both then: `_within_reference_spread`."""
    from tdvc_amd.synth import make_gop, ref_list
    hip, amp, f32 = trained.swa_hip, trained.swa_amp, trained.swa_fp32
    rows = []
    for (H, W, seeds) in ((256, 256, (1234, 1235)), (512, 768, (1234,))):
        big = H * W >= 512 * 768
        for seed in seeds:
            g = make_gop(seed, 3, H, W)
            refs_l = [g[0:1]]
            for t in (1, 2):
                refs = ref_list(refs_l)
                ro, bo = _code_frame(amp, g[t:t + 1], refs, True)
                rf, bf = _code_frame(f32, g[t:t + 1], refs, False)
                tr = {}
                with torch.no_grad():
                    r16, br16, bm16 = hip(g[t:t + 1].cuda(), refs.cuda(), True, trace=tr)
                b16 = float(br16 + bm16)
                r32, b32 = _code_frame(hip, g[t:t + 1].cuda(), refs.cuda(), False)
                p_o, p_f, p_16, p_32 = psnr(ro, g[t:t + 1]), psnr(rf, g[t:t + 1]), psnr(r16.cpu(), g[t:t + 1]), psnr(r32.cpu(), g[t:t + 1])
                report(f"[raw averaged checkpoint, {H}x{W} seed {seed} frame {t}] fp32 oracle {bf:.5f} bpp {p_f:.4f} dB | fp32 islands dbpp {b32 - bf:+.5f} dPSNR {p_32 - p_f:+.4f} | "
                       f"default mode dbpp {b16 - bf:+.5f} dPSNR {p_16 - p_f:+.4f} | AMP oracle against the fp32 oracle dbpp {bo - bf:+.5f} dPSNR {p_o - p_f:+.4f}; "
                       f"HIP against the AMP oracle dPSNR {p_32 - p_o:+.4f} / {p_16 - p_o:+.4f} | PSNR(gpu, fp32 oracle) {psnr(r32.cpu(), rf):.1f} / {psnr(r16.cpu(), rf):.1f} dB, "
                       f"PSNR(gpu, AMP oracle) {psnr(r32.cpu(), ro):.1f} dB, PSNR(fp32, AMP oracle) {psnr(rf, ro):.1f} dB | {_idx_note(tr, amp, f32)}")
                assert bf < 1.2, "not a trained-like operating point"
                gate_b = 1e-3 if big else 3e-3
                assert abs(p_32 - p_f) <= 0.02 and abs(b32 - bf) <= gate_b, "fp32-island mode misses the north_star gates against the fp32 CPU oracle on a raw checkpoint"
                assert abs(p_16 - p_f) <= 0.02 and abs(b16 - bf) <= gate_b, "default mode misses the north_star gates against the fp32 CPU oracle on a raw checkpoint"
                assert _within_reference_spread(p_32 - p_o, p_f - p_o, 0.02) and _within_reference_spread(p_16 - p_o, p_f - p_o, 0.02), \
                    "the HIP path is further from the AMP emulation than the reference's own CPU path is"
                assert abs(b32 - bo) <= gate_b and abs(b16 - bo) <= gate_b, "rate gate against the AMP emulation"
                rows.append((abs(p_32 - p_f), abs(p_16 - p_f), abs(p_32 - p_o), abs(p_f - p_o)))
                refs_l.append(rf)
    report(f"[raw averaged checkpoint, {len(rows)} frames] max |dPSNR| against the fp32 oracle: islands {max(r[0] for r in rows):.4f}, default {max(r[1] for r in rows):.4f}; "
           f"against the AMP oracle {max(r[2] for r in rows):.4f}; the reference's own two paths (fp32 oracle against AMP oracle) {max(r[3] for r in rows):.4f} dB")


def test_trained_raw_checkpoint_parity_amp_oracle_1080p(trained, report):
    """the same statement at the headline size: one cfg-2 P-frame at 1088x1920, raw averaged checkpoint, HIP path (both coder
    modes) against the AMP-emulating oracle; the north_star gates themselves, patch-match indices bit-equal.  (The fp32 oracle
    at this size: test_trained_operating_point_parity_1080p, on the fp16-exact last iterate.)"""
    import time
    import torch.nn.functional as F
    from tdvc_amd.synth import make_gop, ref_list
    hip, amp = trained.swa_hip, trained.swa_amp
    g = F.pad(make_gop(1234, 2, 1080, 1920), (0, 0, 4, 4))
    refs = ref_list([g[0:1]])
    torch.set_num_threads(min(16, torch.get_num_threads()))
    t0 = time.time()
    ro, bo = _code_frame(amp, g[1:2], refs, True)
    t_cpu = time.time() - t0
    tr16 = {}
    with torch.no_grad():
        r16, br16, bm16 = hip(g[1:2].cuda(), refs.cuda(), True, trace=tr16)
    b16 = float(br16 + bm16)
    r32, b32 = _code_frame(hip, g[1:2].cuda(), refs.cuda(), False)
    crop = lambda t: t[:, :, 4:-4]
    p_o, p_16, p_32 = psnr(crop(ro), crop(g[1:2])), psnr(crop(r16.cpu()), crop(g[1:2])), psnr(crop(r32.cpu()), crop(g[1:2]))
    report(f"[raw averaged checkpoint vs AMP oracle, 1088x1920] oracle {t_cpu:.1f} s | oracle {bo:.5f} bpp {p_o:.4f} dB | fp32 islands dbpp {b32 - bo:+.6f} dPSNR {p_32 - p_o:+.5f} | "
           f"default mode dbpp {b16 - bo:+.6f} dPSNR {p_16 - p_o:+.5f} | PSNR(gpu, oracle) {psnr(r32.cpu(), ro):.1f} / {psnr(r16.cpu(), ro):.1f} dB")
    assert bo < 1.2
    assert torch.equal(tr16["ff_idx"].cpu().long(), amp.loopfilter.last_match_index), "in-loop filter patch argmax differs"
    assert abs(p_32 - p_o) <= 0.02 and abs(b32 - bo) <= 1e-3, "fp32-island mode misses the north_star gates at 1088x1920 on a raw checkpoint"
    assert abs(p_16 - p_o) <= 0.02 and abs(b16 - bo) <= 1e-3, "default mode misses the north_star gates at 1088x1920 on a raw checkpoint"


def test_trained_closed_loop_gop(trained, report):
    """BASELINE's metric is a GOP-level PSNR / BPP delta: one closed-loop GOP of 6 P-frames at 512x768 with the raw averaged
    checkpoint, EACH path continuing from its own reconstructions (the reference-list rule of tools/predict.py:55-68): the fp32
    oracle, the AMP-emulating oracle, the HIP path with fp32 islands and in the default mode.  Gates on the GOP means against
    the fp32 oracle: |dPSNR| <= 0.02 dB, |dbpp| <= 0.001; against the AMP emulation `_within_reference_spread`; the per-frame
    drift is reported and bounded at twice the gate (against the fp32 oracle)."""
    from tdvc_amd.synth import make_gop, ref_list
    hip, amp, f32 = trained.swa_hip, trained.swa_amp, trained.swa_fp32
    g = make_gop(1234, 7, 512, 768)
    lists = {"fp32": [g[0:1]], "amp": [g[0:1]], "islands": [g[0:1].cuda()], "default": [g[0:1].cuda()]}
    acc = {k: [] for k in lists}
    for t in range(1, 7):
        x = g[t:t + 1]
        rf, bf = _code_frame(f32, x, ref_list(lists["fp32"]), False)
        ro, bo = _code_frame(amp, x, ref_list(lists["amp"]), True)
        tr = {}
        with torch.no_grad():
            r32, br32, bm32 = hip(x.cuda(), ref_list(lists["islands"]), False, trace=tr)
        b32 = float(br32 + bm32)
        r16, b16 = _code_frame(hip, x.cuda(), ref_list(lists["default"]), True)
        for k, r, b in (("fp32", rf, bf), ("amp", ro, bo), ("islands", r32, b32), ("default", r16, b16)):
            acc[k].append((psnr(r.cpu(), x), b))
            lists[k].append(r)
        pf = acc["fp32"][-1][0]
        report(f"[closed-loop GOP 512x768 frame {t}] fp32 oracle {bf:.5f} bpp {pf:.4f} dB | fp32 islands dbpp {b32 - bf:+.5f} dPSNR {acc['islands'][-1][0] - pf:+.4f} | "
               f"default dbpp {b16 - bf:+.5f} dPSNR {acc['default'][-1][0] - pf:+.4f} | AMP oracle dbpp {bo - bf:+.5f} dPSNR {acc['amp'][-1][0] - pf:+.4f} | {_idx_note(tr, amp, f32)}")
        for k in ("islands", "default"):
            assert abs(acc[k][-1][0] - pf) <= 0.04 and abs(acc[k][-1][1] - bf) <= 2e-3, f"{k}: frame {t} drifts past twice the gate"
    mean = lambda k, i: sum(v[i] for v in acc[k]) / len(acc[k])
    da = mean("fp32", 0) - mean("amp", 0)
    report(f"[closed-loop GOP 512x768, 6 P-frames] the reference's own two paths: fp32 oracle {mean('fp32', 0):.4f} dB {mean('fp32', 1):.5f} bpp, AMP emulation "
           f"{mean('amp', 0):.4f} dB {mean('amp', 1):.5f} bpp (fp32 - AMP {da:+.4f} dB)")
    for k in ("islands", "default"):
        dp, db = mean(k, 0) - mean("fp32", 0), mean(k, 1) - mean("fp32", 1)
        dpa, dba = mean(k, 0) - mean("amp", 0), mean(k, 1) - mean("amp", 1)
        report(f"[closed-loop GOP 512x768, 6 P-frames] {k}: GOP-mean dPSNR {dp:+.4f} dB, dbpp {db:+.5f} against the fp32 oracle; {dpa:+.4f} dB, {dba:+.5f} against the AMP emulation")
        assert abs(dp) <= 0.02 and abs(db) <= 1e-3, f"{k}: closed-loop GOP means miss the north_star gates against the fp32 CPU oracle"
        assert _within_reference_spread(dpa, da, 0.02) and abs(dba) <= 1e-3, f"{k}: closed-loop GOP means further from the AMP emulation than the reference's own CPU path"


def _perturbed(sd, eps, seed):
    g = torch.Generator().manual_seed(seed)
    return {k: (v * (1.0 + eps * (2.0 * torch.rand(v.shape, generator=g) - 1.0)) if (v.dtype.is_floating_point and k.endswith(".weight") and v.dim() >= 4) else v.clone())
            for k, v in sd.items()}


def test_last_iterate_sensitivity(trained, report):
    """why the gates are asserted on the averaged checkpoint (r03 left this "inferred, not isolated"): the sensitivity of a
    checkpoint to half an fp16 ulp.  Every conv weight is multiplied by (1 + 2^-12 xi), xi uniform in [-1, 1) -- what ANY
    fp16-weight evaluation, the reference's own autocast included, does to a raw checkpoint -- and the HIP path (fp32 islands)
    codes the same frames with both.  Last iterate: measured -0.008 ... -0.031 dB, one sign, and the reference's own two paths
    (fp32 / AMP emulation) 0.04 dB apart on it: a statement about that checkpoint, not about anybody's kernels.  Averaged
    checkpoint: under 0.005 dB.  Asserted: the averaged checkpoint is at least as insensitive as the gate needs (< 0.01 dB)."""
    from tdvc_amd.model import VideoCompressor
    from tdvc_amd.synth import make_gop, ref_list

    def hip(sd):
        m = VideoCompressor()
        m.load_state_dict(sd, strict=True)
        return m.cuda().eval()

    out = {}
    for name, sd, base in (("last iterate", trained.last_sd, trained.last), ("averaged", trained.swa_sd, trained.swa_hip)):
        ds = []
        for seed_p in (1, 2):
            pert = hip(_perturbed(sd, 2.0 ** -12, seed_p))
            for seed in (1234, 1235):
                g = make_gop(seed, 2, 256, 256)
                x, refs = g[1:2].cuda(), ref_list([g[0:1]]).cuda()
                r0, _ = _code_frame(base, x, refs, False)
                r1, _ = _code_frame(pert, x, refs, False)
                ds.append(psnr(r1.cpu(), g[1:2]) - psnr(r0.cpu(), g[1:2]))
            del pert
        out[name] = ds
        report(f"[sensitivity to w * (1 + 2^-12 xi), HIP path, 256x256] {name}: dPSNR " + " ".join(f"{d:+.4f}" for d in ds))
    g = make_gop(1234, 2, 256, 256)
    refs = ref_list([g[0:1]])
    ra, _ = _code_frame(trained.last_amp, g[1:2], refs, True)
    last_f32 = type(trained.swa_fp32)().eval()
    last_f32.load_state_dict(trained.last_sd, strict=True)
    rf, _ = _code_frame(last_f32, g[1:2], refs, False)
    report(f"[last iterate, 256x256] the reference's own two paths: AMP emulation {psnr(ra, g[1:2]):.4f} dB, fp32 {psnr(rf, g[1:2]):.4f} dB ({psnr(ra, g[1:2]) - psnr(rf, g[1:2]):+.4f})")
    assert max(abs(d) for d in out["averaged"]) < 0.01
    assert max(abs(d) for d in out["last iterate"]) < 0.1
