"""Where does the one-signed PSNR difference between the HIP path and an oracle on a RAW trained checkpoint come from?

Trains exactly like tests/test_model_gpu.py::_train_to_operating_point (deterministic: the same weights on every run), then codes
P-frames at 256x256 (and one 512x768 frame) with

  A  the HIP path: raw weights, both coder modes;
  B  the fp32 oracle: raw weights / every 4-D weight fp16-rounded / only the autocast regions' conv weights fp16-rounded;
  C  the AMP-emulating oracle (raw weights): full emulation / element-wise ops kept in fp32 (convs only);
  D  the HIP path on PERTURBED checkpoints w * (1 + eps * xi), xi uniform in [-1, 1), eps = 2^-12 (half an fp16 ulp: a second,
     independent "rounding") and 2^-11: if every such perturbation costs the HIP path what the oracles lose, the loss is a
     property of the checkpoint (the exact trained point is special), not of anybody's arithmetic;
  E  the HIP path on the EMA of the last 100 training iterates (a checkpoint neither path was ever evaluated at) against the
     oracles on the same checkpoint.

python tools/parity_diag.py [iters]"""
import copy
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.tdvc_ref import VideoCompressor as Ref  # noqa: E402
from oracle.tdvc_ref import blocks as ref_blocks  # noqa: E402
from tdvc_amd import ops  # noqa: E402
from tdvc_amd.model import VideoCompressor  # noqa: E402
from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402
from tdvc_amd.train import TrainStep  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
EMA_LAST = 100
torch.manual_seed(1111)
ops.DETERMINISTIC = True
net = VideoCompressor()
fill_parameters(net)
net = net.cuda().train()
step = TrainStep(net, train_lambda=256.0, lr=2e-4, loss_scale=128.0)
pool, cursor, ema = [], 0, None
for it in range(iters):
    while len(pool) < 4:
        gop = make_gop(5000 + cursor, 7, 256, 256)
        cursor += 1
        for t in range(1, 7):
            pool.append((gop[t:t + 1], ref_list([gop[k:k + 1] for k in range(0, t)][-4:] if t > 3 else [gop[k:k + 1] for k in range(0, t)])))
    batch, pool = pool[:4], pool[4:]
    log = step(torch.cat([b[0] for b in batch]).cuda(), torch.cat([b[1] for b in batch]).cuda())
    if it >= iters - EMA_LAST:
        sd = {k: v.detach().double().clone() for k, v in net.state_dict().items() if v.dtype.is_floating_point}
        ema = sd if ema is None else {k: ema[k] + sd[k] for k in sd}
ops.DETERMINISTIC = False
net = net.eval()
raw_sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
ema_sd = {k: (ema[k] / EMA_LAST).float().cpu() if k in ema else v.clone() for k, v in raw_sd.items()}
print(f"trained {iters} iterations: rd_loss {log['rd_loss']:.4f} bpp {log['bpp_res'] + log['bpp_mv']:.4f}", flush=True)

psnr = lambda a, b: 10 * math.log10(1.0 / float(((a - b) ** 2).mean()))


def rounded(sd, which):
    """which: 'all' = every 4-D weight; 'amp' = the convs of the reference's autocast regions (not the coders, not the DCN's main weight)"""
    out = {}
    for k, v in sd.items():
        conv_w = k.endswith(".weight") and v.dim() >= 4
        if which == "amp":
            conv_w = conv_w and not k.startswith(("mvCoder.", "resCoder.")) and k != "mcnet.dconv.weight"
        out[k] = v.half().float() if conv_w else v.clone()
    return out


def hip_model(sd):
    m = VideoCompressor()
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


def oracle(sd, amp=False):
    r = Ref().eval()
    r.load_state_dict(sd, strict=True)
    r.amp_emulation = amp
    return r


def perturbed(sd, eps, seed):
    g = torch.Generator().manual_seed(seed)
    return {k: (v * (1.0 + eps * (2.0 * torch.rand(v.shape, generator=g) - 1.0)) if (v.dtype.is_floating_point and k.endswith(".weight") and v.dim() >= 4) else v.clone())
            for k, v in sd.items()}


def frames():
    for (H, W, seeds) in ((256, 256, (1234, 1235, 1236)), (512, 768, (1234,))):
        for seed in seeds:
            g = make_gop(seed, 2, H, W)
            yield f"{H}x{W}/{seed}", g[1:2], ref_list([g[0:1]])


def code(m, x, refs, amp):
    on_gpu = isinstance(m, VideoCompressor)
    with torch.no_grad():
        r, br, bm = m(x.cuda() if on_gpu else x, refs.cuda() if on_gpu else refs, amp)
    return psnr(r.cpu().float(), x), float(br + bm), r.cpu().float()


def table(title, variants, base_name):
    """variants: name -> (model, amp flag); prints PSNR / bpp per frame relative to `base_name`"""
    print(f"\n== {title} (differences against: {base_name})", flush=True)
    acc = {k: [] for k in variants}
    for tag, x, refs in frames():
        res = {k: code(m, x, refs, amp) for k, (m, amp) in variants.items()}
        b = res[base_name]
        line = f"{tag:14s} base {b[0]:.4f} dB {b[1]:.5f} bpp |"
        for k, v in res.items():
            if k == base_name:
                continue
            line += f" {k}: {v[0] - b[0]:+.4f} dB {v[1] - b[1]:+.5f} bpp ({psnr(v[2], b[2]):.1f} dB) |"
            acc[k].append(v[0] - b[0])
        print(line, flush=True)
    for k, v in acc.items():
        if v:
            print(f"   mean dPSNR {k}: {sum(v) / len(v):+.4f} dB", flush=True)


hip_raw = hip_model(raw_sd)
# A / B / C: one table against the HIP path (fp32 islands) on the raw checkpoint
table("raw trained checkpoint: oracles against the HIP path", {
    "hip_islands": (hip_raw, False),
    "hip_default": (hip_raw, True),
    "fp32(raw)": (oracle(raw_sd), False),
    "fp32(all 4-D rounded)": (oracle(rounded(raw_sd, "all")), False),
    "fp32(AMP-region convs rounded)": (oracle(rounded(raw_sd, "amp")), False),
    "amp(raw)": (oracle(raw_sd, amp=True), True),
}, "hip_islands")

# C': AMP emulation with the element-wise ops kept in fp32: convs return fp32 tensors that hold fp16 values
_orig2, _orig3 = ref_blocks.Conv2d.forward, ref_blocks.Conv3d.forward
ref_blocks.Conv2d.forward = lambda self, x: _orig2(self, x).float() if ref_blocks._Amp.on else _orig2(self, x)
ref_blocks.Conv3d.forward = lambda self, x: _orig3(self, x).float() if ref_blocks._Amp.on else _orig3(self, x)
table("AMP emulation, convs only (element-wise ops in fp32)", {"hip_islands": (hip_raw, False), "amp(convs only)": (oracle(raw_sd, amp=True), True)}, "hip_islands")
ref_blocks.Conv2d.forward, ref_blocks.Conv3d.forward = _orig2, _orig3

# D: the HIP path on perturbed checkpoints
var = {"hip(raw)": (hip_raw, False)}
for eps_name, eps in (("2^-12", 2.0 ** -12), ("2^-11", 2.0 ** -11)):
    for seed in (1, 2):
        var[f"hip(w*(1+{eps_name} xi{seed}))"] = (hip_model(perturbed(raw_sd, eps, seed)), False)
table("HIP path, fp32 islands: perturbed checkpoints against the raw one", var, "hip(raw)")

# E: a checkpoint neither path was evaluated at (mean of the last iterates)
hip_ema = hip_model(ema_sd)
table(f"EMA checkpoint (mean of the last {EMA_LAST} iterates)", {
    "hip_islands": (hip_ema, False),
    "hip_default": (hip_ema, True),
    "fp32(raw)": (oracle(ema_sd), False),
    "fp32(all 4-D rounded)": (oracle(rounded(ema_sd, "all")), False),
    "amp(raw)": (oracle(ema_sd, amp=True), True),
}, "hip_islands")
