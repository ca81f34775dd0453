"""Rate of the default (fp16) coders against the fp32-island coders (= the fp32 CPU oracle's bits on identical coder inputs)
at a TRAINED-like operating point, over many frames: python tools/trained_point_sweep.py [n_gops] [iters]
Prints |dbpp| statistics at 256x256 and 512x768.  (Round 2: median 2-3e-5 bpp at both sizes; 1-2 of 32 frames at 256x256 reach
2.4e-3 -- a flipped symbol in the MOTION latents changes the prediction and with it the residual coder's input -- none of 32
at 512x768 exceeds 4e-4.  Running h_a on the fp32 y does not change this: the excursions are not z flips.)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd.model import VideoCompressor  # noqa: E402
from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402
from tdvc_amd.train import TrainStep  # noqa: E402

n_gops = int(sys.argv[1]) if len(sys.argv) > 1 else 12
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
torch.manual_seed(1111)
net = VideoCompressor()
fill_parameters(net)
net = net.cuda().train()
step = TrainStep(net, train_lambda=256.0, lr=2e-4, loss_scale=128.0)
pool, cursor = [], 0
for it in range(iters):
    while len(pool) < 4:
        gop = make_gop(5000 + cursor, 7, 256, 256)
        cursor += 1
        for t in range(1, 7):
            pool.append((gop[t:t + 1], ref_list([gop[k:k + 1] for k in range(0, t)][-4:] if t > 3 else [gop[k:k + 1] for k in range(0, t)])))
    batch, pool = pool[:4], pool[4:]
    log = step(torch.cat([b[0] for b in batch]).cuda(), torch.cat([b[1] for b in batch]).cuda())
print(f"trained {iters} iterations: bpp {log['bpp_res'] + log['bpp_mv']:.4f}", flush=True)
net = net.eval()
for (H, W) in ((256, 256), (512, 768)):
    res, dps = [], []
    for s_ in range(n_gops):
        g = make_gop(7000 + s_, 3, H, W).cuda()
        refs_l = [g[0:1]]
        for t in (1, 2):
            refs = ref_list(refs_l)
            with torch.no_grad():
                r32, br32, bm32 = net(g[t:t + 1], refs, False)            # fp32 islands
                r16, br16, bm16 = net(g[t:t + 1], refs, True)
            b32 = float(br32 + bm32)
            res.append(float(br16 + bm16) - b32)
            ps = lambda r: float(10 * torch.log10(1.0 / ((r - g[t:t + 1]) ** 2).mean()))
            dps.append(ps(r16) - ps(r32))
            refs_l.append(r32)
    a = [abs(v) for v in res]
    print(f"{H}x{W}: n={len(a)} mean|dbpp| {statistics.mean(a):.5f} median {statistics.median(a):.5f} max {max(a):.5f} over 0.001: {sum(v > 1e-3 for v in a)}; "
          f"max|dPSNR| {max(abs(v) for v in dps):.4f} dB (island bpp ~{b32:.3f})", flush=True)
