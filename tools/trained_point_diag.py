"""Where does the PSNR / rate difference against the fp32 CPU oracle come from at a trained-like operating point?
Trains like tests/test_model_gpu.py::_train_to_operating_point, then per frame: FeatureFix patch indices equal?, stage-wise
relative L2 of the trace against the oracle's, symbol flips.  python tools/trained_point_diag.py [iters]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.tdvc_ref import VideoCompressor as Ref  # noqa: E402
from tdvc_amd.model import VideoCompressor  # noqa: E402
from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402
from tdvc_amd.train import TrainStep  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
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
net = net.eval()
ref = Ref().eval()
ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=True)
psnr = lambda a, b: 10 * math.log10(1.0 / float(((a - b) ** 2).mean()))
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-12))
for (H, W) in ((256, 256), (512, 768)):
    g = make_gop(1234, 3, H, W)
    refs_l = [g[0:1]]
    for t in (1, 2):
        refs = ref_list(refs_l)
        tr_o, tr_g = {}, {}
        with torch.no_grad():
            ro, bro, bmo = ref(g[t:t + 1], refs, False, trace=tr_o)
            rg, brg, bmg = net(g[t:t + 1].cuda(), refs.cuda(), True, trace=tr_g)
        idx_eq = torch.equal(tr_g["ff_idx"].cpu().long(), ref.loopfilter.last_match_index)
        st = {k: rel(tr_g[k].to_nchw().cpu(), tr_o[k].float()) for k in ("f_cur", "estmv", "mv_x_hat", "pred1", "pred", "resid", "recon_f")}
        fl = {c: float((tr_g[c]["y_hat"].to_nchw().cpu() != tr_o[c + "_dbg"]["y_hat"]).float().mean()) for c in ("mv", "res")}
        print(f"{H}x{W} frame {t}: dPSNR {psnr(rg.cpu(), g[t:t+1]) - psnr(ro, g[t:t+1]):+.4f} dB PSNR(gpu, oracle) {psnr(rg.cpu(), ro):.2f} dB dbpp {float(brg + bmg) - float(bro + bmo):+.5f} | "
              f"patch indices equal {idx_eq} | flips mv {fl['mv']:.4f} res {fl['res']:.4f} | " + " ".join(f"{k}={v:.1e}" for k, v in st.items()), flush=True)
        refs_l.append(ro)
