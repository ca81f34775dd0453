"""Where the real encode of one 1080p P-frame spends its time: the GPU part of compress() (transforms + wavefront context
loop), the device -> host transfer of symbols / indexes, the host range coder.  python3 tools/time_codec_phases.py [H W]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops  # noqa: E402
from tdvc_amd.codec_utils import pad  # noqa: E402
from tdvc_amd.model import VideoCompressor  # noqa: E402
from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
net = VideoCompressor()
fill_parameters(net)
net = net.cuda().eval()
g = make_gop(1234, 3, H, W).cuda()
refs = ref_list([pad(g[0:1], 64)])
x = pad(g[1:2], 64)
sync = torch.cuda.synchronize
with torch.no_grad():
    for it in range(3):
        sync(); t0 = time.time()
        enc = net.encode(x, refs)
        sync(); t1 = time.time()
        print(f"encode total {1e3 * (t1 - t0):.1f} ms", flush=True)
    # phases of ONE coder's compress() on the motion features
    B, Hp, Wp = 1, x.shape[2], x.shape[3]
    feats = ops.FM.empty(B, Hp, Wp, 192, device="cuda")
    xin = ops.FM(torch.randn(1, Hp, Wp, 64, device="cuda").half() * 0.5)
    c = net.mvCoder
    c.update()
    for it in range(3):
        sync(); t0 = time.time()
        ebt, gct, table = c._coder_tables()
        y32, y16 = c.run_g_a(xin)
        z = c.run_h_a(y16)
        sync(); t1 = time.time()
        med = c.entropy_bottleneck.quantiles.detach()[:, 0, 1].float().contiguous()
        zsym = ops.round_symbols(z, med)
        z_hat = ops.FM((zsym.float() + med).to(torch.float16))
        params = ops.FM.empty(1, y32.H, y32.W, 256, device="cuda")
        c.run_h_s(z_hat, out=params)
        zs = zsym.permute(0, 3, 1, 2).contiguous().cpu().numpy()
        sync(); t2 = time.time()
        zidx = np.broadcast_to(np.arange(128, dtype=np.int32)[:, None, None], zs.shape[1:])
        zstr = ops.rans_encode(zs[0], zidx, ebt)
        t3 = time.time()
        Hl, Wl = y32.H, y32.W
        steps = c.wavefront_steps(Hl, Wl)
        flat = torch.tensor([p for st in steps for p in st], dtype=torch.int32, device="cuda")
        sizes = np.array([len(st) for st in steps], dtype=np.int32)
        chain = c._ar_chain(Hl, torch.float16, "cuda")
        y_hat = ops.FM.zeros(1, Hl, Wl, 128, device="cuda")
        sym = torch.zeros((Hl, Wl, 128), dtype=torch.int32, device="cuda")
        idx = torch.zeros((Hl, Wl, 128), dtype=torch.int32, device="cuda")
        sync(); t4 = time.time()
        ops.ar_wavefront(None, None, y32, y_hat, params, chain["x1"], chain["pc"], chain["descs"], chain["gp"], flat, sizes, 128, Wl, table, idx, sym)
        t5h = time.time()
        sync(); t5 = time.time()
        s_np, i_np = sym.cpu().numpy(), idx.cpu().numpy()
        t6 = time.time()
        ystr = ops.rans_encode(s_np, i_np, gct)
        t7 = time.time()
        print(f"g_a + h_a {1e3 * (t1 - t0):.2f} | z symbols + h_s + D2H {1e3 * (t2 - t1):.2f} | z rANS {1e3 * (t3 - t2):.2f} ({len(zstr)} B) | "
              f"setup {1e3 * (t4 - t3):.2f} | wavefront loop: host enqueue {1e3 * (t5h - t4):.2f}, until done {1e3 * (t5 - t4):.2f} "
              f"({len(steps)} steps) | D2H {1e3 * (t6 - t5):.2f} | y rANS {1e3 * (t7 - t6):.2f} ({len(ystr)} B, {s_np.size} symbols)", flush=True)
