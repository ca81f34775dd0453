import os
import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd.model import VideoCompressor
from tdvc_amd.synth import fill_parameters, make_gop, ref_list
from tdvc_amd.codec_utils import pad
H, W = int(sys.argv[1]), int(sys.argv[2])
net = VideoCompressor(); fill_parameters(net); net = net.cuda().eval()
g = make_gop(1234, 3, H, W).cuda()
refs = [pad(g[0:1], 64)]
x = pad(g[1:2], 64)
with torch.no_grad():
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        enc = net.encode(x, ref_list(refs))
        torch.cuda.synchronize(); t1 = time.time()
        shapes = [enc["shapes"][0], enc["shapes"][1]]
        rec = net.decode(enc["strings"], shapes, ref_list(refs))
        torch.cuda.synchronize(); t2 = time.time()
        print(f"{H}x{W}: encode {t1 - t0:.3f} s, decode {t2 - t1:.3f} s, bytes {sum(len(s[0]) for s in enc['strings'])}, equal {torch.equal(rec, enc['recon'])}", flush=True)
