"""offset statistics of the fused DCN call inside a 1080p P-frame of the bench (filler weights, synthetic GOP)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops  # noqa: E402
from tdvc_amd.codec_utils import pad  # noqa: E402
from tdvc_amd.model import VideoCompressor  # noqa: E402
from tdvc_amd.synth import fill_parameters, make_gop, ref_list  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
net = VideoCompressor(); fill_parameters(net); net = net.cuda().eval()
g = make_gop(1234, 3, H, W).cuda()
seen = {}
orig = ops.dcn_fused
def spy(x, om, pc, out, groups=8, **kw):
    seen["om"] = om.t[..., :om.C].float().clone()
    seen["G"] = groups
    return orig(x, om, pc, out, groups=groups, **kw)
ops.dcn_fused = spy
import tdvc_amd.model.modules as M
if hasattr(M, "ops"):
    M.ops.dcn_fused = spy
with torch.no_grad():
    refs = [pad(g[0:1], 64)]
    net(pad(g[1:2], 64), ref_list(refs), True)
om, G = seen["om"], seen["G"]
off = om[..., :18 * G].abs()
print("offset |.|: mean %.3f  p50 %.3f  p90 %.3f  p99 %.3f  max %.3f" % (float(off.mean()), float(off.median()), float(off.flatten().kthvalue(int(0.9 * off.numel())).values),
      float(off.flatten().kthvalue(int(0.99 * off.numel())).values), float(off.max())))
for R in (2, 3, 4, 5, 6, 8):
    print(f"  fraction of offsets with |.| < {R}: {float((off < R).float().mean()):.4f}")
# spread of the integer sample position across the 8 groups of a pixel (coalescing of the current kernel)
o = om[..., :18 * G].reshape(*om.shape[:3], G, 9, 2)
fl = torch.floor(o)
same = (fl == fl[..., :1, :, :]).all(dim=-1).all(dim=-2).float().mean()
print("pixels whose 8 groups share the integer offset for every tap: %.4f" % float(same))
