import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tdvc_amd import ops
from util import randn, rnd16, to_fm, fm_to_cpu
C, H, W = 128, 96, 112
x = rnd16(randn(1, C, H, W, seed=61))
gamma = rnd16(torch.rand(C, C, generator=torch.Generator().manual_seed(62)) * 0.02 + 0.1 * torch.eye(C))
beta = torch.rand(C, generator=torch.Generator().manual_seed(63)) + 0.5
xf = to_fm(x, ops)
pc = ops.pack_conv(gamma.view(C, C, 1, 1), beta, stride=1, pad=0)
for name, kw in (("plain1x1", dict()), ("square", dict(square=True)), ("gdn", dict(square=True, gdn=ops.GDN_FWD, aux=xf))):
    y = fm_to_cpu(ops.conv(xf, pc, **kw))
    xin = rnd16(x * x) if kw.get("square") else x
    ref = F.conv2d(xin, gamma.view(C, C, 1, 1), beta)
    if "gdn" in kw: ref = x * torch.rsqrt(ref)
    bad = (y - ref).abs() > (4e-3 + 4e-3 * ref.abs())
    print(name, "bad frac", float(bad.float().mean()), "by channel-block", [round(float(bad[:, i*32:(i+1)*32].float().mean()), 3) for i in range(4)],
          "by col-tile", [round(float(bad[..., i*32:(i+1)*32].float().mean()), 3) for i in range(4)],
          "by row%16", [round(float(bad[:, :, i::16].float().mean()), 3) for i in range(0, 16, 2)])
