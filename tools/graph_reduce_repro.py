import os
import sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd.model import VideoCompressor
from tdvc_amd.synth import fill_parameters, make_gop, ref_list
from tdvc_amd import train as T, autograd, ops
torch.manual_seed(1000)
m = VideoCompressor(); fill_parameters(m); m = m.cuda().train()
xs, rs = [], []
for i in range(4):
    g = make_gop(1000 + i, 7, 256, 256).cuda()
    xs.append(g[3:4]); rs.append(ref_list([g[0:1], g[1:2], g[2:3]]))
x, refs = torch.cat(xs), torch.cat(rs)
step = T.TrainStep(m, loss_scale=128.0, graph=False)
step(x, refs); step(x, refs)
keep = {}
def fb():
    B, _, H, W = x.shape
    step.buckets.zero()
    with autograd.record(step.loss_scale) as tape:
        recon, bpp_res, bpp_mv, _, _ = m(x, refs, True)
        diff = recon - x.float()
        mse = (diff * diff).mean()
        mse2 = (diff * diff).sum()
        tape.grad_tensor(recon).copy_(diff * (2.0 * step.lam * step.loss_scale / diff.numel()))
        tape.rate_grad = 1.0 / float(B * H * W)
        tape.backward()
    keep.update(recon=recon, diff=diff, mse=mse, mse2=mse2)
gr = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(gr):
    fb()
def rep(tag):
    gr.replay(); torch.cuda.synchronize()
    r = keep["recon"]
    print(tag, "mse", float(keep["mse"]), "sum/n", float(keep["mse2"]) / r.numel(), "eager mse of static recon", float(((r - x.float()) ** 2).mean()),
          "recon finite", bool(torch.isfinite(r).all()), "diff==recon-x", bool(torch.equal(keep["diff"], r - x.float())),
          "nan params", sum(int((~torch.isfinite(p)).sum()) for p in m.parameters()), flush=True)
rep("fresh")
for it in range(3):
    step.optimizer.step()
    rep("opt%d" % it); rep("opt%d" % it)
