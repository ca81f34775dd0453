"""dcn_col2im timing vs the spread of the offsets (the scatter window is +-8 pixels around a 16x16 tile)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tdvc_amd import ops
from tdvc_amd.ops import FM

N, H, W, G = 4, 256, 256, 8
torch.manual_seed(0)
x = FM(torch.randn(N, H, W, 64, device="cuda").half())
dcol = FM((torch.randn(N, H, W, 72 * G, device="cuda") * 0.1).half())
for sigma in (0.0, 1.0, 3.0, 8.0, 32.0):
    om = torch.randn(N, H, W, 27 * G + (8 - 27 * G % 8) % 8, device="cuda")
    om[..., :18 * G] *= sigma
    om = FM(om.half())
    dom = FM.zeros(N, H, W, om.C, device="cuda")
    for _ in range(2):
        ops.dcn_col2im(x, om, dcol, G, dom)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.dcn_col2im(x, om, dcol, G, dom)
    e1.record()
    torch.cuda.synchronize()
    print(f"offset sigma {sigma:5.1f}: {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us per call (incl. zero fill + gather)")
