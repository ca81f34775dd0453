"""seeded image pairs for the MS-SSIM parity tests (shared by the golden generator and the tests)"""
import torch

from tdvc_amd import synth

# (seed, N, H, W, noise sigma): sizes chosen so that the pyramid meets odd extents (the padded pooling) and the smallest
# level is still >= the 11-tap window; 256x256 and 1080p-like aspect included
CASES = [(11, 2, 256, 256, 0.02), (12, 1, 180, 200, 0.05), (13, 3, 177, 211, 0.01), (14, 1, 360, 636, 0.08), (15, 1, 192, 176, 0.0)]


def make_pair(seed, N, H, W, sigma):
    gop = synth.make_gop(seed, max(N, 2), H, W).float()
    X = gop[:N].contiguous()
    g = torch.Generator().manual_seed(seed)
    Y = (X + sigma * torch.randn(X.shape, generator=g) + 0.03 * torch.sin(torch.arange(W, dtype=torch.float32) / 7.0)).clamp(0, 1)
    if sigma == 0.0:
        Y = (X * 0.9 + 0.05).contiguous()
    return X, Y.contiguous()
