"""Generates tests/golden/msssim.npz by importing the REFERENCE's own main/model/ms_ssim_torch.py (this container only;
/root/reference never travels).  Inputs are regenerated from seeds by tests (tdvc_amd.synth.make_gop + a seeded
distortion), so only the case table and the reference's outputs are stored."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from main.model.ms_ssim_torch import ms_ssim, ssim  # noqa: E402  (the reference)
from tests.helpers_metrics import CASES, make_pair  # noqa: E402

out = {}
for i, case in enumerate(CASES):
    X, Y = make_pair(*case)
    out[f"ms_{i}"] = ms_ssim(X, Y, data_range=1.0, size_average=False).numpy()
    s, cs = ssim(X, Y, data_range=1.0, size_average=False, full=True)
    out[f"ssim_{i}"] = s.numpy()
    out[f"cs_{i}"] = cs.numpy()
    out[f"ms255_{i}"] = ms_ssim(X * 255, Y * 255, data_range=255, size_average=True).numpy()
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "msssim.npz"), **out)
print({k: v.tolist() for k, v in out.items()})
