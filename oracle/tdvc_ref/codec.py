"""CPU ORACLE (test infrastructure) — `VideoCompressor` of `main/model/pnet.py:15-83`.

fp32 CPU restatement of the per-P-frame encode + reconstruct path; the autocast regions of
the reference are no-ops on CPU, the one reduced-precision quirk that survives is the
DCN's unconditional fp16 output (see blocks.DCN).  State-dict keys match the reference
(`mvCoder, resCoder, extra_fea, motion_est, mcnet, loopfilter, mcfilter`).

`amp_emulation = True` (off by default) makes `forward(..., enabled_amp=True)` compute what the
reference computes on a GPU with `enable_amp: True` (`cfg/predict.yaml:7`): the three autocast
regions of `pnet.py:27-31,51-55,75-78` run under `blocks.amp_region` (fp16 convolutions and fp16
intermediates with torch's own type promotion), both coders stay untouched fp32 islands
(`pnet.py:33-49,57-73`: `estmv.float()`, `input_residual.float()`).  This is the function a raw
fp32 checkpoint defines for the reference's users, and the one the HIP path's fp32-island mode
claims to reproduce; with the switch off the class is the plain fp32 CPU path.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .blocks import FeaExtra, FeatureFix, LoopFilter, MCNet, OffsetGen, amp_region
from .coder import MVCoder, ResCoder


def bpp_from_likelihoods(liks, num_pixels):
    """`main/model/pnet.py:38-43`."""
    return sum(torch.log(l).sum() / (-math.log(2) * num_pixels) for l in liks.values())


class VideoCompressor(nn.Module):
    def __init__(self):
        super().__init__()
        self.mvCoder = MVCoder(N=128)
        self.resCoder = ResCoder(N=128)
        self.extra_fea = FeaExtra(2)
        self.motion_est = OffsetGen()
        self.mcnet = MCNet(3)
        self.loopfilter = FeatureFix()      # in-loop filter (sic, `pnet.py:23`)
        self.mcfilter = LoopFilter()        # multi-frame fusion (sic, `pnet.py:24`)
        self.amp_emulation = False          # see the module docstring

    def forward(self, input_image, refer_frames, enabled_amp=False, is_compress=False, trace=None, noise=None):
        noise = noise or {}            # test hook: {"mv": {...}, "res": {...}} replaces the coders' training-mode draws
        amp = bool(self.amp_emulation and enabled_amp)
        with amp_region(amp):                                         # pnet.py:27-31
            ref = refer_frames[:, -1].clone()
            f_cur = self.extra_fea(input_image)
            f_ref = self.extra_fea(ref)
            estmv = self.motion_est(f_cur, f_ref, input_image, ref)

        mv = self.mvCoder(estmv.float(), noise.get("mv"))
        mv_aux = self.mvCoder.aux_loss()
        N, _, H, W = input_image.shape
        npx = N * H * W
        bpp_mv = bpp_from_likelihoods(mv["likelihoods"], npx)
        strings = {}
        if is_compress:
            self.mvCoder.eval()
            self.mvCoder.update(force=True)
            strings["mv"] = self.mvCoder.compress(estmv.float())

        with amp_region(amp):                                         # pnet.py:51-55
            pred1 = self.mcnet(mv["x_hat"], f_ref)
            pred = self.mcfilter(pred1, refer_frames)
            resid = f_cur - pred

        rs = self.resCoder(resid.float(), noise.get("res"))
        res_aux = self.resCoder.aux_loss()
        bpp_res = bpp_from_likelihoods(rs["likelihoods"], npx)
        if is_compress:
            self.resCoder.eval()
            self.resCoder.update(force=True)
            strings["res"] = self.resCoder.compress(resid.float())

        with amp_region(amp):                                         # pnet.py:75-78
            recon_f = pred + rs["x_hat"]
            recon = self.loopfilter(recon_f, refer_frames).clamp(0.0, 1.0)
        recon = recon.float()          # a no-op on the fp32 path; under AMP emulation the fp16 picture the reference returns, widened

        if trace is not None:
            trace.update(f_cur=f_cur, f_ref=f_ref, estmv=estmv, mv_x_hat=mv["x_hat"], pred1=pred1, pred=pred,
                         resid=resid, res_x_hat=rs["x_hat"], recon_f=recon_f, recon=recon,
                         mv_dbg=mv["_debug"], res_dbg=rs["_debug"], strings=strings)
        if self.training:
            return recon, bpp_res.view(-1), bpp_mv.view(-1), mv_aux, res_aux
        return recon, bpp_res.view(-1), bpp_mv.view(-1)
