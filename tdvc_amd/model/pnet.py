"""`VideoCompressor` — drop-in for `main/model/pnet.py:15-83` on MI355X.

Same constructor (no arguments), same `forward(input_image, refer_frames, enabled_amp,
is_compress=False)` signature and return tuple, same state-dict keys (`mvCoder, resCoder,
extra_fea, motion_est, mcnet, loopfilter, mcfilter`).  The whole forward runs in hand-written
gfx950 kernels (libtdvc_hip.so): fp16 NHWC activations with fp32 accumulation for every conv
(the reference under AMP computes its convs in fp16 too, and keeps the two coders in fp32 —
here they are fp16-in / fp32-accumulate as well), fp32 for flow, SE gates,
matching and rate terms.  There is no PyTorch/CPU fallback: tensors must live on a HIP device.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..ops import FM
from .coder import MVCoder, ResCoder
from .modules import FeaExtra, FeatureFix, LoopFilter, MCNet, OffsetGen


def _ref_views(refs8: FM, B: int):
    """(r-1, I) of every batch item as views of the converted reference stack (B*4, H, W, 8): items 4b + 3 and 4b"""
    return (FM(refs8.t, refs8.off + 3 * refs8.sn, B, refs8.C, 4 * refs8.sn), FM(refs8.t, refs8.off, B, refs8.C, 4 * refs8.sn))


class VideoCompressor(nn.Module):
    fp32_islands_supported = True

    def __init__(self):
        super().__init__()
        # the reference keeps both coders outside autocast (pnet.py:33,57).  Default here: fp16-in / fp32-accumulate coders
        # (the 30 fps path); `enabled_amp=False` in forward(), or this flag for encode() / decode(), runs them as true fp32
        # islands (fp32 activations + weights on v_mfma_f32_32x32x2_f32): symbols and byte streams then equal the fp32 CPU
        # reference's on identical coder inputs (tests/test_entropy_coding_gpu.py).  NOTE for reference checkpoints: the reference's
        # `enable_amp: True` (cfg/predict.yaml) keeps fp32 coders; here that value selects the fp16-in coders, so the reference-faithful
        # setting is coder_fp32 = True (tools/predict: yaml key `coder_fp32: true` or --coder-fp32), independent of `enabled_amp`
        self.coder_fp32 = False
        # symbol order of the y streams encode() writes and decode() expects: "raster" (compressai's, what the reference's
        # decoder reads) or "wavefront" (an extension: the decoder takes an anti-diagonal per step instead of a position)
        self.stream_order = "raster"
        self.mvCoder = MVCoder(N=128)
        self.resCoder = ResCoder(N=128)
        self.extra_fea = FeaExtra(2)
        self.motion_est = OffsetGen()
        self.mcnet = MCNet(3)
        self.loopfilter = FeatureFix()      # reference-based in-loop filter (sic, pnet.py:23)
        self.mcfilter = LoopFilter()        # multi-frame feature fusion (sic, pnet.py:24)

    # packed-weight cache -------------------------------------------------------------------
    def clear_packed(self):
        for m in self.modules():
            m.__dict__.pop("_packed", None)
            m.__dict__.pop("_tables", None)         # host copies of the coder CDF tables (Cheng2020Anchor._coder_tables)
            m.__dict__.pop("_ar_cache", None)       # wavefront-loop descriptors point at the packed weights

    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        self.clear_packed()
        return r

    def _apply(self, fn, *a, **kw):
        self.clear_packed()
        return super()._apply(fn, *a, **kw)

    # ----------------------------------------------------------------------------------------
    def forward(self, input_image, refer_frames, enabled_amp=True, is_compress=False, trace=None, noise=None):
        noise = noise or {}            # test hook: {"mv": {...}, "res": {...}} of fp32 FMs replaces the coders' training-mode draws
        if not (input_image.is_cuda and refer_frames.is_cuda):
            raise RuntimeError("tdvc_amd.VideoCompressor runs on a HIP device only (no CPU fallback)")
        f32 = (not enabled_amp) or self.coder_fp32
        if f32 and self.training:
            if not self.__dict__.get("_warned_fp32_train"):
                import warnings
                warnings.warn("tdvc_amd: the fp32-island coders are an inference / coding mode; training runs the default "
                              "fp16-in / fp32-accumulate coders (enabled_amp=False ignored in .train())")
                self.__dict__["_warned_fp32_train"] = True
            f32 = False
        training = self.training       # noise quantisation, scale-8 matching, 5-tuple return (pnet.py:80-83); gradients
        # come from the tape of tdvc_amd/autograd.py (`with autograd.record(): model(...)`), not from torch.autograd
        B, _, H, W = input_image.shape
        if H % 64 or W % 64:
            raise RuntimeError(f"input must be padded to a multiple of 64 (got {H}x{W}); see tools/predict.py:51-53")
        dev = input_image.device
        with torch.cuda.device(dev), torch.no_grad():
            x = input_image.float()
            refs = refer_frames.float().reshape(B * 4, 3, H, W)
            cur32 = ops.from_nchw(x, Cpad=4, dtype=torch.float32)
            cur8 = ops.from_nchw(x, Cpad=8)
            refs8 = ops.from_nchw(refs, Cpad=8)                          # (B*4,H,W,8): [I, r-3, r-2, r-1]
            last = refer_frames[:, 3].float()
            ref32 = ops.from_nchw(last, Cpad=4, dtype=torch.float32)
            ref8, iframe8 = _ref_views(refs8, B)                          # frames r-1 and I: strided batch views, not second conversions
            if ops.TAPE is not None:                                     # network inputs carry no gradient
                for t in (cur32, cur8, refs8, ref32, ref8, iframe8):
                    ops.TAPE.mark_input(t)

            feats = FM.empty(B, H, W, 192, device=dev)                   # [f_cur | f_ref | dcn_out]
            npx_ = float(B * H * W)
            f_cur = self.extra_fea.run(cur8, feats.ch(0, 64))
            self.extra_fea.run(ref8, feats.ch(64, 64))
            estmv = self.motion_est.run(feats, cur32, ref32)

            tr_mv = {} if trace is not None else None
            mv_hat, mv_bits = self.mvCoder.run(estmv, training=training, trace=tr_mv, noise=noise.get("mv"), f32=f32)
            coded = {}
            if is_compress:                                      # pnet.py:45-49
                self.mvCoder.update(force=True)
                coded["mv"] = self.mvCoder.compress(estmv, f32=f32)

            xt = FM.empty(B, H, W, 256, device=dev)                      # 4 frames x 64 ch
            pred1 = self.mcnet.run(mv_hat, feats, xt.ch(192, 64))
            pred = FM.empty(B, H, W, 64, device=dev)
            self.mcfilter.run(xt, refs8, pred)
            resid = ops.scale_act_res(f_cur, FM.empty(B, H, W, 64, device=dev), res=pred, res_sign=-1.0)

            tr_res = {} if trace is not None else None
            recon_f, res_bits = self.resCoder.run(resid, training=training, res=pred, trace=tr_res, noise=noise.get("res"), f32=f32)
            if is_compress:                                      # pnet.py:69-73
                self.resCoder.update(force=True)
                coded["res"] = self.resCoder.compress(resid, f32=f32)
                # the reference computes these and drops them (pnet.py:49,73); kept for inspection
                self.last_strings = {k: v["strings"] for k, v in coded.items()}
                self.last_ac_bpp = {k: sum(len(s[0]) for s in v["strings"]) * 8.0 / npx_ for k, v in coded.items()}

            recon = self.loopfilter.run(recon_f, iframe8, training=training, trace=trace)

            npx = float(B * H * W)
            bpp_mv = (mv_bits.sum() / npx).float().view(-1)
            bpp_res = (res_bits.sum() / npx).float().view(-1)
            if trace is not None:
                trace.update(f_cur=f_cur, f_ref=feats.ch(64, 64), estmv=estmv, mv_x_hat=mv_hat, pred1=pred1, pred=pred,
                             resid=resid, recon_f=recon_f, mv=tr_mv, res=tr_res)
        if training:
            return recon, bpp_res, bpp_mv, self.mvCoder.aux_loss(), self.resCoder.aux_loss()
        return recon, bpp_res, bpp_mv

    # ---- real encode / decode (the reference only sketches it: tools/utils/encoder.py, decoder.py) ----
    def _prepare(self, refer_frames, input_image=None):
        B, _, _, H, W = refer_frames.shape
        if H % 64 or W % 64:
            raise RuntimeError(f"frames must be padded to a multiple of 64 (got {H}x{W})")
        dev = refer_frames.device
        refs8 = ops.from_nchw(refer_frames.float().reshape(B * 4, 3, H, W), Cpad=8)
        last = refer_frames[:, 3].float()
        ref8, iframe8 = _ref_views(refs8, B)
        feats = FM.empty(B, H, W, 192, device=dev)
        self.extra_fea.run(ref8, feats.ch(64, 64))
        return B, H, W, dev, refs8, last, iframe8, feats

    def _reconstruct(self, mv_y_hat: FM, res_y_hat_fn, feats: FM, refs8: FM, iframe8: FM):
        """decoder-side reconstruction from the decoded motion latents; `res_y_hat_fn(pred)` supplies the residual latents
        (the encoder codes them against this very prediction, the decoder reads them from the stream)"""
        B, H, W, dev = feats.N, feats.H, feats.W, feats.t.device
        mv_hat = self.mvCoder.run_g_s(mv_y_hat)
        xt = FM.empty(B, H, W, 256, device=dev)
        self.mcnet.run(mv_hat, feats, xt.ch(192, 64))
        pred = FM.empty(B, H, W, 64, device=dev)
        self.mcfilter.run(xt, refs8, pred)
        recon_f = self.resCoder.run_g_s(res_y_hat_fn(pred), res=pred)
        return self.loopfilter.run(recon_f, iframe8, training=False)

    @torch.no_grad()
    def encode(self, input_image, refer_frames):
        """-> {"strings": [mv_y, mv_z, res_y, res_z] (lists over the batch), "shapes": z shapes, "recon": (B,3,H,W)}.
        The reconstruction is the DECODER's: it is built from the coded symbols, so a closed-loop encoder and the decoder
        stay bit-identical."""
        assert not self.training
        B, H, W, dev, refs8, last, iframe8, feats = self._prepare(refer_frames)
        x = input_image.float()
        f_cur = self.extra_fea.run(ops.from_nchw(x, Cpad=8), feats.ch(0, 64))
        estmv = self.motion_est.run(feats, ops.from_nchw(x, Cpad=4, dtype=torch.float32), ops.from_nchw(last, Cpad=4, dtype=torch.float32))
        self.mvCoder.update()
        self.resCoder.update()
        f32 = self.coder_fp32
        mv = self.mvCoder.compress(estmv, f32=f32, order=self.stream_order, defer=True)       # range coding on worker threads
        cat = lambda dbg: FM(torch.cat([d["y_hat"].t for d in dbg], 0))
        out = {}

        def res_y_hat(pred):
            resid = ops.scale_act_res(f_cur, FM.empty(B, H, W, 64, device=dev), res=pred, res_sign=-1.0)
            out["res"] = self.resCoder.compress(resid, f32=f32, order=self.stream_order, defer=True)
            return cat(out["res"]["_debug"])
        recon = self._reconstruct(cat(mv["_debug"]), res_y_hat, feats, refs8, iframe8)
        rs = out["res"]
        mvs, rss = mv["strings"].result(), rs["strings"].result()                     # join the range coders
        return {"strings": [mvs[0], mvs[1], rss[0], rss[1]], "shapes": [mv["shape"], rs["shape"]], "recon": recon}

    @torch.no_grad()
    def decode(self, strings, shapes, refer_frames):
        """inverse of encode(): strings [mv_y, mv_z, res_y, res_z] + z shapes + the reference list -> (B,3,H,W)"""
        assert not self.training
        B, H, W, dev, refs8, last, iframe8, feats = self._prepare(refer_frames)
        self.mvCoder.update()
        self.resCoder.update()
        f32 = self.coder_fp32
        so = self.stream_order
        mv = self.mvCoder.decompress([strings[0], strings[1]], shapes[0], f32=f32, order=so)
        return self._reconstruct(mv["y_hat"], lambda pred: self.resCoder.decompress([strings[2], strings[3]], shapes[1], synth=False, f32=f32, order=so)["y_hat"],
                                 feats, refs8, iframe8)
