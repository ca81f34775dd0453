"""ctypes binding of libtdvc_hip.so (C-ABI declared in include/tdvc_hip.h).

The product path FAILS LOUDLY when the library is missing: there is no CPU or eager-PyTorch
fallback anywhere in `tdvc_amd`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtdvc_hip.so")

MAX_TAPS = 49
F16, F32 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_CLAMP01, ACT_SIGMOID = 0, 1, 2, 3, 4
GDN_NONE, GDN_FWD, GDN_INV = 0, 1, 2
OUT_NHWC, OUT_SHUFFLE2, OUT_NCHW_F32 = 0, 1, 2


class FMapDesc(C.Structure):
    _fields_ = [("p", C.c_void_p), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("sn", C.c_int64), ("sp", C.c_int32), ("dtype", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("x", FMapDesc), ("y", FMapDesc), ("w", C.c_void_p), ("bias", C.c_void_p),
                ("cout", C.c_int32), ("ntaps", C.c_int32),
                ("tap_dy", C.c_int8 * MAX_TAPS), ("tap_dx", C.c_int8 * MAX_TAPS),
                ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("ck", C.c_int32), ("square_input", C.c_int32), ("gdn", C.c_int32),
                ("aux", FMapDesc), ("act", C.c_int32), ("slope", C.c_float),
                ("round_before_act", C.c_int32), ("res", FMapDesc), ("res2", FMapDesc), ("out_mode", C.c_int32), ("s2d", C.c_int32),
                ("bcast_T", C.c_int32), ("bcast_slope", C.c_float), ("chan_sum", C.c_void_p)]


class DcnDesc(C.Structure):
    _fields_ = [("x", FMapDesc), ("om", FMapDesc), ("y", FMapDesc), ("w", C.c_void_p), ("bias", C.c_void_p),
                ("groups", C.c_int32), ("act", C.c_int32), ("slope", C.c_float), ("round_before_act", C.c_int32), ("x_planar", C.c_void_p)]


class ConvPairDesc(C.Structure):
    _fields_ = [("x", FMapDesc), ("y", FMapDesc), ("w", C.c_void_p), ("bias", C.c_void_p), ("act1", C.c_int32), ("slope1", C.c_float),
                ("act2", C.c_int32), ("slope2", C.c_float), ("add_input", C.c_int32), ("res2", FMapDesc)]


class WgradReduceJob(C.Structure):
    _fields_ = [("work", C.c_void_p), ("bwork", C.c_void_p), ("row_off", C.c_void_p), ("chan_off", C.c_void_p), ("tap_off", C.c_void_p),
                ("bias_index", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("scale", C.c_float),
                ("nworkers", C.c_int32), ("cow", C.c_int32), ("ciw", C.c_int32), ("ntaps", C.c_int32), ("cout", C.c_int32), ("cin", C.c_int32),
                ("wblocks", C.c_int32), ("nblocks", C.c_int32)]


_P = C.c_void_p
_FM = C.POINTER(FMapDesc)
_i, _f, _i64 = C.c_int, C.c_float, C.c_int64

# name -> (restype, argtypes); every symbol declared in include/tdvc_hip.h
SIGNATURES = {
    "tdvc_abi_version": (_i, []),
    "tdvc_conv_chan_sum_rows": (_i, [C.POINTER(ConvDesc)]),
    "tdvc_last_error": (C.c_char_p, []),
    "tdvc_last_conv_kernel": (C.c_char_p, []),
    "tdvc_prepare_device": (_i, []),
    "tdvc_conv_plan": (_i, [_i, _i, _i, _i]),
    "tdvc_conv_packed_bytes": (_i64, [_i, _i, _i, _i]),
    "tdvc_pack_conv_weights": (_i, [_P, _i, _i, _i, _i, _i, _i, _P, _P, _i, _P]),
    "tdvc_conv2d": (_i, [C.POINTER(ConvDesc), _P]),
    "tdvc_conv2d_f32": (_i, [C.POINTER(ConvDesc), _P]),
    "tdvc_pack_conv_weights_indexed": (_i, [_P] * 7 + [_i] * 4 + [_P, _P]),
    "tdvc_pack_conv_weights_indexed_f32": (_i, [_P] * 7 + [_i] * 4 + [_P, _P]),
    "tdvc_pack_job_blocks": (_i64, [_i, _i, _i, _i]),
    "tdvc_pack_conv_weights_batch": (_i, [_P, _P, _i, _i, _P]),
    "tdvc_act_backward": (_i, [_FM, _FM, _FM, _i, _f, _FM, _P]),
    "tdvc_pixel_unshuffle": (_i, [_FM, _FM, _P]),
    "tdvc_bias_grad_work_floats": (_i64, [_i, _i]),
    "tdvc_bias_grad": (_i, [_FM, _i, _P, _f, _P, _P, _i64, _P]),
    "tdvc_copy_cast": (_i, [_FM, _FM, _P]),
    "tdvc_clamp01_backward": (_i, [_FM, _FM, _P]),
    "tdvc_gate_backward_work_floats": (_i64, [_i, _i]),
    "tdvc_gate_backward": (_i, [_FM, _FM, _P, _FM, _P, _P, _i64, _P]),
    "tdvc_se_gate_backward": (_i, [_P, _i, _f, _i, _i, _i] + [_P] * 6 + [_f] + [_P] * 6),
    "tdvc_bcast_channel_add": (_i, [_FM, _P, _f, _P]),
    "tdvc_add_flow_backward": (_i, [_FM, _FM, _P]),
    "tdvc_bcast_add_act_backward": (_i, [_FM, _FM, _FM, _f, _P]),
    "tdvc_upsample2x_backward": (_i, [_FM, _FM, _P]),
    "tdvc_resize_bilinear_backward": (_i, [_FM, _FM, _P, _P]),
    "tdvc_spynet_level_input_backward": (_i, [_FM, _FM, _FM, _FM, _FM, _P]),
    "tdvc_sigmoid_f32": (_i, [_P, _P, _i64, _P]),
    "tdvc_sigmoid_backward_f32": (_i, [_P, _P, _i64, _P]),
    "tdvc_axpy_f32": (_i, [_P, _P, _f, _i64, _P]),
    "tdvc_match_gather_backward": (_i, [_FM, _FM, _P, _i, _i, _i, _FM, _FM, _FM, _P]),
    "tdvc_eb_backward": (_i, [_FM, _P, _FM, _f, _FM, _P, _P]),
    "tdvc_eb_pack": (_i, [_P, _P, _P, _i, _P]),
    "tdvc_eb_param_chain": (_i, [_P, _P, _P, _f, _i, _P]),
    "tdvc_eb_aux": (_i, [_P, _P, _f, _P, _P, _i, _P]),
    "tdvc_gc_backward": (_i, [_FM, _FM, _FM, _f, _FM, _FM, _P]),
    "tdvc_dcn_columns": (_i, [_FM, _FM, _i, _FM, _P]),
    "tdvc_ar_decode_serial": (_i, [_P, _i64, _P, _i, _P, _P, _FM, _FM, _FM, _FM, _P, _i, _FM, _P, _i, _i, _i, _P, _i, _P, _P, _P]),
    "tdvc_ar_wavefront": (_i, [_P, _i64, _P, _i, _P, _P, _FM, _FM, _FM, _FM, _FM, _P, _i, _FM, _P, _P, _i, _i, _i, _P, _i, _P, _P, _P]),
    "tdvc_ssim_level_work_floats": (_i64, [_i] * 5),
    "tdvc_ssim_level": (_i, [_P, _P, _i, _i, _i, _i, _P, _i, _f, _f, _P, _P, _P, _i64, _P]),
    "tdvc_avgpool2_pad_f32": (_i, [_P, _i64, _i, _i, _P, _P]),
    "tdvc_dcn_col2im_work_floats": (_i64, [_i, _i, _i, _i]),
    "tdvc_dcn_col2im": (_i, [_FM, _FM, _FM, _i, _P, _FM, _P, _i64, _P]),
    "tdvc_dcn_col2im_det": (_i, [_FM, _FM, _FM, _i, _P, _FM, _P, _i64, _P, _P, _P, _i, _P]),
    "tdvc_dcn_far_apply": (_i, [_P, _P, _P, _i, _P, _P]),
    "tdvc_conv_wgrad_work_floats": (_i64, [_i] * 6),
    "tdvc_conv_wgrad": (_i, [_FM, _FM] + [_i] * 6 + [_P] * 5 + [_i, _f, _P, _P, _i64, _P]),
    "tdvc_conv_wgrad_bias": (_i, [_FM, _FM] + [_i] * 6 + [_P] * 5 + [_i, _f, _P, _P, _P, _P, _i64, _P]),
    "tdvc_conv_wgrad_partials": (_i, [_FM, _FM] + [_i] * 6 + [_P] * 5 + [_i, _f, _P, _P, _P, _P, _i64, C.POINTER(WgradReduceJob), _P]),
    "tdvc_wgrad_reduce_batch": (_i, [_P, _P, _i, _i, _P]),
    "tdvc_gdn_backward": (_i, [_FM, _FM, _FM, _i, _FM, _FM, _P]),
    "tdvc_mul2_accumulate": (_i, [_FM, _FM, _FM, _P]),
    "tdvc_dcn_fused": (_i, [C.POINTER(DcnDesc), _P]),
    "tdvc_conv_pair_packed_bytes": (_i64, []),
    "tdvc_pack_conv_pair_weights": (_i, [_P, _P, _P]),
    "tdvc_conv_pair_supported": (_i, [C.POINTER(ConvPairDesc)]),
    "tdvc_conv_pair": (_i, [C.POINTER(ConvPairDesc), _P]),
    "tdvc_dcn_v2_forward_f32": (_i, [_P] * 6 + [_i] * 14 + [_P]),
    "tdvc_dcn_v2_backward_f32": (_i, [_P] * 12 + [_i] * 14 + [_P]),
    "tdvc_nchw_to_fmap": (_i, [_P, _i, _FM, _P]),
    "tdvc_fmap_to_nchw": (_i, [_FM, _i, _P, _P]),
    "tdvc_scale_act_res": (_i, [_FM, _P, _i, _f, _FM, _f, _FM, _FM, _P]),
    "tdvc_add_flow": (_i, [_FM, _FM, _P]),
    "tdvc_bcast_add_act": (_i, [_FM, _FM, _i, _f, _P]),
    "tdvc_channel_sum": (_i, [_FM, _P, _i, _P]),
    "tdvc_se_gate": (_i, [_P, _i, _f, _i, _i, _i, _P, _P, _P, _P, _P, _P]),
    "tdvc_upsample2x": (_i, [_FM, _FM, _P]),
    "tdvc_avgpool2": (_i, [_FM, _FM, _P]),
    "tdvc_spynet_level_input": (_i, [_FM, _FM, _FM, _FM, _FM, _P]),
    "tdvc_resize_bilinear": (_i, [_FM, _FM, _P, _P]),
    "tdvc_avgpool_k_work_floats": (_i64, [_i, _i, _i, _i, _i]),
    "tdvc_avgpool_k": (_i, [_FM, _i, _P, _i, _i, _P, _i64, _P]),
    "tdvc_patch_match": (_i, [_P, _P, _i, _i, _i, _i, _P, _P]),
    "tdvc_match_gather": (_i, [_FM, _FM, _P, _i, _i, _i, _FM, _P]),
    "tdvc_eb_forward": (_i, [_FM, _P, _FM, _FM, _P, _P, _i, _P]),
    "tdvc_gc_forward": (_i, [_FM, _FM, _FM, _P, _P, _i, _P]),
    "tdvc_quantize": (_i, [_FM, _FM, _FM, _P]),
    "tdvc_rans_encode": (_i64, [_P, _P, _i64, _P, C.c_int32, _P, _P, _P, _i64]),
    "tdvc_rans_decode": (_i, [_P, _i64, _P, _i64, _P, C.c_int32, _P, _P, _P]),
    "tdvc_rans_decoder_create": (_P, [_P, _i64]),
    "tdvc_rans_decoder_decode": (_i, [_P, _P, _i64, _P, C.c_int32, _P, _P, _P]),
    "tdvc_rans_decoder_destroy": (None, [_P]),
    "tdvc_pmf_to_quantized_cdf": (_i, [_P, _i, _i, _P]),
    "tdvc_ar_gather": (_i, [_FM, _FM, _P, _i, _FM, _FM, _P]),
    "tdvc_ar_quantize": (_i, [_FM, _FM, _P, _i, _P, _i, _P, _FM, _P, _P, _P]),
    "tdvc_ar_indexes": (_i, [_FM, _P, _i, _P, _i, _i, _i, _P, _P]),
    "tdvc_round_symbols": (_i, [_FM, _P, _P, _P]),
}

_lib = None


class TdvcHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TdvcHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  tdvc_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)        # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().tdvc_last_error().decode("utf-8", "replace")
        raise TdvcHipError(f"{what}: rc={rc}: {msg}")
