"""Top-level `_ext` module — drop-in for the reference's DCNv2 extension
(`main/utils/dcnv2/src/vision.cpp:3-8`, imported as `import _ext as _backend` by
`main/utils/dcnv2/dcn_v2_amp.py:13`).  Same function names and arity; the work is done by
hand-written gfx950 kernels in tdvc_amd/libtdvc_hip.so through the C-ABI of include/tdvc_hip.h.
The PS-RoI pooling entry points are off the P-frame path and raise.
"""
from tdvc_amd.ops import dcn_v2_forward  # noqa: F401
from tdvc_amd.dcn_ext import dcn_v2_backward  # noqa: F401


def dcn_v2_psroi_pooling_forward(*args, **kwargs):
    raise RuntimeError("dcn_v2_psroi_pooling_forward: not on the TDVC P-frame path; not provided by tdvc_amd")


def dcn_v2_psroi_pooling_backward(*args, **kwargs):
    raise RuntimeError("dcn_v2_psroi_pooling_backward: not on the TDVC P-frame path; not provided by tdvc_amd")
