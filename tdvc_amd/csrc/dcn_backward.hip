// `_ext.dcn_v2_backward` for gfx950 (fp32 NCHW) — placeholder until the kernels land.
#include "common.h"

extern "C" int tdvc_dcn_v2_backward_f32(const float* input, const float* weight, const float* bias,
                                        const float* offset, const float* mask, const float* grad_output,
                                        float* grad_input, float* grad_offset, float* grad_mask,
                                        float* grad_weight, float* grad_bias, float* columns,
                                        int B, int C, int H, int W, int Cout,
                                        int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                        int deformable_group, void* stream) {
  tdvc_set_error("tdvc_dcn_v2_backward_f32: not implemented yet");
  return TDVC_ENOSUP;
}
