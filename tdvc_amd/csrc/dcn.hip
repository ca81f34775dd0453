// Modulated deformable convolution (DCNv2) for gfx950 — the motion-compensation operator
// (SURVEY §8 a11/a12).
//
// (1) tdvc_dcn_fused: fp16 NHWC, sampling + modulation + 64x576 contraction in ONE kernel.
//     The reference materialises `columns` (576 values / pixel = 4.8 GB at 1080p,
//     src/cuda/dcn_v2_cuda.cu:67-92); here each lane bilinearly samples the 8 channels of one
//     deformable group (= one 16-byte NHWC chunk per corner) straight into the MFMA B fragment
//     (B[k = 8*(lane>>5)+j][pixel = lane&31]), so sampled values never leave registers.
// (2) tdvc_dcn_v2_forward_f32: fp32 NCHW operator with `_ext.dcn_v2_forward` semantics.
#include "common.h"

namespace {

struct DcnParams {
  const half_t* x; long x_sn; int x_sp; int H, W;
  const half_t* om; long om_sn; int om_sp;
  FMap y;
  const half_t* w; const float* bias;
  int G; int act; float slope; int round16;
  long npix;   // H*W
};

// bilinear sample of 8 consecutive channels, zero outside (dcn_v2_im2col_cuda.cu:25-54,180)
__device__ __forceinline__ void sample8(const half_t* xg, int H, int W, int sp, float h_im, float w_im, float mask,
                                        half8& out) {
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h_low = (int)hf, w_low = (int)wf;
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h_im - hf, lw = w_im - wf;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    half8 v1 = {0, 0, 0, 0, 0, 0, 0, 0}, v2 = v1, v3 = v1, v4 = v1;
    if (h_low >= 0 && w_low >= 0) v1 = *reinterpret_cast<const half8*>(xg + ((long)h_low * W + w_low) * sp);
    if (h_low >= 0 && w_high <= W - 1) v2 = *reinterpret_cast<const half8*>(xg + ((long)h_low * W + w_high) * sp);
    if (h_high <= H - 1 && w_low >= 0) v3 = *reinterpret_cast<const half8*>(xg + ((long)h_high * W + w_low) * sp);
    if (h_high <= H - 1 && w_high <= W - 1) v4 = *reinterpret_cast<const half8*>(xg + ((long)h_high * W + w_high) * sp);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      acc[j] = (w1 * (float)v1[j] + w2 * (float)v2[j] + w3 * (float)v3[j] + w4 * (float)v4[j]) * mask;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = (half_t)acc[j];
}

// 3x3, stride 1, pad 1, dilation 1, 8 channels / deformable group, Cin = Cout = 8G <= 64.
// One wave = 32 pixels x 64 output channels; 4 waves / block.
__global__ __launch_bounds__(256) void dcn_fused_kernel(const DcnParams p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, r = lane & 31;
  const int n = blockIdx.y;
  const long pix = ((long)blockIdx.x * 4 + wave) * 32 + r;
  const bool valid = pix < p.npix;
  const long pc = valid ? pix : p.npix - 1;
  const int oy = (int)(pc / p.W), ox = (int)(pc % p.W);

  const half_t* xn = p.x + (long)n * p.x_sn;
  const half_t* omp = p.om + (long)n * p.om_sn + pc * p.om_sp;
  const int G = p.G;
  const int steps_per_tap = G / 2;

  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

  const half_t* wbase = p.w + (long)lane * 8;
  const long tile_stride = (long)9 * steps_per_tap * 512;   // halves per 32-row cout tile

  for (int s2 = 0; s2 < steps_per_tap; ++s2) {
    const int g = 2 * s2 + hh;
    const half_t* xg = xn + g * 8;
    // offsets of group g: channels [g*18, g*18+18) = (dh,dw) x 9 taps; mask logits at 18G + g*9 + t
    const half_t* og = omp + g * 18;
    const half_t* mg = omp + 18 * G + g * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const half2v o2 = *reinterpret_cast<const half2v*>(og + 2 * t);
      const float ml = (float)mg[t];
      const float mask = 1.f / (1.f + __expf(-ml));
      const float h_im = (float)(oy - 1 + t / 3) + (float)o2[0];
      const float w_im = (float)(ox - 1 + t % 3) + (float)o2[1];
      half8 b;
      sample8(xg, p.H, p.W, p.x_sp, h_im, w_im, mask, b);
      const half_t* wp = wbase + (long)(t * steps_per_tap + s2) * 512;
      const half8 a0 = *reinterpret_cast<const half8*>(wp);
      const half8 a1 = *reinterpret_cast<const half8*>(wp + tile_stride);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b, acc[1], 0, 0, 0);
    }
  }

  if (!valid) return;
  half_t* yp = reinterpret_cast<half_t*>(p.y.p) + (long)n * p.y.sn + pix * p.y.sp;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int co = mt * 32 + 8 * gq + 4 * hh;
      if (co >= p.y.C) continue;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + co);
      half4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[mt][4 * gq + i] + b4[i];
        if (p.round16) v = (float)(half_t)v;
        v = act_apply(v, p.act, p.slope);
        o[i] = (half_t)v;
      }
      *reinterpret_cast<half4*>(yp + co) = o;
    }
}

// ------------------------------------------------------------------------------------------
// fp32 NCHW operator (`_ext.dcn_v2_forward`)
// block: 64 output pixels; loops over input channels, sampling kh*kw taps of one channel into
// LDS, then every thread accumulates its (pixel, cout-subset) outputs.
// ------------------------------------------------------------------------------------------
struct DcnF32Params {
  const float *input, *weight, *bias, *offset, *mask;
  float* output;
  int B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, G, Ho, Wo;
};

__device__ __forceinline__ float bilinear_f32(const float* im, int H, int W, float h, float w) {
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h - h_low, lw = w - w_low, hh = 1.f - lh, hw = 1.f - lw;
  float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
  if (h_low >= 0 && w_low >= 0) v1 = im[h_low * W + w_low];
  if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * W + w_high];
  if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * W + w_low];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * W + w_high];
  const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

constexpr int F32_COUT_PER_THREAD = 16;   // 4 waves x 16 = 64 output channels per pass

__global__ __launch_bounds__(256) void dcn_f32_forward_kernel(const DcnF32Params p) {
  __shared__ float col[49][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int npix = p.Ho * p.Wo;
  const int pix = blockIdx.x * 64 + lane;
  const bool valid = pix < npix;
  const int pcl = valid ? pix : npix - 1;
  const int ho = pcl / p.Wo, wo = pcl % p.Wo;
  const int K = p.kh * p.kw;
  const int cpg = p.C / p.G;
  const float* in_b = p.input + (long)b * p.C * p.H * p.W;
  const float* off_b = p.offset + (long)b * p.G * 2 * K * npix;
  const float* msk_b = p.mask + (long)b * p.G * K * npix;

  for (int cob = 0; cob < p.Cout; cob += 4 * F32_COUT_PER_THREAD) {
    float acc[F32_COUT_PER_THREAD];
#pragma unroll
    for (int i = 0; i < F32_COUT_PER_THREAD; ++i) acc[i] = 0.f;
    for (int c = 0; c < p.C; ++c) {
      const int g = c / cpg;
      __syncthreads();
      for (int t = wave; t < K; t += 4) {
        const int i = t / p.kw, j = t % p.kw;
        const float oh = off_b[((long)(g * 2 * K + 2 * t)) * npix + pcl];
        const float ow = off_b[((long)(g * 2 * K + 2 * t + 1)) * npix + pcl];
        const float m = msk_b[((long)(g * K + t)) * npix + pcl];
        const float h_im = (float)(ho * p.sh - p.ph + i * p.dh) + oh;
        const float w_im = (float)(wo * p.sw - p.pw + j * p.dw) + ow;
        float val = 0.f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W)
          val = bilinear_f32(in_b + (long)c * p.H * p.W, p.H, p.W, h_im, w_im);
        col[t][lane] = val * m;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < F32_COUT_PER_THREAD; ++i) {
        const int co = cob + wave * F32_COUT_PER_THREAD + i;
        if (co < p.Cout) {
          const float* wr = p.weight + ((long)co * p.C + c) * K;
          float a = acc[i];
          for (int t = 0; t < K; ++t) a = fmaf(wr[t], col[t][lane], a);
          acc[i] = a;
        }
      }
    }
    if (valid) {
#pragma unroll
      for (int i = 0; i < F32_COUT_PER_THREAD; ++i) {
        const int co = cob + wave * F32_COUT_PER_THREAD + i;
        if (co < p.Cout) p.output[((long)b * p.Cout + co) * npix + pix] = acc[i] + p.bias[co];
      }
    }
  }
}

}  // namespace

extern "C" int tdvc_dcn_fused(const tdvc_dcn_desc* d, void* stream) {
  TDVC_CHECK(d, "tdvc_dcn_fused: null descriptor");
  TDVC_CHECK(fmap_ok16(d->x) && fmap_ok16(d->om) && fmap_ok16(d->y), "tdvc_dcn_fused: fmaps must be fp16, C/sp %% 8, aligned");
  const int G = d->groups;
  TDVC_CHECK(G >= 2 && G <= 8 && (G % 2) == 0, "tdvc_dcn_fused: groups=%d unsupported (even, 2..8)", G);
  TDVC_CHECK(d->x.C == 8 * G && d->y.C == 8 * G, "tdvc_dcn_fused: needs Cin = Cout = 8*groups (8 channels per group)");
  TDVC_CHECK(d->om.C >= 27 * G, "tdvc_dcn_fused: offset/mask fmap needs >= 27*groups channels");
  TDVC_CHECK(d->x.N == d->y.N && d->x.N == d->om.N && d->x.H == d->y.H && d->x.W == d->y.W && d->om.H == d->x.H && d->om.W == d->x.W,
             "tdvc_dcn_fused: geometry mismatch");
  TDVC_CHECK(d->w && aligned16(d->w) && d->bias && aligned16(d->bias), "tdvc_dcn_fused: weights/bias null or unaligned");
  DcnParams p;
  p.x = reinterpret_cast<const half_t*>(d->x.p); p.x_sn = d->x.sn; p.x_sp = d->x.sp; p.H = d->x.H; p.W = d->x.W;
  p.om = reinterpret_cast<const half_t*>(d->om.p); p.om_sn = d->om.sn; p.om_sp = d->om.sp;
  p.y = to_dev(d->y);
  p.w = reinterpret_cast<const half_t*>(d->w); p.bias = d->bias;
  p.G = G; p.act = d->act; p.slope = d->slope; p.round16 = d->round_before_act;
  p.npix = (long)d->x.H * d->x.W;
  dim3 grid((unsigned)((p.npix + 127) / 128), d->x.N);
  hipLaunchKernelGGL(dcn_fused_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  return tdvc_launch_status("tdvc_dcn_fused");
}

extern "C" int tdvc_dcn_v2_forward_f32(const float* input, const float* weight, const float* bias,
                                       const float* offset, const float* mask, float* output,
                                       int B, int C, int H, int W, int Cout,
                                       int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                       int deformable_group, void* stream) {
  TDVC_CHECK(input && weight && bias && offset && mask && output, "dcn_v2_forward: null tensor");
  TDVC_CHECK(B > 0 && C > 0 && H > 0 && W > 0 && Cout > 0, "dcn_v2_forward: empty tensor");
  TDVC_CHECK(kh >= 1 && kw >= 1 && kh * kw <= 49, "dcn_v2_forward: kernel %dx%d unsupported (<= 49 taps)", kh, kw);
  TDVC_CHECK(sh >= 1 && sw >= 1 && dh >= 1 && dw >= 1 && ph >= 0 && pw >= 0, "dcn_v2_forward: bad stride/dilation/pad");
  TDVC_CHECK(deformable_group >= 1 && C % deformable_group == 0, "dcn_v2_forward: channels %d not divisible by deformable_group %d", C, deformable_group);
  DcnF32Params p;
  p.input = input; p.weight = weight; p.bias = bias; p.offset = offset; p.mask = mask; p.output = output;
  p.B = B; p.C = C; p.H = H; p.W = W; p.Cout = Cout; p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw;
  p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw; p.G = deformable_group;
  p.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  p.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  TDVC_CHECK(p.Ho > 0 && p.Wo > 0, "dcn_v2_forward: empty output");
  dim3 grid((p.Ho * p.Wo + 63) / 64, B);
  hipLaunchKernelGGL(dcn_f32_forward_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  return tdvc_launch_status("tdvc_dcn_v2_forward_f32");
}
