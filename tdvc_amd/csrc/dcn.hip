// Modulated deformable convolution (DCNv2) for gfx950 — the motion-compensation operator
// (SURVEY §8 a11/a12).
//
// (1) tdvc_dcn_fused: fp16 NHWC, sampling + modulation + 64x576 contraction in ONE kernel.
//     The reference materialises `columns` in HBM (576 values / pixel = 4.8 GB at 1080p,
//     src/cuda/dcn_v2_cuda.cu:67-92); here the modulated samples of an 8x8 pixel tile live in a
//     25 KB LDS tile, three taps at a time, and are consumed as MFMA B fragments.
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
  const half_t* xp;   // group-planar copy of x ([N][G][H][W][8]) or null
};

// x (NHWC, 8G channels) -> [n][g][H][W][8]: 8 lanes read one pixel's 128-byte line, every group plane receives 128
// contiguous bytes per 8 pixels
__global__ void dcn_planarise_kernel(const half_t* x, long x_sn, int x_sp, long npix, int G, half_t* xp) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (i >= npix * G) return;
  const int g = (int)(i % G);
  const long pix = i / G;
  const half8 v = *reinterpret_cast<const half8*>(x + (long)n * x_sn + pix * x_sp + g * 8);
  *reinterpret_cast<half8*>(xp + (((long)n * G + g) * npix + pix) * 8) = v;
}

// Fused kernel, 3x3 / stride 1 / pad 1 / dilation 1, 8 channels per deformable group.
// One 256-thread workgroup = an 8x8 pixel tile x all 8G output channels.
//   sampling phase: thread (pixel = tid>>3 [+32], group = tid&7) -> the 8 lanes of a pixel read
//     one full 128-byte NHWC line per bilinear corner (coalesced), branch-free (clamped address,
//     zeroed weight) so all corner loads of a tap are in flight together; modulated samples go
//     to an LDS column tile of 3 taps x 8G channels per pixel (stride +16 B: conflict-free).
//   MFMA phase: wave (mt, nt) multiplies the 32-row weight tile mt with the 32-pixel half nt,
//     B fragments = 16-byte LDS reads (lane = pixel, k-chunk = group), 3 taps per barrier pair.
constexpr int DCN_TPX = 8, DCN_TPY = 8, DCN_TAPS_PER_CHUNK = 3;

__device__ __forceinline__ half8 sample8_bf(const half_t* xg, int H, int W, int sp, float h_im, float w_im, float mask) {
  // dcn_v2_im2col_cuda.cu:25-54 (bilinear, zero outside) and :180 (open-interval test), branch-free
  const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
  const float hc = fminf(fmaxf(h_im, -2.f), (float)H + 1.f), wc = fminf(fmaxf(w_im, -2.f), (float)W + 1.f);
  const float hf = floorf(hc), wf = floorf(wc);
  const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
  const float lh = hc - hf, lw = wc - wf, hh = 1.f - lh, hw = 1.f - lw;
  const bool hl = h_low >= 0 && h_low <= H - 1, hhg = h_high >= 0 && h_high <= H - 1;
  const bool wl = w_low >= 0 && w_low <= W - 1, whg = w_high >= 0 && w_high <= W - 1;
  // corner weights with the modulation mask folded in; the 8-channel combine is 4 fused multiply-adds per
  // channel straight from the fp16 samples (v_fma_mix_f32), ~40 vector instructions instead of ~100
  const float w1 = (inside && hl && wl) ? hh * hw * mask : 0.f, w2 = (inside && hl && whg) ? hh * lw * mask : 0.f;
  const float w3 = (inside && hhg && wl) ? lh * hw * mask : 0.f, w4 = (inside && hhg && whg) ? lh * lw * mask : 0.f;
  const int y0 = min(max(h_low, 0), H - 1), y1 = min(max(h_high, 0), H - 1);
  const int x0 = min(max(w_low, 0), W - 1), x1 = min(max(w_high, 0), W - 1);
  const int r0 = y0 * W, r1 = y1 * W;                     // pixel indices fit 32 bits (checked by the host)
  const half8 v1 = *reinterpret_cast<const half8*>(xg + (long)(r0 + x0) * sp);
  const half8 v2 = *reinterpret_cast<const half8*>(xg + (long)(r0 + x1) * sp);
  const half8 v3 = *reinterpret_cast<const half8*>(xg + (long)(r1 + x0) * sp);
  const half8 v4 = *reinterpret_cast<const half8*>(xg + (long)(r1 + x1) * sp);
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    o[j] = (half_t)__builtin_fmaf(w1, (float)v1[j], __builtin_fmaf(w2, (float)v2[j], __builtin_fmaf(w3, (float)v3[j], w4 * (float)v4[j])));
  return o;
}

__global__ __launch_bounds__(256) void dcn_fused_kernel(const DcnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char col[];
  const int G = p.G;
  const int PS = DCN_TAPS_PER_CHUNK * G * 16 + 16;       // LDS bytes per pixel (3 taps x 8G halves + pad)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, r = lane & 31;
  const int mt = wave & 1, nt = wave >> 1;
  const int tiles_x = (p.W + DCN_TPX - 1) / DCN_TPX;
  const int tile_id = tdvc_xcd_tile(blockIdx.x);
  const int tx = tile_id % tiles_x, ty = tile_id / tiles_x;
  const int n = blockIdx.y;
  const half_t* xn = p.x + (long)n * p.x_sn;

  // ---- sampling identity: two pixels per thread (pass 0: pixels 0..31, pass 1: 32..63)
  const int g = tid & 7;
  const bool g_on = g < G;
  const int gg = g_on ? g : 0;
  // gather source of this thread's group: the NHWC map (pixel stride x_sp: every 16-byte corner of every lane is a line
  // of its own -- the L1 takes one line per clock, profiles/r01_dcn_fused_1080p_pmc.txt) or the group-planar copy
  // [n][g][H][W][8] (pixel stride 8: the 8 lanes of a group sample 8 neighbouring pixels, whose corners share lines)
  const half_t* xg = p.xp ? p.xp + ((long)n * G + gg) * p.npix * 8 : xn + gg * 8;
  const int xsp = p.xp ? 8 : p.x_sp;
  int soy[2], sox[2];
  half2v off[2][9];
  float msk[2][9];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int pl = (tid >> 3) + 32 * ps;
    soy[ps] = ty * DCN_TPY + (pl >> 3);
    sox[ps] = tx * DCN_TPX + (pl & 7);
    const int cy = min(soy[ps], p.H - 1), cx = min(sox[ps], p.W - 1);
    const half_t* omp = p.om + (long)n * p.om_sn + ((long)cy * p.W + cx) * p.om_sp;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      off[ps][t] = *reinterpret_cast<const half2v*>(omp + gg * 18 + 2 * t);
      msk[ps][t] = __builtin_amdgcn_rcpf(1.f + __expf(-(float)omp[18 * G + gg * 9 + t]));    // sigmoid (dcn_v2_amp.py: mask = sigmoid(mask))
    }
  }

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int steps_per_tap = G / 2;
  const half_t* wbase = p.w + ((long)mt * 9 * steps_per_tap * 64 + lane) * 8;
  const unsigned char* bcol = col + (nt * 32 + r) * PS + hh * 16;

#pragma unroll
  for (int chunk = 0; chunk < 3; ++chunk) {             // fully unrolled: tap registers are selected at compile time
    if (chunk > 0) __syncthreads();
    if (g_on) {
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
        for (int tc = 0; tc < DCN_TAPS_PER_CHUNK; ++tc) {
          const half2v o2 = off[ps][chunk * 3 + tc];
          const float h_im = (float)(soy[ps] - 1 + chunk) + (float)o2[0];      // tap = chunk*3 + tc: dy = chunk, dx = tc
          const float w_im = (float)(sox[ps] - 1 + tc) + (float)o2[1];
          const half8 v = sample8_bf(xg, p.H, p.W, xsp, h_im, w_im, msk[ps][chunk * 3 + tc]);
          const int pl = (tid >> 3) + 32 * ps;
          *reinterpret_cast<half8*>(col + pl * PS + tc * G * 16 + gg * 16) = v;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int tc = 0; tc < DCN_TAPS_PER_CHUNK; ++tc) {
      const int tap = chunk * 3 + tc;
      for (int s2 = 0; s2 < steps_per_tap; ++s2) {
        const half8 a = *reinterpret_cast<const half8*>(wbase + (long)(tap * steps_per_tap + s2) * 512);
        const half8 b = *reinterpret_cast<const half8*>(bcol + tc * G * 16 + s2 * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      }
    }
  }

  const int pl = nt * 32 + r;
  const int oy = ty * DCN_TPY + (pl >> 3), ox = tx * DCN_TPX + (pl & 7);
  if (oy >= p.H || ox >= p.W) return;
  half_t* yp = reinterpret_cast<half_t*>(p.y.p) + (long)n * p.y.sn + ((long)oy * p.W + ox) * p.y.sp;
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int co = mt * 32 + 8 * gq + 4 * hh;
    if (co >= p.y.C) continue;
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + co);
    half4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = acc[4 * gq + i] + b4[i];
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
  TDVC_CHECK(G == 8, "tdvc_dcn_fused: groups=%d unsupported (the fused kernel is built for the hot-path geometry: 8 groups x 8 channels; use tdvc_dcn_v2_forward_f32 otherwise)", G);
  TDVC_CHECK(d->x.C == 8 * G && d->y.C == 8 * G, "tdvc_dcn_fused: needs Cin = Cout = 8*groups (8 channels per group)");
  TDVC_CHECK(d->om.C >= 27 * G, "tdvc_dcn_fused: offset/mask fmap needs >= 27*groups channels");
  TDVC_CHECK(d->x.N == d->y.N && d->x.N == d->om.N && d->x.H == d->y.H && d->x.W == d->y.W && d->om.H == d->x.H && d->om.W == d->x.W,
             "tdvc_dcn_fused: geometry mismatch");
  TDVC_CHECK(d->w && aligned16(d->w) && d->bias && aligned16(d->bias), "tdvc_dcn_fused: weights/bias null or unaligned");
  TDVC_CHECK((long)d->x.H * d->x.W < 2147483647L, "tdvc_dcn_fused: image too large (pixel indices are 32-bit)");
  DcnParams p;
  p.x = reinterpret_cast<const half_t*>(d->x.p); p.x_sn = d->x.sn; p.x_sp = d->x.sp; p.H = d->x.H; p.W = d->x.W;
  p.om = reinterpret_cast<const half_t*>(d->om.p); p.om_sn = d->om.sn; p.om_sp = d->om.sp;
  p.y = to_dev(d->y);
  p.w = reinterpret_cast<const half_t*>(d->w); p.bias = d->bias;
  p.G = G; p.act = d->act; p.slope = d->slope; p.round16 = d->round_before_act;
  p.npix = (long)d->x.H * d->x.W;
  p.xp = reinterpret_cast<const half_t*>(d->x_planar);
  if (p.xp) {
    TDVC_CHECK(aligned16(d->x_planar), "tdvc_dcn_fused: x_planar scratch unaligned");
    const long items = p.npix * G;
    hipLaunchKernelGGL(dcn_planarise_kernel, dim3((unsigned)((items + 255) / 256), d->x.N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       p.x, p.x_sn, p.x_sp, p.npix, G, reinterpret_cast<half_t*>(d->x_planar));
  }
  dim3 grid((unsigned)(((d->x.W + DCN_TPX - 1) / DCN_TPX) * ((d->x.H + DCN_TPY - 1) / DCN_TPY)), d->x.N);
  const size_t lds = (size_t)64 * (DCN_TAPS_PER_CHUNK * G * 16 + 16);
  hipLaunchKernelGGL(dcn_fused_kernel, grid, dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p);
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
