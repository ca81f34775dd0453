// gdn128 — GDN / inverse GDN over 128 channels in ONE pass over the map:
//     y[p][c] = x[p][c] * rsqrt(n[p][c]) (GDN) or x[p][c] * sqrt(n[p][c]) (inverse GDN) [+ res[p][c]],
//     n[p][c] = beta[c] + sum_k gamma[c][k] * x[p][k]^2
// (compressai's GDN inside ResidualBlockWithStride / ResidualBlockUpsample, main/model/encoder_v3.py:17-40 via
// compressai.layers; eight launches per coder and frame).  The layer is HBM-bound: x in, res in, y out = 768 B per pixel.
//
// As a 1x1 conv with a squared-input prologue and a GDN epilogue on conv_mfma_v5 it ran at 1.9 TB/s algorithmic: two
// workgroups (one per 64-cout block) each staged the whole input, the generic transposed epilogue fetched x a second time as
// the multiplicand, and its fp32 rsqrt path shared the issue port with the matrix phase.  Here, per wave and 32-pixel tile:
//   * x arrives ONCE (8 coalesced 1-KB loads) into a wave-private LDS tile and serves as MFMA operand (squared in registers as
//     packed fp16, exactly what the squared-input prologue computed) AND as the multiplicand;
//   * the 128 x 128 gamma matrix is 32 A fragments = 128 VGPRs, loaded once per launch (the standard packed 1x1 blob);
//   * 32 MFMAs (32x32x16) -> norm, + beta, rounded to fp16 into a second wave-private tile (the two-launch path rounded the
//     norm to fp16 as well: the results are bit-identical to it);
//   * transposed domain: every lane takes 8 channels of a pixel from both tiles, x * rsqrt(n) + res in fp32, ONE rounding,
//     full 256-byte pixel rows out.
// No barrier after the prologue; 4 waves per workgroup, two workgroups per CU (2 x 70 KB of LDS, <= 256 VGPRs).
#include "conv_common.h"

using convk::ConvParams;

namespace {

constexpr int GD_PS = 272;                           // LDS bytes per pixel row (256 + 16: conflict-free b128 / b64 accesses)
constexpr int GD_TILE = 32 * GD_PS;                  // one 32-pixel tile
constexpr int GD_NW = 4, GD_NTHR = 256;
constexpr int GD_LDS = 512 + GD_NW * 2 * GD_TILE;    // beta + per wave {x tile, norm tile}

template <bool HAS_RES>
__global__ __launch_bounds__(GD_NTHR, 2) void gdn128_kernel(const ConvParams p, unsigned total_px, unsigned npix) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* beta_s = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, r = lane & 31;
  if (tid < 128) beta_s[tid] = p.bias[tid];
  unsigned char* xt = smem + 512 + wave * (2 * GD_TILE);
  unsigned char* nt = xt + GD_TILE;

  // gamma: packed 1x1 blob [cout tile 4][chunk 4][step 2][lane 64][8 halves] (ck = 32)
  half8 ga[4][8];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) ga[mt][ks] = *reinterpret_cast<const half8*>(p.w + (((long)mt * 8 + ks) * 64 + lane) * 8);
  __syncthreads();                                   // beta visible; the only barrier

  const unsigned ntiles = (total_px + 31u) >> 5;
  const unsigned nwaves = gridDim.x * GD_NW;
  const int lp = lane >> 4, lc = lane & 15;          // transposed domain: pixel 4j + lp, 16-byte chunk lc (channels 8 lc ..)
  const bool inverse = p.gdn == TDVC_GDN_INV;
  const half_t* resp = reinterpret_cast<const half_t*>(p.res.p);
  half_t* yp = reinterpret_cast<half_t*>(p.y.p);

  for (unsigned tile = blockIdx.x * GD_NW + wave; tile < ntiles; tile += nwaves) {
    const unsigned px0 = tile << 5;
    // flattened pixel index over (n, y, x): a tile may straddle rows and images, so every pixel row is addressed on its own:
    // image n = q / npix, pixel rem = q - n * npix (x, res and y share the geometry, not the strides)
    unsigned qn[8], qr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned q = px0 + 4 * j + lp;
      q = q < total_px ? q : total_px - 1;
      qn[j] = q / npix;
      qr[j] = q - qn[j] * npix;
    }
    // ---- x tile in: 8 x (4 pixels x 256 B), straight through registers into the wave-private tile
    {
      half8 xin[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xin[j] = *reinterpret_cast<const half8*>(p.x + (long)qn[j] * p.x_sn + (long)qr[j] * p.x_sp + lc * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<half8*>(xt + (4 * j + lp) * GD_PS + lc * 16) = xin[j];
    }
    // ---- norm = gamma . x^2 (+ beta below): B fragment = 8 channels of pixel r, squared as packed fp16
    f32x16 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      half8 b = *reinterpret_cast<const half8*>(xt + r * GD_PS + (ks * 16 + hh * 8) * 2);
      b = b * b;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[mt][ks], b, acc[mt], 0, 0, 0);
    }
    // the residual rows: issued here so that their latency runs under the norm write-out and the LDS round trip
    half8 rin[8];
    if constexpr (HAS_RES) {
#pragma unroll
      for (int j = 0; j < 8; ++j) rin[j] = *reinterpret_cast<const half8*>(resp + (long)qn[j] * p.res.sn + (long)qr[j] * p.res.sp + lc * 8);
    }
    // ---- norm + beta -> fp16 -> norm tile (C/D layout: channels mt*32 + 8g + 4hh + i of pixel r)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = mt * 32 + 8 * g + 4 * hh;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta_s + c0);
        half4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (half_t)(acc[mt][4 * g + i] + b4[i]);
        *reinterpret_cast<half4*>(nt + r * GD_PS + c0 * 2) = o;
      }
    // ---- transposed domain: x * (r)sqrt(norm) + res in fp32, one rounding, full rows out
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const half8 xv = *reinterpret_cast<const half8*>(xt + (4 * j + lp) * GD_PS + lc * 16);
      const half8 nv = *reinterpret_cast<const half8*>(nt + (4 * j + lp) * GD_PS + lc * 16);
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (float)xv[k] * (inverse ? sqrtf((float)nv[k]) : rsqrtf((float)nv[k]));
      if constexpr (HAS_RES) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += (float)rin[j][k];
      }
      half8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (half_t)v[k];
      if (px0 + 4 * j + lp < total_px) *reinterpret_cast<half8*>(yp + (long)qn[j] * p.y.sn + (long)qr[j] * p.y.sp + lc * 8) = o;
    }
  }
}

}  // namespace

static bool g_gdn128_enabled = true;
// tests and A/B benchmarks switch the kernel off to send the same layers to conv_mfma_v5 (1x1 conv + GDN epilogue)
extern "C" void tdvc_debug_enable_gdn128(int enable) { g_gdn128_enabled = enable != 0; }

bool gdn128_eligible(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo) {
  static const bool off = getenv("TDVC_CONV_NO_GDN128") != nullptr || getenv("TDVC_CONV_V1") != nullptr;
  if (off || !g_gdn128_enabled) return false;
  // the GDN call of ops.conv: 1x1 / stride 1 / pad 0 over x^2, multiplicand aux == x itself, fp16 NHWC output, optional fp16
  // residual of the output geometry, no activation
  return d->kh == 1 && d->kw == 1 && d->ntaps == 1 && d->stride == 1 && d->pad == 0 && d->ck == 32 && d->square_input && d->gdn != TDVC_GDN_NONE &&
         d->x.C == 128 && d->cout == 128 && !d->s2d && d->out_mode == TDVC_OUT_NHWC && d->y.dtype == TDVC_F16 && d->y.C >= 128 && d->bias &&
         d->act == TDVC_ACT_NONE && !d->round_before_act && !d->res2.p && d->aux.p == d->x.p && d->aux.sp == d->x.sp && d->aux.sn == d->x.sn &&
         d->aux.dtype == TDVC_F16 && (!d->res.p || (d->res.dtype == TDVC_F16 && d->res.C >= 128)) &&
         (long)Ho * Wo >= 8192 && (long)d->x.N * Ho * Wo < (1L << 31) - 64;
}

int launch_gdn128(const ConvParams& p, int N, hipStream_t st) {
  const long total = (long)N * p.Ho * p.Wo;
  long blocks = ((total + 31) / 32 + GD_NW - 1) / GD_NW;
  if (blocks > 512) blocks = 512;                    // two workgroups per CU
  auto go = [&](auto kern) -> int {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GD_LDS);
    if (err != hipSuccess) { tdvc_set_error("gdn128: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(GD_NTHR), GD_LDS, st, p, (unsigned)total, (unsigned)((long)p.Ho * p.Wo));
    return 0;
  };
  const int rc = p.res.p ? go(&gdn128_kernel<true>) : go(&gdn128_kernel<false>);
  if (rc) return rc;
  return tdvc_launch_status("tdvc_conv2d(gdn128)");
}
