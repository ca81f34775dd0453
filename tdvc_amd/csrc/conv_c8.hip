// conv_c8 — 3x3 / stride 1 / pad 1, Cin = 8 (an RGB frame padded to one 16-byte pixel), Cout = 64, lean epilogue without
// residuals: the first layers of FeaExtra (x2), LoopFilter (conv01 on the three reference frames) and FeatureExtract_ref
// (main/model/pnet.py:89,268,323) -- four launches per 1080p frame.
//
// The layer is a 33 MB read and a 267 MB write: its floor is the store rate (~60 us at 1080p).  On the generic direct kernel
// (conv_mfma<1,2,1>: one 8x32 tile per workgroup, tile staged through LDS behind two barriers, the ten weight fragments
// fetched from L2 by every wave of every tile, nothing overlapping across tiles) it took 147 us = 1.8 TB/s.  Here:
//   * persistent waves, NO barrier anywhere: a wave owns a sequence of 32-pixel row segments (its "tiles") and walks it alone;
//   * the ten A fragments (64 couts x K = 9 taps x 8 channels, padded to five k-steps of 16) stay in 40 registers for the
//     whole launch;
//   * the B fragment of k-step s is, per lane (pixel r, half h), the 16-byte pixel at tap 2s + h: one global load straight
//     from L1 / L2 (the whole input is 33 MB; every pixel is read nine times within neighbouring lanes and rows), five per
//     tile, issued one tile ahead (double-buffered registers) so that their latency runs under the previous tile's epilogue;
//   * 10 MFMAs per tile, then the shared lean epilogue (conv_common.h): bias, packed-fp16 LeakyReLU, transpose through a
//     wave-private 4.6 KB LDS region, four full-line stores.
// Four waves per SIMD (128 VGPRs): 16 waves per CU each with 5 loads / 4 stores in flight.
#include "conv_common.h"

using convk::ConvParams;

namespace {

constexpr int NW_C8 = 4, NTHR_C8 = 256;
constexpr int EW_C8 = 32 * 144;                      // one transposed output row per wave
constexpr int LDS_C8 = 256 + NW_C8 * EW_C8;          // bias + 4 wave-private regions

__global__ __launch_bounds__(NTHR_C8, 4) void conv_c8_kernel(const ConvParams p, int tiles_x, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* bias_s = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, r = lane & 31;
  if (tid < 64) bias_s[tid] = p.bias[tid];
  unsigned char* ew = smem + 256 + wave * EW_C8;

  // A fragments: packed blob [cout tile 2][chunk 1][step 5][lane 64][8 halves] (tdvc_pack_conv_weights, ck = 8)
  half8 wa[2][5];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int s = 0; s < 5; ++s) wa[mt][s] = *reinterpret_cast<const half8*>(p.w + (((long)mt * 5 + s) * 64 + lane) * 8);

  // this lane's tap of k-step s: t = 2s + hh (the 10th "tap" is the zero-weight padding of K: any in-range pixel will do)
  int tdy[5], tdx[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int t = min(2 * s + hh, 8);
    tdy[s] = t / 3 - 1;
    tdx[s] = t % 3 - 1;
  }
  __syncthreads();                                   // bias visible; the only barrier of the kernel

  const int nwaves = (int)gridDim.x * NW_C8;
  const int w0 = (int)blockIdx.x * NW_C8 + wave;
  const int per_img = tiles_x * p.Ho;
  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto load_b = [&](int tile, half8 (&b)[5]) {
    const int n = tile / per_img, rem = tile - n * per_img;
    const int oy = rem / tiles_x, ox0 = (rem - oy * tiles_x) * 32;
    const half_t* xn = p.x + (long)n * p.x_sn;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int iy = oy + tdy[s], ix = ox0 + r + tdx[s];
      const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const half_t* src = xn + ((long)(ok ? iy : 0) * p.W + (ok ? ix : 0)) * p.x_sp;
      const half8 v = *reinterpret_cast<const half8*>(src);
      b[s] = ok ? v : zero8;
    }
  };

  half8 bcur[5], bnext[5];
  int tile = w0;
  if (tile < total_tiles) load_b(tile, bnext);
  for (; tile < total_tiles; tile += nwaves) {
#pragma unroll
    for (int s = 0; s < 5; ++s) bcur[s] = bnext[s];
    if (tile + nwaves < total_tiles) load_b(tile + nwaves, bnext);
    f32x16 acc[2][1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][0][i] = 0.f;
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[mt][s], bcur[s], acc[mt][0], 0, 0, 0);
    const int n = tile / per_img, rem = tile - n * per_img;
    const int oy = rem / tiles_x, ox0 = (rem - oy * tiles_x) * 32;
    // lean epilogue without residuals (conv_common.h's epilogue_lean_seq keeps two residual rows in registers: 32 VGPRs this
    // kernel does not have at three waves per SIMD): bias, packed fp16 activation, transpose, four full-line stores
    {
      constexpr int EPS = 144;
      const int chunk = lane & 7, prow = lane >> 3;
      const bool ch_ok = chunk * 8 < p.y.C;
      const half_t sl = (half_t)p.slope;
      const half2v sl2 = {sl, sl};
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + mt * 32 + 8 * g + 4 * hh);
          half2v lo = {(half_t)(acc[mt][0][4 * g + 0] + b4[0]), (half_t)(acc[mt][0][4 * g + 1] + b4[1])};
          half2v hi = {(half_t)(acc[mt][0][4 * g + 2] + b4[2]), (half_t)(acc[mt][0][4 * g + 3] + b4[3])};
          lo = __builtin_elementwise_max(lo, lo * sl2);      // slope 1: none, 0: ReLU, else LeakyReLU (conv_simple_slope)
          hi = __builtin_elementwise_max(hi, hi * sl2);
          const half4 o = {lo[0], lo[1], hi[0], hi[1]};
          *reinterpret_cast<half4*>(ew + r * EPS + (mt * 32 + 8 * g + 4 * hh) * 2) = o;
        }
      half_t* yb = reinterpret_cast<half_t*>(p.y.p) + (long)n * p.y.sn + ((long)oy * p.Wo + ox0) * p.y.sp + chunk * 8;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int px = k * 8 + prow;
        const half8 v = *reinterpret_cast<const half8*>(ew + px * EPS + chunk * 16);
        if (ch_ok && ox0 + px < p.Wo) *reinterpret_cast<half8*>(yb + (long)px * p.y.sp) = v;
      }
    }
  }
}

}  // namespace

static bool g_c8_enabled = true;
// tests and A/B benchmarks switch the kernel off to send the same layers to the direct kernel (conv_mfma<1,2,1>)
extern "C" void tdvc_debug_enable_conv_c8(int enable) { g_c8_enabled = enable != 0; }

bool conv_c8_eligible(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo) {
  static const bool off = getenv("TDVC_CONV_NO_C8") != nullptr || getenv("TDVC_CONV_V1") != nullptr;
  if (off || !g_c8_enabled) return false;
  bool taps33 = d->ntaps == 9 && d->kh == 3 && d->kw == 3 && d->pad == 1;
  for (int t = 0; taps33 && t < 9; ++t) taps33 = d->tap_dy[t] == t / 3 && d->tap_dx[t] == t % 3;
  return taps33 && d->ck == 8 && d->stride == 1 && d->cout == 64 && d->x.C == 8 && !d->s2d && !d->square_input && (long)Ho * Wo >= 8192 &&
         convk::conv_is_lean(p) && !p.res.p && !p.res2.p && p.y.C >= 64 && (long)d->x.N * Ho * ((Wo + 31) / 32) < (1L << 30);
}

int launch_conv_c8(const ConvParams& p, int N, hipStream_t st) {
  ConvParams q = p;
  q.slope = convk::conv_simple_slope(p);
  const int tiles_x = (p.Wo + 31) / 32;
  const long total = (long)N * p.Ho * tiles_x;
  int grid = (int)((total + NW_C8 - 1) / NW_C8);
  if (grid > 256 * 4) grid = 256 * 4;                // four workgroups per CU: 16 persistent waves
  hipLaunchKernelGGL(conv_c8_kernel, dim3(grid), dim3(NTHR_C8), LDS_C8, st, q, tiles_x, (int)total);
  return tdvc_launch_status("tdvc_conv2d(c8)");
}
