// conv_mfma_v9 — small feature maps (<= 8192 output pixels over the batch): the hyperprior / context / entropy-parameter
// convs at H/16 .. H/64, the coders' inner blocks at training crop sizes, and their dgrad forms.
//
// On these maps the tiled kernels are latency chains: a handful of workgroups each walk the whole K = Cin x taps
// contraction stage by stage (stage a tile, barrier, MFMAs, barrier ...): 22 us for a 3x3 128->128 conv at 16x16
// regardless of its 0.3 GFLOP.  v9 has no staging and no barrier in the contraction:
//   * one workgroup = 32 output channels x 32 output pixels (flattened over batch, rows, columns); its 8 waves SPLIT K:
//     wave w contracts k-steps [w T/8, (w+1) T/8) of the layer's packed weight stream;
//   * both MFMA operands come straight from global memory / L2 — the weight fragment is one coalesced 1 KB load (the host
//     packing order), the activation fragment is a 16-byte piece of pixel (oy*s - pad + dy, ox*s - pad + dx) per lane
//     (clamped address, zeroed by a select: no branch, so the loads of several k-steps are in flight together);
//   * the 8 partial accumulators meet in LDS and are summed in a fixed order; the general epilogue (bias, GDN,
//     activation, residuals, any output format) runs on 4 consecutive channels per thread.
#include "conv_common.h"

using convk::ConvParams;

namespace {

// NW9 = waves per workgroup = K slices: 8 for long contractions, fewer when a slice would be under ~4 k-steps
template <int CK8, int NW9>
__global__ __launch_bounds__(NW9 * 64) void conv_mfma_v9_kernel(const ConvParams p, int total_px) {
  __shared__ float red[NW9][32][33];
  __shared__ int tdy[TDVC_MAX_TAPS], tdx[TDVC_MAX_TAPS];
  constexpr int CK = CK8 * 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, r = lane & 31;
  const int ct = blockIdx.y;
  if (tid < p.ntaps) { tdy[tid] = p.tap_dy[tid]; tdx[tid] = p.tap_dx[tid]; }
  __syncthreads();

  const int px = blockIdx.x * 32 + r;
  const bool pv = px < total_px;
  const int pxc = pv ? px : 0;
  const int hw = p.Ho * p.Wo;
  const int n = pxc / hw, rem = pxc - n * hw;
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
  const int iy0 = oy * p.in_stride - p.pad, ix0 = ox * p.in_stride - p.pad;
  const half_t* xn = p.x + (long)n * p.x_sn;

  const int T = p.nchunks * p.steps;
  const int g0 = (int)((long)T * wave / NW9), g1 = (int)((long)T * (wave + 1) / NW9);
  const half_t* wp = p.w + ((long)ct * T) * 512 + lane * 8;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  half8 zero8;
#pragma unroll
  for (int q = 0; q < 8; ++q) zero8[q] = (half_t)0.f;

  // four k-steps per trip: eight independent loads are issued before the first MFMA consumes one (a tail trip repeats
  // the last k-step with a zeroed activation fragment)
  for (int g4 = g0; g4 < g1; g4 += 4) {
    half8 a[4], b[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = min(g4 + u, g1 - 1);
      const int ch = g / p.steps, s = g - ch * p.steps;
      int tap, cofs;
      if constexpr (CK8 == 1) {
        tap = min(2 * s + hh, p.ntaps - 1);               // the padded half step carries zero weights
        cofs = ch * 8;
      } else {
        constexpr int H2 = CK8 / 2;
        tap = s / H2;
        cofs = ch * CK + (s - tap * H2) * 16 + hh * 8;
      }
      const int iy = iy0 + tdy[tap], ix = ix0 + tdx[tap];
      ok[u] = pv && g4 + u < g1 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && cofs < p.Cin;
      const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1), cc = min(cofs, p.Cin - 8);
      b[u] = *reinterpret_cast<const half8*>(xn + ((long)iyc * p.W + ixc) * p.x_sp + cc);
      a[u] = *reinterpret_cast<const half8*>(wp + (long)g * 512);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      half8 bv = ok[u] ? b[u] : zero8;
      if (p.square) bv = bv * bv;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], bv, acc, 0, 0, 0);
    }
  }

#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][(i & 3) + 8 * (i >> 2) + 4 * hh][r] = acc[i];
  __syncthreads();
  for (int it = tid; it < 256; it += NW9 * 64) {
    const int pl = it & 31, grp = it >> 5;                 // pixel, group of 4 consecutive output channels
    const int q = blockIdx.x * 32 + pl;
    if (q < total_px) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW9; ++w) s += red[w][4 * grp + i][pl];
        v[i] = s;
      }
      const int qn = q / hw, qr = q - qn * hw;
      const int qy = qr / p.Wo, qx = qr - qy * p.Wo;
      convk::epilogue4(p, qn, qy, qx, ct * 32 + 4 * grp, v);
    }
  }
}

}  // namespace

static bool g_v9_enabled = true;
static long g_v9_work_limit = 1L << 20;      // pixels x output channels above which a v3-eligible layer stays on v3
extern "C" void tdvc_debug_set_conv_v9_work_limit(long v) { g_v9_work_limit = v; }
// tests switch the small-map kernel off to keep exercising the tiled kernels on small shapes
extern "C" void tdvc_debug_enable_conv_v9(int enable) { g_v9_enabled = enable != 0; }

bool conv_v9_eligible(const tdvc_conv_desc* d, int Ho, int Wo, bool v3_ok) {
  static const bool off = getenv("TDVC_CONV_NO_V9") != nullptr || getenv("TDVC_CONV_V1") != nullptr;
  if (off || !g_v9_enabled || d->s2d) return false;
  const long px = (long)Ho * Wo * d->x.N;
  static const long maxpx = getenv("TDVC_V9_MAX_PX") ? atol(getenv("TDVC_V9_MAX_PX")) : 8192;
  static const long minpx = getenv("TDVC_V9_MIN_PX") ? atol(getenv("TDVC_V9_MIN_PX")) : 0;
  const long ksteps = (long)((d->x.C + d->ck - 1) / d->ck) * ((d->ntaps * (d->ck / 8) + 1) / 2);
  // every 32-channel output tile gathers its own activation fragments: with many output channels on the larger maps the
  // stage-pipelined kernel (one staged tile for 64 output channels) wins again (3x3 128->512 at 68x120: 31 vs 72 us)
  if (v3_ok && px * d->cout > g_v9_work_limit) return false;
  return px <= maxpx && px >= minpx && ksteps >= 4 && (d->x.C % 8) == 0;
}

template <int CK8>
static void launch_v9_nw(const ConvParams& p, dim3 grid, int total, hipStream_t st) {
  const int T = p.nchunks * p.steps;
  if (T >= 32) hipLaunchKernelGGL((conv_mfma_v9_kernel<CK8, 8>), grid, dim3(512), 0, st, p, total);
  else if (T >= 16) hipLaunchKernelGGL((conv_mfma_v9_kernel<CK8, 4>), grid, dim3(256), 0, st, p, total);
  else if (T >= 8) hipLaunchKernelGGL((conv_mfma_v9_kernel<CK8, 2>), grid, dim3(128), 0, st, p, total);
  else hipLaunchKernelGGL((conv_mfma_v9_kernel<CK8, 1>), grid, dim3(64), 0, st, p, total);
}

int launch_conv_v9(const ConvParams& p, int ck8, int cout_tiles32, int N, hipStream_t st) {
  const int total = N * p.Ho * p.Wo;
  const dim3 grid((unsigned)((total + 31) / 32), (unsigned)cout_tiles32);
  switch (ck8) {
    case 1: launch_v9_nw<1>(p, grid, total, st); break;
    case 2: launch_v9_nw<2>(p, grid, total, st); break;
    case 4: launch_v9_nw<4>(p, grid, total, st); break;
    default: launch_v9_nw<8>(p, grid, total, st); break;
  }
  return tdvc_launch_status("tdvc_conv2d");
}
