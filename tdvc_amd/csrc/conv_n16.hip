// conv_n16 — stride-1 convs with FEW output channels (Cout <= 32) on v_mfma_f32_16x16x32_f16: the layers whose 32-row MFMA
// tiles are mostly padding on the direct kernel (conv_mfma<CK8, 1, 1>):
//     FeatureFix.featdown 3x3 64 -> 3 (planar fp32 out, clamp)            main/model/pnet.py:262
//     SPyNet basic module 7x7 32 -> 16, 16 -> 2 (fp32 flow out), 8 -> 32  main/model/flownet.py:187-227
// On a 32x32x16 tile a 3-channel layer wastes 29 of 32 rows (206 us for a 267 MB read at 1080p), the 2-channel flow head 30 of
// 32 (182 us for 67 MB); the direct kernel also re-fetches every weight fragment from L2 per k-step and stages one 8x32 tile
// per workgroup.  Here:
//   * 16 output channels per MFMA tile (two tiles for 17..32 channels): A = weights [16 couts][32 k], B = activations
//     [32 k][16 pixels]; a lane's accumulator is 4 consecutive output channels of one pixel, the shape convk::epilogue4 takes, so
//     every output mode of the direct kernel (fp16 / fp32 NHWC, planar fp32, bias, activation, residuals) is kept;
//   * k runs over the flattened (tap, channel) order in steps of 32: one tap of 32 channels, half a tap of 64, two taps of 16,
//     four taps of 8 -- the k-group of a lane picks its own tap;
//   * ALL weights of the layer sit in LDS ([k-step][cout block][lane][8], <= 50 KB), gathered once per workgroup straight from
//     the standard packed blob (a 16x16x32 A fragment is four 16-byte pieces of 32x32x16 fragments there: no second packing);
//   * persistent workgroups of 8 waves walk 8 x 64-pixel tiles, a wave owns one row: per k-step 1 A read + 4 B reads feed 4
//     MFMAs (x 2 cout blocks); the fragments of k-step S + 1 are read under the MFMAs of step S (the first version read, waited
//     and multiplied step by step with 4 waves per CU: latency-bound, SLOWER than the padded direct kernel);
//   * dense kh x kw windows only: a lane derives its tap's tile offset arithmetically (no table look-up in the loop).
#include "conv_common.h"

using convk::ConvParams;

namespace {

constexpr int N16_TH = 8, N16_TW = 64, N16_NTHR = 512;     // 8 waves, one output row of 64 pixels each

struct N16Extra {
  int ksteps;          // 32-wide k-steps = ceil(ntaps * cin / 32)
  int ps;              // LDS bytes per staged pixel (2 * cin + 16)
  int a_bytes;         // weight region
  int ntiles, tiles_x;
  int cin_shift;       // log2(cin)
};

template <int NCB>     // cout blocks of 16
__global__ __launch_bounds__(N16_NTHR, 1) void conv_n16_kernel(const ConvParams p, const N16Extra e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;                                          // [kstep][cb][lane 64][16 B]
  unsigned char* tile = wl + e.a_bytes;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, kg = lane >> 4;
  const int TIW = N16_TW + p.kw - 1, TIH = N16_TH + p.kh - 1;
  const int cin = p.Cin, c8n = cin >> 3;

  // A fragments: 16x16x32 lane (row = l & 15, k = 8 (l >> 4) + j) of k-step S = old 32x32x16 lane (16 cb + row) + 32 h of step s,
  // 2 s + h = 4 S + (l >> 4)
  for (int i = tid; i < e.ksteps * NCB * 64; i += N16_NTHR) {
    const int ln = i & 63, cb = (i >> 6) % NCB, S = (i >> 6) / NCB;
    const int q = 4 * S + (ln >> 4), s = q >> 1, h = q & 1;
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (s < p.steps) v = *reinterpret_cast<const half8*>(p.w + ((long)s * 64 + (16 * cb + (ln & 15)) + 32 * h) * 8);
    *reinterpret_cast<half8*>(wl + (long)i * 16) = v;
  }

  const int base = (wave * TIW + px) * e.ps;                         // this wave's row, this lane's pixel of column block 0
  const int cstep = 16 * e.ps;
  // the tile offset of the lane's tap in k-step S (dense window: tap t = (t / kw, t % kw); past the last tap the weights are
  // zero and any in-tile address will do)
  auto frag_off = [&](int S) {
    const int kk0 = 32 * S + 8 * kg;
    const int tap = min(kk0 >> e.cin_shift, p.ntaps - 1);
    const int dy = tap / p.kw, dx = tap - dy * p.kw;
    return base + (dy * TIW + dx) * e.ps + (kk0 & (cin - 1)) * 2;
  };

  for (int tile_i = blockIdx.x; tile_i < e.ntiles; tile_i += gridDim.x) {
    const int per_img = e.tiles_x * ((p.Ho + N16_TH - 1) / N16_TH);
    const int n = tile_i / per_img, rem = tile_i - n * per_img;
    const int ty = rem / e.tiles_x, tx = rem - ty * e.tiles_x;
    const half_t* xn = p.x + (long)n * p.x_sn;
    const int iy0 = ty * N16_TH - p.pad, ix0 = tx * N16_TW - p.pad;
    __syncthreads();                                                 // everyone is done with the previous tile
    // ---- stage the halo tile: eight 16-byte loads of a thread in flight at once
    const int total = TIH * TIW * c8n;
    for (int idx0 = tid; idx0 < total; idx0 += 8 * N16_NTHR) {
      half8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = idx0 + u * N16_NTHR;
        const int pix = idx >> (e.cin_shift - 3), c8 = idx & (c8n - 1);
        const int rr = pix / TIW, cc = pix - rr * TIW;
        const int iy = iy0 + rr, ix = ix0 + cc;
        const bool ok = (int)(idx < total) & (int)(iy >= 0) & (int)(iy < p.H) & (int)(ix >= 0) & (int)(ix < p.W);
        const half_t* src = xn + ((long)(ok ? iy : 0) * p.W + (ok ? ix : 0)) * p.x_sp + c8 * 8;
        // unconditional load at a clamped address, zeroed by an OPAQUE bit mask: given `ok ? load : 0` hipcc sinks the load into a
        // branch and waits for it there (conv_f32.hip has the measurement)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 bits = *reinterpret_cast<const u32x4*>(src);
        unsigned m = ok ? 0xFFFFFFFFu : 0u;
        asm volatile("" : "+v"(m));
        bits &= m;
        v[u] = __builtin_bit_cast(half8, bits);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = idx0 + u * N16_NTHR;
        if (idx < total) {
          const int pix = idx >> (e.cin_shift - 3), c8 = idx & (c8n - 1);
          *reinterpret_cast<half8*>(tile + pix * e.ps + c8 * 16) = v[u];
        }
      }
    }
    __syncthreads();

    f32x4 acc[NCB][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[cb][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    half8 a0[NCB], b0[4], a1[NCB], b1[4];
    auto load = [&](int S, half8 (&a)[NCB], half8 (&b)[4]) {
      const int off = frag_off(S);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) a[cb] = *reinterpret_cast<const half8*>(wl + ((S * NCB + cb) * 64 + lane) * 16);
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const half8*>(tile + off + c * cstep);
    };
    auto mm = [&](half8 (&a)[NCB], half8 (&b)[4]) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cb], b[c], acc[cb][c], 0, 0, 0);
    };
    const int KS = e.ksteps;
    load(0, a0, b0);
    for (int S = 0; S < KS; S += 2) {                                // fragments of the next k-step are read under this step's MFMAs
      load(min(S + 1, KS - 1), a1, b1);
      mm(a0, b0);
      load(min(S + 2, KS - 1), a0, b0);
      if (S + 1 < KS) mm(a1, b1);
    }

    // ---- epilogue: lane = (pixel px, output channels 16 cb + 4 kg .. + 3)
    const int oy = ty * N16_TH + wave;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ox = tx * N16_TW + 16 * c + px;
      if (oy < p.Ho && ox < p.Wo) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          float v[4] = {acc[cb][c][0], acc[cb][c][1], acc[cb][c][2], acc[cb][c][3]};
          convk::epilogue4(p, n, oy, ox, 16 * cb + 4 * kg, v);
        }
      }
    }
  }
}

inline int n16_ps(int cin) { return 2 * cin + 16; }
inline int n16_ksteps(int ntaps, int cin) { return (ntaps * cin + 31) / 32; }
inline int n16_lds(int cin, int kh, int kw, int ntaps, int ncb) {
  return n16_ksteps(ntaps, cin) * ncb * 1024 + (N16_TH + kh - 1) * (N16_TW + kw - 1) * n16_ps(cin);
}

}  // namespace

static bool g_n16_enabled = true;
// tests and A/B benchmarks switch the kernel off to send the same layers to the direct kernel
extern "C" void tdvc_debug_enable_conv_n16(int enable) { g_n16_enabled = enable != 0; }

bool conv_n16_eligible(const tdvc_conv_desc* d, int Ho, int Wo) {
  static const bool off = getenv("TDVC_CONV_NO_N16") != nullptr || getenv("TDVC_CONV_V1") != nullptr;
  if (off || !g_n16_enabled) return false;
  const int cin = d->x.C;
  const bool cin_ok = cin == 8 || cin == 16 || cin == 32 || cin == 64;
  // a single channel chunk (ck == cin): the packed blob's k order is then the flattened (tap, channel) order this kernel walks
  bool dense = d->ntaps == d->kh * d->kw;                        // the kernel derives (dy, dx) from the tap index
  for (int t = 0; dense && t < d->ntaps; ++t) dense = d->tap_dy[t] == t / d->kw && d->tap_dx[t] == t % d->kw;
  return cin_ok && dense && d->ck == cin && d->stride == 1 && d->cout <= 32 && !d->s2d && !d->square_input && !d->gdn && (long)Ho * Wo >= 8192 &&
         n16_lds(cin, d->kh, d->kw, d->ntaps, d->cout <= 16 ? 1 : 2) <= 150 * 1024;
}

int launch_conv_n16(const ConvParams& p, int N, hipStream_t st) {
  const int ncb = p.cout <= 16 ? 1 : 2;
  N16Extra e;
  e.ksteps = n16_ksteps(p.ntaps, p.Cin);
  e.ps = n16_ps(p.Cin);
  e.a_bytes = e.ksteps * ncb * 1024;
  e.tiles_x = (p.Wo + N16_TW - 1) / N16_TW;
  e.ntiles = N * e.tiles_x * ((p.Ho + N16_TH - 1) / N16_TH);
  e.cin_shift = p.Cin == 8 ? 3 : p.Cin == 16 ? 4 : p.Cin == 32 ? 5 : 6;
  const int lds = n16_lds(p.Cin, p.kh, p.kw, p.ntaps, ncb);
  const int per_cu = lds <= 78 * 1024 ? 2 : 1;     // 16 or 8 waves per CU
  int grid = e.ntiles < 256 * per_cu ? e.ntiles : 256 * per_cu;
  auto go = [&](auto kern) -> int {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) { tdvc_set_error("conv n16: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(N16_NTHR), lds, st, p, e);
    return 0;
  };
  const int rc = ncb == 1 ? go(&conv_n16_kernel<1>) : go(&conv_n16_kernel<2>);
  if (rc) return rc;
  return tdvc_launch_status("tdvc_conv2d(n16)");
}
