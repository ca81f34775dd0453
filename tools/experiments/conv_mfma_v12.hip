// conv_mfma_v12 — 3x3, stride 1, pad 1, Cin = 128, Cout = 128, lean epilogue (fp16 NHWC out, bias, none / ReLU / LeakyReLU,
// up to two fp16 residuals), maps of >= 256 k pixels: the 128 -> 128 convs of the coders' ResidualBlocks at half resolution
// (12 launches, 1.9 ms per 1080p frame on conv_mfma_v11).
//
// v11 streams, per 32-channel stage, 36 KB of weights next to 39 KB of tile through LDS-DMA (the 144 KB of a 64-cout weight
// set cannot stay in LDS beside the tiles): 75 DMA pieces per stage at ~100 cycles of issue each, MFMA busy 49 %.  This
// kernel takes the structure of conv_pair instead -- WEIGHTS IN REGISTERS, row streaming -- for one conv of 128 channels:
//   * 4 waves, one per SIMD (up to 512 VGPRs).  Wave w owns output channels [32 w, 32 w + 32) as two blocks of 16 and keeps
//     their weights as v_mfma_f32_16x16x32_f16 A fragments: 2 blocks x 9 taps x 4 chunks of 32 input channels = 72
//     fragments = 288 VGPRs, gathered ONCE per launch straight from the standard packed blob (a fragment of this shape is
//     one 16-byte piece of a 32x32x16 fragment there: no second packing).  LDS holds activations only.
//   * a workgroup walks a strip of 30 output columns top to bottom.  Per step one input row (32 px x 256 B, 8 LDS-DMA pieces,
//     two per wave, into an 8-row ring 6 rows ahead); its B fragments (16 px x 32 channels) come in four chunk groups of 6,
//     double-buffered (the next group is read under the current group's 36 MFMAs, the first group of the NEXT row before
//     the barrier); each fragment feeds 2 cout blocks x 3 live output rows (dy = 0, 1, 2) = 6 MFMAs: 0.17 LDS reads per
//     MFMA (v11: 0.85 per 32x32x16).  144 MFMAs (2304 cycles) per step and wave against ~600 cycles of everything else.
//   * the finished output row goes through an LDS staging row (packed fp16 activation) and leaves as full 256-byte pixels,
//     residuals added there; one barrier per two row steps.
// LDS image of a ring row: 256 B per pixel, the 16-byte chunk c of pixel q at slot c ^ F[q & 15] with the searched table
// F = {0,1,2,3,4,6,8,9,10,11,12,13,4,6,14,15}: the ds_read_b128 of a B fragment (lane = pixel q0 + (l & 15), chunk block
// l >> 4; the instruction's four 16-lane groups give the middle eight pixels the neighbouring chunk) is conflict-free for
// the three window positions dx = 0, 1, 2 (conv_pair.hip has the argument; no bijective table does it).
#include <type_traits>

#include "conv_common.h"

using convk::ConvParams;

namespace {

constexpr int PW12 = 30;                     // output columns per strip
constexpr int ROWB12 = 32 * 256;             // ring row: 32 pixels x 256 B = 8 DMA pieces
constexpr int XR12 = 8, PF12 = 6, BI12 = 2;  // ring rows, DMA distance, row steps per barrier
constexpr int X012 = 0, S012 = XR12 * ROWB12, SROW12 = 32 * 256;
constexpr int LDS12 = S012 + 2 * BI12 * SROW12 + 1024;     // 98 KB + slack for the fragment reads past the last ring row
constexpr int NTHR12 = 256;

struct V12Extra {
  const half_t* zeros;
  half_t* dump;
  int strips, segs, seg_rows, jobs;
};

__device__ __forceinline__ int swz12(int q) { return (int)((0xFE64DCBA98643210ull >> (4 * (q & 15))) & 15ull); }

__device__ __forceinline__ void glds16_12(const half_t* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ half8 lds_read128_pinned12(unsigned addr) {
  half8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void barrier12(bool skip) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (!skip) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int NRES>
__global__ __launch_bounds__(NTHR12, 1) void conv_mfma_v12_kernel(const ConvParams p, const V12Extra e) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kb = lane >> 4;

  // ---- 72 A fragments: cout = 32 wave + 16 cob + (l & 15), channels 32 kc + 8 (l >> 4) .. + 7 of tap t.  In the packed blob
  // (tdvc_pack_conv_weights, ck = 32: [cout tile 32][chunk][step][lane 64][8]) that is lane' = (cout & 31) + 32 h of step s
  // with 2 s + h = 4 t + (l >> 4)
  half8 wf[2][9][4];
#pragma unroll
  for (int cob = 0; cob < 2; ++cob)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const int kk = 4 * t + kb, s = kk >> 1, h = kk & 1;
        const long off = ((((long)(wave * 4 + kc) * 18 + s) * 64) + (16 * cob + r16 + 32 * h)) * 8;
        wf[cob][t][kc] = *reinterpret_cast<const half8*>(p.w + off);
      }
  f32x4 bias4[2];
#pragma unroll
  for (int cob = 0; cob < 2; ++cob) bias4[cob] = *reinterpret_cast<const f32x4*>(p.bias + 32 * wave + 16 * cob + 4 * kb);

  // ---- per-lane LDS offsets inside a ring row: fragment (dx, kc) of column block 0; block 1 is 16 pixels = 4096 B further
  int foff[3][4];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int q = dx + r16, sw = swz12(q);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) foff[dx][kc] = q * 256 + (((4 * kc + kb) ^ sw) << 4);
  }
  int doff[2][2];                              // C/D layout (pixel l & 15, channels 32 w + 16 cob + 4 kb ..+3) in a staging row
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int cob = 0; cob < 2; ++cob) {
      const int q = 16 * cb + r16, c = 4 * wave + 2 * cob + (kb >> 1);
      doff[cb][cob] = q * 256 + ((c ^ swz12(q)) << 4) + 8 * (kb & 1);
    }
  // DMA: piece j of a row = pixels 4 j .. 4 j + 3; this wave sends pieces wave and wave + 4; lane = (pixel lane >> 4, slot lane & 15)
  int soff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = 4 * (wave + 4 * u) + (lane >> 4), c = (lane & 15) ^ swz12(q);
    soff[u] = q * p.x_sp + c * 8;
  }
  // store items: j = tid + 256 u: (output column j >> 4, slot j & 15)
  int s_px[2], s_c[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int j = tid + 256 * u;
    s_px[u] = j >> 4;
    s_c[u] = (j & 15) ^ swz12(j >> 4);
  }
  const half_t hs = (half_t)p.slope;
  const half4 sl4 = {hs, hs, hs, hs};

  const int nwg = (int)gridDim.x, b = (int)blockIdx.x;
  int jfirst, jstep, jend;
  if ((nwg & 7) == 0) {
    const int per = (e.jobs + 7) >> 3;
    jfirst = (b & 7) * per + (b >> 3);
    jstep = nwg >> 3;
    jend = min(e.jobs, ((b & 7) + 1) * per);
  } else {
    jfirst = b; jstep = nwg; jend = e.jobs;
  }

  f32x4 acc[2][3][2];                          // [cout block][row slot][column block]
  half8 f0[6], f1[6];                          // B fragments of a chunk group (cb, dx), double-buffered over the groups
  half8 r1v[2] = {}, r2v[2] = {};

  for (int job = jfirst; job < jend; job += jstep) {
    const int n = job / (e.strips * e.segs);
    const int rem = job - n * (e.strips * e.segs);
    const int seg = rem / e.strips, strip = rem - seg * e.strips;
    const int c0 = strip * PW12, ra = seg * e.seg_rows, rb = min(p.H, ra + e.seg_rows);
    const int rows = rb - ra;
    const half_t* xn = p.x + (long)n * p.x_sn;
    half_t* yn = reinterpret_cast<half_t*>(p.y.p) + (long)n * p.y.sn;

    bool col_ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int col = c0 - 1 + 4 * (wave + 4 * u) + (lane >> 4);
      col_ok[u] = col >= 0 && col < p.W;
    }
    bool s_ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) s_ok[u] = s_px[u] < PW12 && c0 + s_px[u] < p.W;
#pragma unroll
    for (int cob = 0; cob < 2; ++cob)
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[cob][s3][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto issue_row = [&](int kk) __attribute__((always_inline)) {     // x row ra - 1 + kk into ring slot kk & 7
      const int row = ra - 1 + kk;
      const bool rowok = row >= 0 && row < p.H;
      const half_t* base = xn + ((long)row * p.W + (c0 - 1)) * p.x_sp;
      const unsigned dst = lds0 + X012 + (kk & (XR12 - 1)) * ROWB12;
#pragma unroll
      for (int u = 0; u < 2; ++u) glds16_12((rowok && col_ok[u]) ? base + soff[u] : e.zeros, dst + (wave + 4 * u) * 1024);
    };
    // the 36 MFMAs of one chunk group: fragments f[(cb, dx)] of chunk kc x 2 cout blocks x 3 live rows
    auto group_mfmas = [&](auto SNc, auto SMc, auto SDc, auto KCc, half8 (&f)[6]) __attribute__((always_inline)) {
      constexpr int SN = decltype(SNc)::value, SM = decltype(SMc)::value, SD = decltype(SDc)::value, kc = decltype(KCc)::value;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int dyo = 0; dyo < 3; ++dyo) {
          const int dy = 2 - dyo;
          const int slot = dy == 2 ? SD : (dy == 1 ? SM : SN);
          const bool first = dy == 0 && dx == 0 && kc == 0;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int cob = 0; cob < 2; ++cob)
              acc[cob][slot][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cob][dy * 3 + dx][kc], f[cb * 3 + dx],
                                                                           first ? bias4[cob] : acc[cob][slot][cb], 0, 0, 0);
        }
    };
    auto load_group = [&](half8 (&f)[6], unsigned rowbase, int kc) __attribute__((always_inline)) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) f[cb * 3 + dx] = *reinterpret_cast<const half8*>(smem + rowbase + foff[dx][kc] + cb * 4096);
    };

    // ---- prologue
#pragma unroll
    for (int kk = 0; kk < PF12; ++kk)
      if (kk <= rows + 1) issue_row(kk);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier12(false);
    load_group(f0, X012, 0);

    const int K = (rows + 4 + BI12 - 1) & ~(BI12 - 1);
    unsigned hist = 0;
    auto land_wait = [&](int nvm) __attribute__((always_inline)) {
      hist = (hist << 8) | (unsigned)nvm;
      const unsigned sum = (hist & 0xFFu) + ((hist >> 8) & 0xFFu) + ((hist >> 16) & 0xFFu) + (hist >> 24);
      if (sum == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (sum == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (sum == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else if (sum == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // One row step: x row i = ra - 1 + k feeds y rows i + 1 (started), i, i - 1 (finished -> staging row k & 3); the y row
    // finished BI steps ago goes to memory.
    auto step = [&](auto PHc, int k) __attribute__((always_inline)) {
      constexpr int PH = decltype(PHc)::value;
      using S0c = std::integral_constant<int, PH>;
      using S1c = std::integral_constant<int, (PH + 1) % 3>;
      using S2c = std::integral_constant<int, (PH + 2) % 3>;
      int nvm = 0;
      // (a) store the row finished BI steps ago (lanes outside the strip / rows outside the segment write to a dump line)
      {
        const int srow = ra + k - 2 - BI12;
        const bool rowin = srow >= ra && srow < rb;
        const unsigned char* sb = smem + S012 + ((k - BI12) & (2 * BI12 - 1)) * SROW12;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          half8 v = *reinterpret_cast<const half8*>(sb + (tid + 256 * u) * 16);
          if constexpr (NRES >= 1) v = v + r1v[u];
          if constexpr (NRES >= 2) v = v + r2v[u];
          const bool ok = rowin && s_ok[u];
          half_t* dst = ok ? yn + ((long)srow * p.W + c0 + s_px[u]) * p.y.sp + s_c[u] * 8 : e.dump + (tid + 256 * u) * 8;
          *reinterpret_cast<half8*>(dst) = v;
        }
        nvm += 2;
        if constexpr (NRES >= 1) {
          const int nrow = srow + 1;
          const bool nin = nrow >= ra && nrow < rb;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const bool ok = nin && s_ok[u];
            const long po = ((long)nrow * p.W + c0 + s_px[u]);
            r1v[u] = *reinterpret_cast<const half8*>(ok ? reinterpret_cast<const half_t*>(p.res.p) + (long)n * p.res.sn + po * p.res.sp + s_c[u] * 8 : e.zeros);
            if constexpr (NRES >= 2)
              r2v[u] = *reinterpret_cast<const half8*>(ok ? reinterpret_cast<const half_t*>(p.res2.p) + (long)n * p.res2.sn + po * p.res2.sp + s_c[u] * 8 : e.zeros);
          }
          nvm += 2 * NRES;
        }
      }
      // (b) DMA of the row PF steps ahead
      if (k + PF12 <= rows + 1) { issue_row(k + PF12); nvm += 2; }
      // (c) the row's 144 MFMAs in four chunk groups; group g + 1 is read under group g
      const unsigned rb0 = X012 + (k & (XR12 - 1)) * ROWB12;
      load_group(f1, rb0, 1);
      group_mfmas(S1c{}, S0c{}, S2c{}, std::integral_constant<int, 0>{}, f0);
      load_group(f0, rb0, 2);
      group_mfmas(S1c{}, S0c{}, S2c{}, std::integral_constant<int, 1>{}, f1);
      load_group(f1, rb0, 3);
      group_mfmas(S1c{}, S0c{}, S2c{}, std::integral_constant<int, 2>{}, f0);
      {                                        // the first group of the NEXT row, pinned in front of the barrier
        __builtin_amdgcn_sched_barrier(0);
        const unsigned nb = lds0 + X012 + ((k + 1) & (XR12 - 1)) * ROWB12;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) f0[cb * 3 + dx] = lds_read128_pinned12(nb + foff[dx][0] + cb * 4096);
      }
      group_mfmas(S1c{}, S0c{}, S2c{}, std::integral_constant<int, 3>{}, f1);
      // (d) y row i - 1 -> fp16, activation, staging row
      {
        unsigned char* sb = smem + S012 + (k & (2 * BI12 - 1)) * SROW12;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int cob = 0; cob < 2; ++cob) {
            const f32x4 v = acc[cob][(PH + 2) % 3][cb];
            half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            h = __builtin_elementwise_max(h, h * sl4);
            *reinterpret_cast<half4*>(sb + doff[cb][cob]) = h;
          }
      }
      land_wait(nvm);
      barrier12((k & (BI12 - 1)) != BI12 - 1);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    for (int k = 0; k < K; k += 3) {           // surplus steps (K rounded up, then to a multiple of 3) work on rows nobody stores
      step(I0{}, k);
      step(I1{}, k + 1);
      step(I2{}, k + 2);
    }
    // K rounded to a multiple of 3 may leave the barrier phase odd: close the interval
    if ((((K + 2) / 3) * 3) & (BI12 - 1)) barrier12(false);
  }
}

}  // namespace

static bool g_v12_enabled = true;
// tests and A/B benchmarks switch the kernel off to send the same layers to conv_mfma_v11
extern "C" void tdvc_debug_enable_conv_v12(int enable) { g_v12_enabled = enable != 0; }

bool conv_v12_eligible(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo) {
  static const bool off = getenv("TDVC_CONV_NO_V12") != nullptr || getenv("TDVC_CONV_V1") != nullptr;
  if (off || !g_v12_enabled) return false;
  bool taps33 = d->ntaps == 9 && d->kh == 3 && d->kw == 3 && d->pad == 1;
  for (int t = 0; taps33 && t < 9; ++t) taps33 = d->tap_dy[t] == t / 3 && d->tap_dx[t] == t % 3;
  return taps33 && d->ck == 32 && d->stride == 1 && d->cout == 128 && d->x.C == 128 && !d->s2d && !d->square_input &&
         (long)Ho * Wo >= 262144 && Ho >= 32 && convk::conv_is_lean(p) && p.y.C == 128 &&
         (!p.res.p || p.res.C >= 128) && (!p.res2.p || p.res2.C >= 128);
}

int launch_conv_v12(const ConvParams& p, int N, hipStream_t st) {
  static half_t* zeros = nullptr;        // [0, 256): zeros; [256, 256 + 8192): dump lines
  if (!zeros) {
    hipError_t err = hipMalloc(reinterpret_cast<void**>(&zeros), 256 + 8192);
    if (err == hipSuccess) err = hipMemset(zeros, 0, 256 + 8192);
    if (err != hipSuccess) { zeros = nullptr; tdvc_set_error("conv v12: zero page allocation failed: %s", hipGetErrorString(err)); return (int)err; }
  }
  ConvParams q = p;
  q.slope = convk::conv_simple_slope(p);
  V12Extra e;
  e.zeros = zeros;
  e.dump = zeros + 128;
  e.strips = (p.W + PW12 - 1) / PW12;
  const long base = (long)N * e.strips;
  int best = 1;
  double best_eff = 0.;
  for (int sg = 1; sg <= 64 && (p.H + sg - 1) / sg >= 32; ++sg) {
    const int sr = (p.H + sg - 1) / sg, nseg = (p.H + sr - 1) / sr;
    const long jobs = base * nseg;
    const double eff = (double)jobs / (double)(((jobs + 255) / 256) * 256) * sr / (sr + 6.0);
    if (eff > best_eff + 1e-9) { best_eff = eff; best = sg; }
  }
  e.seg_rows = (p.H + best - 1) / best;
  e.segs = (p.H + e.seg_rows - 1) / e.seg_rows;
  e.jobs = (int)(base * e.segs);
  const int grid = e.jobs < 256 ? e.jobs : 256;
  const int nres = (p.res.p ? 1 : 0) + (p.res2.p ? 1 : 0);
  if (nres == 1 && !p.res.p) { q.res = q.res2; q.res2 = null_fmap(); }
  auto go = [&](auto kern) -> int {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS12);
    if (err != hipSuccess) { tdvc_set_error("conv v12: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR12), LDS12, st, q, e);
    return 0;
  };
  const int rc = nres == 0 ? go(&conv_mfma_v12_kernel<0>) : nres == 1 ? go(&conv_mfma_v12_kernel<1>) : go(&conv_mfma_v12_kernel<2>);
  if (rc) return rc;
  return tdvc_launch_status("tdvc_conv2d(v12)");
}
