// conv_mfma_v4 — weight-stationary variant of v3 for the layers whose whole weight set fits in LDS
// next to a tile: 3x3 (<= 9 taps), stride 1, Cin <= 64 (<= 2 chunks of 32), Cout >= 64 — the 64->64
// and 64->216 convs that make up most of the full-resolution work of the TDVC path.
//
// v3 stamps: with two 4-wave workgroups per CU, every stage re-pulled its 36 KB of
// weights from a handful of hot L2 lines and the load-issue step alone stalled 4.3k cycles.  Here
//   * one persistent 8-wave workgroup per CU keeps ALL of the layer's weights for its 64 output
//     channels in LDS (<= 72 KB, loaded once), in MFMA fragment order;
//   * stage = (16x32-pixel tile, 32-channel chunk): only the halo tile (18x34 px x 80 B = 48 KB) is
//     streamed, prefetched into registers (5 x 16 B per thread) during the previous stage's matrix phase;
//   * matrix phase: LDS reads + MFMA only, no barriers; wave w owns rows 2w, 2w+1 (64 px) x 64 channels;
//   * epilogue through wave-private LDS scratch, full 128-byte line stores.
#include <type_traits>

#include "conv_common.h"

using convk::ConvParams;

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
static_assert(sizeof(u32x4) == 16, "");

constexpr int TH4 = 16, TW4 = 32, NT4 = 2, CK4 = 32, PS4 = 80, NTHR = 512;
constexpr int WSL4 = 4096;
constexpr int TLOADS4 = ((TH4 + 2) * (TW4 + 2) * 4 + NTHR - 1) / NTHR;     // 5 for 3x3

struct V4Extra {
  int ntiles, tile_bytes;
  int stagger;     // start-phase stagger between workgroups, in units of s_sleep(32) (~2k cycles)
};

static long long* g_stamp4 = nullptr;
static int g_stamp4_cap = 0;

template <bool SIMPLE, bool STAMP = false>
__global__ __launch_bounds__(NTHR, 1) void conv_mfma_v4_kernel(const ConvParams p, const V4Extra e, long long* stamps = nullptr, int stamp_cap = 0) {
  long long stv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ST4(i) do { if constexpr (STAMP) { if (S == 3) stv[i] = clock64(); } } while (0)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* tapoff = reinterpret_cast<int*>(smem);
  float* bias_s = reinterpret_cast<float*>(smem + 256);   // 64 floats: no global load in the epilogue
  unsigned char* tbuf = smem + 512;
  unsigned char* wlds = smem + 512 + e.tile_bytes;       // nchunks x ntaps x 4 KB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, r = lane & 31;
  const int cb = blockIdx.y, n = blockIdx.z;
  const int TIW = TW4 + p.kw - 1, TIH = TH4 + p.kh - 1;
  const int total_items = TIH * TIW * 4;
  const int ntaps = p.ntaps, nchunks = p.nchunks;

  const int first = blockIdx.x, stride = gridDim.x;
  const int my_tiles = (e.ntiles - first + stride - 1) / stride;
  const int nstages = my_tiles * nchunks;
  if (nstages <= 0) return;

  if (tid < 64) bias_s[tid] = p.bias ? p.bias[blockIdx.y * 64 + tid] : 0.f;
  if (tid < ntaps) tapoff[tid] = (p.tap_dy[tid] * TIW + p.tap_dx[tid]) * PS4;

  // ---- resident weights: slice (ch, t) = 4 KB [mt 2][s2 2][lane 64][8 halves] -------------------
  {
    const int nslices = nchunks * ntaps;
    for (int i = tid; i < nslices * 256; i += NTHR) {
      const int sl = i >> 8, u = i & 255;            // u = q*64 + lane, q = mt*2 + s2
      const int qq = u >> 6, ln = u & 63;
      const half_t* src = p.w + ((((long)(cb * 2 + (qq >> 1)) * nchunks * ntaps + sl) * 2 + (qq & 1)) * 64 + ln) * 8;
      *reinterpret_cast<half8*>(wlds + sl * WSL4 + u * 16) = *reinterpret_cast<const half8*>(src);
    }
  }

  int it_rr[TLOADS4], it_c[TLOADS4], it_dst[TLOADS4];
#pragma unroll
  for (int j = 0; j < TLOADS4; ++j) {
    const int idx = j * NTHR + tid;
    const int c8 = idx & 3, pix = idx >> 2;
    it_rr[j] = pix / TIW;
    it_c[j] = pix - it_rr[j] * TIW;
    it_dst[j] = idx < total_items ? pix * PS4 + c8 * 16 : -1;
  }
  const int c8off = (tid & 3) * 8;
  const half_t* xn = p.x + (long)n * p.x_sn;

  // Prefetch: the next stage's tile loads are issued right after this stage is published and stay in
  // flight during the whole matrix phase (plain, compiler-tracked loads).  A depth-2 variant with
  // hand-counted inline-asm loads measured 4 % faster, but asm loads leave their destination registers
  // "valid" to the register allocator before the data has landed — a copy or spill in that window reads
  // stale data (seen once in a sibling kernel) — so it is not used.  Out-of-image items are
  // zero-selected at publish time.
  u32x4 treg[1][TLOADS4];
  unsigned okmask[1] = {0u};
  auto issue = [&](auto setc, int S) {
    constexpr int SET = decltype(setc)::value;
    const int tile_i = S / nchunks, ch = S - tile_i * nchunks;
    const int tile = first + tile_i * stride;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const int iy0 = ty * TH4 - p.pad, ix0 = tx * TW4 - p.pad;
    const int cg = ch * CK4 + c8off;
    const bool cok = cg < p.Cin;
    const int cgc = cok ? cg : 0;
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < TLOADS4; ++j) {
      const int iy = iy0 + it_rr[j], ix = ix0 + it_c[j];
      const bool ok = cok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
      const half_t* src = xn + ((long)iyc * p.W + ixc) * p.x_sp + cgc;
      treg[SET][j] = *reinterpret_cast<const u32x4*>(src);
      m |= (ok ? 1u : 0u) << j;
    }
    okmask[SET] = m;
  };
  using I0 = std::integral_constant<int, 0>;

  int base[NT4];
#pragma unroll
  for (int nt = 0; nt < NT4; ++nt) base[nt] = ((wave * NT4 + nt) * TIW + r) * PS4 + hh * 16;

  f32x16 acc[2][NT4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT4; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  // De-synchronise the persistent workgroups: identical work keeps all CUs in lockstep, so every CU
  // would prefetch / store in the same instant and HBM would alternate between overload and idle.
  for (int i = 0; i < (int)(blockIdx.x & 3) * e.stagger; ++i) __builtin_amdgcn_s_sleep(32);
  issue(I0{}, 0);
  auto stage = [&](auto setc, int S) {
    constexpr int SET = decltype(setc)::value;
    const int tile_i = S / nchunks, ch = S - tile_i * nchunks;
    ST4(0);
    __syncthreads();                     // previous stage's reads / epilogue scratch done (and, at S = 0, weights visible after the next barrier)
    ST4(1);
#pragma unroll
    for (int j = 0; j < TLOADS4; ++j) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      u32x4 val = ((okmask[SET] >> j) & 1u) ? treg[SET][j] : z;
      if (p.square) {                       // GDN norm pool: stage x^2
        half8 hv = __builtin_bit_cast(half8, val);
        hv = hv * hv;
        val = __builtin_bit_cast(u32x4, hv);
      }
      if (it_dst[j] >= 0) *reinterpret_cast<u32x4*>(tbuf + it_dst[j]) = val;
    }
    ST4(2);
    __syncthreads();
    ST4(3);
    if (S + 1 < nstages) issue(setc, S + 1);      // in flight during the matrix phase
    ST4(4);

    const unsigned char* wch = wlds + ch * ntaps * WSL4 + lane * 16;
    for (int t = 0; t < ntaps; ++t) {
      const unsigned char* wslot = wch + t * WSL4;
      const int toff = tapoff[t];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        half8 a[2], b[NT4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const half8*>(wslot + (mt * 2 + s2) * 1024);
#pragma unroll
        for (int nt = 0; nt < NT4; ++nt) b[nt] = *reinterpret_cast<const half8*>(tbuf + base[nt] + toff + s2 * 32);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT4; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
      }
    }

    ST4(5);
    if (ch != nchunks - 1) return;
    const int tile = first + tile_i * stride;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    if constexpr (SIMPLE) {
      __syncthreads();
      convk::epilogue_simple_rows<NT4>(p, acc, bias_s, tbuf + wave * (32 * 144), n, cb * 64,
                                       ty * TH4 + wave * NT4, tx * TW4, lane, true);
    } else {
      const int ox = tx * TW4 + r;
#pragma unroll
      for (int nt = 0; nt < NT4; ++nt) {
        const int oy = ty * TH4 + wave * NT4 + nt;
        const bool ok = oy < p.Ho && ox < p.Wo;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = acc[mt][nt][4 * g + i]; acc[mt][nt][4 * g + i] = 0.f; }
            if (ok) convk::epilogue4(p, n, oy, ox, (cb * 2 + mt) * 32 + 8 * g + 4 * hh, v);
          }
        }
      }
    }
    ST4(6);
    if constexpr (STAMP) {
      if (S == 3 && threadIdx.x == 0) {
        const int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (bid < stamp_cap) for (int i = 0; i < 8; ++i) stamps[(long)bid * 8 + i] = stv[i];
      }
    }
  };
  for (int S = 0; S < nstages; ++S) stage(I0{}, S);
}

inline int v4_tile_bytes(int kh, int kw) { return (TH4 + kh - 1) * (TW4 + kw - 1) * PS4; }
inline int v4_lds_bytes(int kh, int kw, int ntaps, int nchunks) { return 512 + v4_tile_bytes(kh, kw) + nchunks * ntaps * WSL4; }

}  // namespace

extern "C" void tdvc_debug_set_stamp_buffer_v4(void* buf, int cap_blocks) { g_stamp4 = (long long*)buf; g_stamp4_cap = cap_blocks; }

bool conv_v4_eligible(const tdvc_conv_desc* d, int Ho, int Wo) {
  static const bool off = getenv("TDVC_CONV_NO_V4") != nullptr || getenv("TDVC_CONV_V1") != nullptr;
  if (off) return false;
  const int nchunks = (d->x.C + CK4 - 1) / CK4;
  return d->ck == 32 && d->stride == 1 && d->ntaps >= 1 && d->ntaps <= 9 && d->kh <= 3 && d->kw <= 3 && d->cout >= 64 &&
         d->x.C >= 32 && !d->s2d && (long)Ho * Wo >= 8192 &&
         v4_lds_bytes(d->kh, d->kw, d->ntaps, nchunks) <= 150 * 1024;
}

int launch_conv_v4(const ConvParams& p, int cout_blocks, int N, hipStream_t st) {
  ConvParams q = p;
  q.tiles_x = (p.Wo + TW4 - 1) / TW4;
  const int tiles_y = (p.Ho + TH4 - 1) / TH4;
  V4Extra e;
  e.ntiles = q.tiles_x * tiles_y;
  e.tile_bytes = v4_tile_bytes(p.kh, p.kw);
  {
    static const int stg = getenv("TDVC_V4_STAGGER") ? atoi(getenv("TDVC_V4_STAGGER")) : 0;
    e.stagger = stg;
  }
  const int lds = v4_lds_bytes(p.kh, p.kw, p.ntaps, p.nchunks);
  const bool simple = convk::conv_is_simple(p);
  if (simple) q.slope = convk::conv_simple_slope(p);
  int gx = 256 / (cout_blocks * N);
  if (gx < 1) gx = 1;
  if (gx > e.ntiles) gx = e.ntiles;
  dim3 grid(gx, cout_blocks, N);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err == hipSuccess)
      err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) { tdvc_set_error("conv v4: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    attr_done = true;
  }
  if (g_stamp4 && simple) {
    static bool a2 = false;
    if (!a2) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v4_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a2 = true; }
    hipLaunchKernelGGL((conv_mfma_v4_kernel<true, true>), grid, dim3(NTHR), lds, st, q, e, g_stamp4, g_stamp4_cap);
  } else if (simple) hipLaunchKernelGGL((conv_mfma_v4_kernel<true>), grid, dim3(NTHR), lds, st, q, e, (long long*)nullptr, 0);
  else hipLaunchKernelGGL((conv_mfma_v4_kernel<false>), grid, dim3(NTHR), lds, st, q, e, (long long*)nullptr, 0);
  return tdvc_launch_status("tdvc_conv2d(v4)");
}
