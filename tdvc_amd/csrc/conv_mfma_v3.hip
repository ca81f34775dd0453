// conv_mfma_v3 — stage-pipelined implicit-GEMM conv for the 3x3 (<= 9 taps) stride-1 layers that
// dominate the TDVC path (Cin >= 32, Cout >= 64).
//
// Measured on v2 with in-kernel stamps: per 16x32 tile the MFMA phases took ~19k
// cycles but staging bursts took ~32k: every workgroup stages at the same time, the bursts run at
// the HBM fair share, and the matrix pipe idles meanwhile.  v3 removes every global load from the
// matrix phase and puts the NEXT stage's loads under it instead:
//   stage = (tile, 32-channel chunk).  LDS holds the stage's halo tile (10x34 px x 80 B) AND all of
//   its weights (ntaps x 4 KB, fragment order).  Per stage:
//     wait for the prefetched registers -> barrier -> write them to LDS -> barrier ->
//     issue the next stage's loads (6 tile + <=9 weight 16-byte loads per thread, ~60 VGPRs, in
//     flight for the whole matrix phase; vmcnt ordering is harmless because nothing else is loaded)
//     -> 9 taps x 2 k-steps x 4 MFMAs per wave from LDS only, no barriers.
//   Workgroups are persistent over tiles (the prefetch crosses tile boundaries; the epilogue of a tile
//   runs while the next tile's data is in flight); 64 KB of LDS -> two workgroups per CU, so one's
//   barriers / epilogue overlap the other's MFMAs.
//   Epilogue: fp16 through a wave-private LDS region, full 128-byte line stores (as v2).
#include "conv_common.h"

using convk::ConvParams;

namespace {

constexpr int TH3 = 8, TW3 = 32, NT3 = 2, CK3 = 32, PS3 = 80;
constexpr int WSL = 4096;                  // one (chunk, tap) weight slice
constexpr int TILE_ITEMS_MAX = (TH3 + 2) * (TW3 + 2) * 4;     // 1360 for 3x3
constexpr int TLOADS = (TILE_ITEMS_MAX + 255) / 256;          // 6
constexpr int WLOADS = 9;

struct V3Extra {
  int ntiles, tile_bytes;
};

static long long* g_stamp3 = nullptr;
static int g_stamp3_cap = 0;

template <int SIMPLE, bool STAMP = false>      // 0: per-register epilogue4, 1: transposed generic, 2: transposed lean
__global__ __launch_bounds__(256, 2) void conv_mfma_v3_kernel(const ConvParams p, const V3Extra e, long long* stamps = nullptr, int stamp_cap = 0) {
  long long stv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ST3(i) do { if constexpr (STAMP) { if (S == 3) stv[i] = clock64(); } } while (0)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* tapoff = reinterpret_cast<int*>(smem);          // 256 B
  float* bias_s = reinterpret_cast<float*>(smem + 256);   // 64 floats: no global load in the epilogue
  unsigned char* tbuf = smem + 512;                    // tile (also epilogue scratch)
  unsigned char* wlds = smem + 512 + e.tile_bytes;     // ntaps x 4 KB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, r = lane & 31;
  const int cb = blockIdx.y, n = blockIdx.z;
  const int TIW = TW3 + p.kw - 1, TIH = TH3 + p.kh - 1;
  const int total_items = TIH * TIW * 4;
  const int ntaps = p.ntaps, nchunks = p.nchunks;

  int first, stride, my_tiles;                             // XCD-aware: one contiguous band of tiles per L2
  convk::xcd_tile_walk(e.ntiles, first, stride, my_tiles);
  const int nstages = my_tiles * nchunks;
  if (nstages <= 0) return;

  if (tid < 64) bias_s[tid] = p.bias ? p.bias[blockIdx.y * 64 + tid] : 0.f;
  if (tid < ntaps) tapoff[tid] = (p.tap_dy[tid] * TIW + p.tap_dx[tid]) * PS3;

  // ---- per-thread constant staging geometry ---------------------------------------------------
  int it_rr[TLOADS], it_c[TLOADS], it_dst[TLOADS];
#pragma unroll
  for (int j = 0; j < TLOADS; ++j) {
    const int idx = j * 256 + tid;
    const int c8 = idx & 3, pix = idx >> 2;
    it_rr[j] = pix / TIW;
    it_c[j] = pix - it_rr[j] * TIW;
    it_dst[j] = idx < total_items ? pix * PS3 + c8 * 16 : -1;
  }
  const int c8off = (tid & 3) * 8;
  const half_t* xn = p.x + (long)n * p.x_sn;
  // weight slice of tap t for this thread: q = tid>>6 -> (mt = q>>1, s2 = q&1)
  const int q = tid >> 6;
  const half_t* wthread = p.w + ((long)(cb * 2 + (q >> 1)) * nchunks * p.steps * 64 + lane) * 8 + (long)(q & 1) * 512;

  half8 treg[TLOADS], wreg[WLOADS];
  auto issue = [&](int S) {
    const int tile_i = S / nchunks, ch = S - tile_i * nchunks;
    const int tile = p.reverse ? e.ntiles - 1 - (first + tile_i * stride) : first + tile_i * stride;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const int iy0 = ty * TH3 - p.pad, ix0 = tx * TW3 - p.pad;
    int cg = ch * CK3 + c8off;
    int sy = 1, oy_ = 0, ox_ = 0;                 // source pixel = sy * virtual + (oy_, ox_)
    if (p.s2d) {
      const int per = p.Corig / CK3;              // chunks per parity
      const int qd = ch / per;
      cg = (ch - qd * per) * CK3 + c8off;
      sy = 2; oy_ = qd >> 1; ox_ = qd & 1;
    }
#pragma unroll
    for (int j = 0; j < TLOADS; ++j) {
      const int iy = (iy0 + it_rr[j]) * sy + oy_, ix = (ix0 + it_c[j]) * sy + ox_;
      half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (it_dst[j] >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && (p.s2d || cg < p.Cin))
        v = *reinterpret_cast<const half8*>(xn + ((long)iy * p.W + ix) * p.x_sp + cg);
      treg[j] = v;
    }
    const half_t* wc = wthread + (long)ch * ntaps * 1024;
#pragma unroll
    for (int t = 0; t < WLOADS; ++t)
      if (t < ntaps) wreg[t] = *reinterpret_cast<const half8*>(wc + (long)t * 1024);
  };

  int base[NT3];
#pragma unroll
  for (int nt = 0; nt < NT3; ++nt) base[nt] = ((wave * NT3 + nt) * TIW + r) * PS3 + hh * 16;

  f32x16 acc[2][NT3];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT3; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  issue(0);
  for (int S = 0; S < nstages; ++S) {
    const int tile_i = S / nchunks, ch = S - tile_i * nchunks;
    // ---- publish the prefetched stage -------------------------------------------------------
    ST3(0);
    __syncthreads();                     // everyone is done reading the previous stage / epilogue scratch
    ST3(1);
#pragma unroll
    for (int j = 0; j < TLOADS; ++j)
      if (it_dst[j] >= 0) *reinterpret_cast<half8*>(tbuf + it_dst[j]) = treg[j];
#pragma unroll
    for (int t = 0; t < WLOADS; ++t)
      if (t < ntaps) *reinterpret_cast<half8*>(wlds + t * WSL + tid * 16) = wreg[t];
    ST3(2);
    __syncthreads();
    ST3(3);
    if (S + 1 < nstages) issue(S + 1);   // in flight during the whole matrix phase
    ST3(4);

    // ---- matrix phase: LDS only ----------------------------------------------------------------
    // space-to-depth view of a 3x3 stride-2 conv: virtual tap (dy, dx) of parity block (py, px) is original tap
    // (2dy + py - 1, 2dx + px - 1); 7 of the 16 (tap, parity) blocks fall outside the 3x3 window and hold zero weights:
    // skipped (wave-uniform), 9/16 of the MFMAs remain
    unsigned tapmask = ~0u;
    if (p.s2d) {
      const int q = (ch * CK3) / p.Corig, py = q >> 1, px = q & 1;
      tapmask = 0u;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if ((py == 1 || (t >> 1) == 1) && (px == 1 || (t & 1) == 1)) tapmask |= 1u << t;
    }
    for (int t = 0; t < ntaps; ++t) {
      if (!((tapmask >> t) & 1u)) continue;
      const unsigned char* wslot = wlds + t * WSL + lane * 16;
      const int toff = tapoff[t];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        half8 a[2], b[NT3];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const half8*>(wslot + (mt * 2 + s2) * 1024);
#pragma unroll
        for (int nt = 0; nt < NT3; ++nt) b[nt] = *reinterpret_cast<const half8*>(tbuf + base[nt] + toff + s2 * 32);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT3; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
      }
    }

    ST3(5);
    if constexpr (STAMP) {
      if (S == 3 && threadIdx.x == 0) {
        const int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (bid < stamp_cap) for (int i = 0; i < 8; ++i) stamps[(long)bid * 8 + i] = stv[i];
      }
    }
    if (ch != nchunks - 1) continue;
    // ---- tile finished: epilogue -----------------------------------------------------------------
    const int tile = p.reverse ? e.ntiles - 1 - (first + tile_i * stride) : first + tile_i * stride;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    if constexpr (SIMPLE) {
      __syncthreads();                   // the tile buffer becomes per-wave scratch
      convk::epilogue_simple_rows<NT3, false, SIMPLE>(p, acc, bias_s, tbuf + wave * (32 * 144), n, cb * 64,
                                       ty * TH3 + wave * NT3, tx * TW3, lane, true);
    } else {
      const int ox = tx * TW3 + r;
#pragma unroll
      for (int nt = 0; nt < NT3; ++nt) {
        const int oy = ty * TH3 + wave * NT3 + nt;
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
  }
}

inline int v3_tile_bytes(int kh, int kw) { return (TH3 + kh - 1) * (TW3 + kw - 1) * PS3; }
inline int v3_lds_bytes(int kh, int kw, int ntaps) { return 512 + v3_tile_bytes(kh, kw) + ntaps * WSL; }

}  // namespace

extern "C" void tdvc_debug_set_stamp_buffer_v3(void* buf, int cap_blocks) { g_stamp3 = (long long*)buf; g_stamp3_cap = cap_blocks; }

bool conv_v3_eligible(const tdvc_conv_desc* d, int Ho, int Wo) {
  static const bool off = getenv("TDVC_CONV_NO_V3") != nullptr || getenv("TDVC_CONV_V1") != nullptr;
  if (off) return false;
  const long minpix = 256;        // below: a handful of tiles, the direct kernel's shorter prologue wins
  return d->ck == 32 && d->stride == 1 && d->ntaps >= 2 && d->ntaps <= WLOADS && d->kh <= 3 && d->kw <= 3 && d->cout >= 64 &&
         d->x.C >= 32 && !d->square_input && ((long)Ho * Wo * d->x.N >= minpix || d->s2d);
}

int launch_conv_v3(const ConvParams& p, int cout_blocks, int N, hipStream_t st) {
  ConvParams q = p;
  q.tiles_x = (p.Wo + TW3 - 1) / TW3;
  const int tiles_y = (p.Ho + TH3 - 1) / TH3;
  V3Extra e;
  e.ntiles = q.tiles_x * tiles_y;
  e.tile_bytes = v3_tile_bytes(p.kh, p.kw);
  const int lds = v3_lds_bytes(p.kh, p.kw, p.ntaps);
  const bool simple = convk::conv_is_simple(p);
  if (simple) q.slope = convk::conv_simple_slope(p);
  // persistent grid: two workgroups per CU over all (cout block, image) pairs
  int gx = 512 / (cout_blocks * N);
  if (gx < 1) gx = 1;
  if (gx > e.ntiles) gx = e.ntiles;
  dim3 grid(gx, cout_blocks, N);
  static TdvcPerDeviceFlag attr_flags;
  bool& attr_done = attr_flags.flag();
  if (!attr_done) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (err == hipSuccess)
      err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (err == hipSuccess)
      err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (err != hipSuccess) { tdvc_set_error("conv v3: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    attr_done = true;
  }
  const bool lean = simple && convk::conv_is_lean(p);
  if (g_stamp3 && simple) {
    static bool a2 = false;
    if (!a2) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v3_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); a2 = true; }
    hipLaunchKernelGGL((conv_mfma_v3_kernel<1, true>), grid, dim3(256), lds, st, q, e, g_stamp3, g_stamp3_cap);
  } else if (lean) hipLaunchKernelGGL((conv_mfma_v3_kernel<2>), grid, dim3(256), lds, st, q, e, (long long*)nullptr, 0);
  else if (simple) hipLaunchKernelGGL((conv_mfma_v3_kernel<1>), grid, dim3(256), lds, st, q, e, (long long*)nullptr, 0);
  else hipLaunchKernelGGL((conv_mfma_v3_kernel<0>), grid, dim3(256), lds, st, q, e, (long long*)nullptr, 0);
  return tdvc_launch_status("tdvc_conv2d(v3)");
}
