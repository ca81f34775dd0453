// conv_mfma_v10 — 3x3, stride 1, pad 1, Cin = 64, Cout >= 64 (the 64->64 / 64->216 convs: the dominant layer class),
// successor of conv_mfma_v7 for that class.
//
// What bounded v7 (rocprofv3 PMC + in-kernel stamps): 39 % MFMA busy; per tile 2 x 6.1k cycles of matrix
// phase (ideal 2 x 4.6k: the younger wave of each SIMD loses VALU / LDS arbitration and the stage barrier waits for it)
// + 5.6k of epilogue in which no wave issues an MFMA.  v7's wave computes 64 couts x (2 rows x 32 px): one LDS fragment
// read per MFMA, and the two waves of a SIMD fight for the issue port.
//
// v10 keeps v7's memory system (persistent workgroup per CU, all weights resident in LDS, 16x32-px tiles in 32-channel
// stages by LDS-DMA into two buffers, XOR-swizzled lane-linear image, counted vmcnt + one barrier per stage, XCD-aware
// tile walk) and changes the register tile:
//   * 4 waves, ONE PER SIMD (256 threads, up to 512 VGPRs): a wave owns the matrix pipe of its SIMD, nobody arbitrates;
//   * each wave computes 64 couts x (4 rows x 32 px): 2 x 4 accumulators (128 registers);
//   * row reuse: for a fixed (k-half s2, dx) the B fragment of input row ir serves output rows ir, ir-1, ir-2 (dy = 0, 1, 2),
//     so a group of 24 MFMAs needs 6 A reads (3 dy x 2 mt) + 6 B reads (input rows 0..5): 0.5 LDS reads per MFMA
//     (v7: 1.0) and half the address arithmetic; the next group's 12 reads are issued under the current group's MFMAs
//     (768 cycles of cover for ~100 cycles of LDS latency);
//   * epilogue: two rows at a time through two wave-private LDS regions (the finished tile buffer).
// LDS map (bytes) as v7: [0, 40K) tile buffer 0 | [40K, 64K) weight slices 0..5 | [64K, 104K) tile buffer 1 |
// [104K, 152K) weight slices 6..17 | [152K, +512) bias, so that switching buffers is `addr ^ 0x10000`.
//
//   * memory traffic kept off the critical path: residual rows are fetched at the start of a tile's second stage, the
//     epilogue only packs / transposes / adds, and the tile's 16 stores go out under the next tile's first stage; every
//     vector-memory wait is a plain vmcnt(0) at a stage top (no hand-counted wait: see "epilogue state" in the kernel).
#include <type_traits>

#include "conv_common.h"

using convk::ConvParams;

namespace {

constexpr int TH10 = 16, TW10 = 32, NT10 = 4, NW10 = 4, CK10 = 32, NTHR10 = 256;
constexpr int TIW10 = TW10 + 2, TIH10 = TH10 + 2, NPIX10 = TIW10 * TIH10;   // 34 x 18 = 612 halo pixels
constexpr int PIECES10 = (NPIX10 + 15) / 16;                                // 39 DMA pieces of 16 pixels x 64 B
constexpr int DMA10 = (PIECES10 + NW10 - 1) / NW10;                         // 10 per wave
constexpr int BUF1_10 = 0x10000, WLO10 = 40 * 1024, WHI10 = 104 * 1024, MISC10 = 152 * 1024;
constexpr int LDS10 = MISC10 + 512;
constexpr int WSL10 = 4096;
constexpr int EROW10 = 32 * 144;                                            // one transposed output row (epilogue scratch)
static_assert(DMA10 * NW10 * 1024 <= WLO10, "tile buffer");
static_assert(NW10 * 2 * EROW10 <= WLO10, "epilogue scratch aliases a tile buffer");

struct V10Extra {
  int ntiles;
  const half_t* zeros;      // >= 16 bytes of zeros: the DMA source of out-of-image halo pixels
};

static long long* g_stamp10 = nullptr;
static int g_stamp10_cap = 0;

__device__ __host__ constexpr int wslice10(int sl) { return sl < 6 ? WLO10 + sl * WSL10 : WHI10 + (sl - 6) * WSL10; }

__device__ __forceinline__ void glds16_10(const half_t* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void raw_barrier10() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int NRES, bool STAMP = false>
__global__ __launch_bounds__(NTHR10, 1) void conv_mfma_v10_kernel(const ConvParams p, const V10Extra e, long long* stamps = nullptr, int stamp_cap = 0) {
  long long stv[24];
  if constexpr (STAMP) { for (int i = 0; i < 24; ++i) stv[i] = 0; }
#define ST10(i) do { if constexpr (STAMP) { if (ti == 1 && ch == 1) stv[i] = clock64(); } } while (0)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* bias_s = reinterpret_cast<float*>(smem + MISC10);   // 64 floats
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, r = lane & 31;
  const int cb = blockIdx.y, n = blockIdx.z;
  constexpr int nchunks = 2;                                 // Cin = 64 (eligibility)

  int first, stride, my_tiles;                               // XCD-aware: one contiguous band of tiles per L2
  convk::xcd_tile_walk(e.ntiles, first, stride, my_tiles);
  if (my_tiles <= 0) return;

  // ---- per-lane DMA item: piece u = j * 4 + wave covers halo pixels q = 16u .. 16u + 15; this lane moves 16-byte slot
  // (lane & 3) of pixel q = 16u + (lane >> 2), which holds logical chunk c = slot ^ ((q >> 2) & 3).
  const int csw = (lane & 3) ^ ((lane >> 4) & 3);            // (q >> 2) & 3 == (lane >> 4) & 3 for every piece
  int it_off[DMA10];
#pragma unroll
  for (int j = 0; j < DMA10; ++j) {
    const int q = 16 * (j * NW10 + wave) + (lane >> 2);
    const int rr = q / TIW10, cc = q - rr * TIW10;
    it_off[j] = q < NPIX10 ? (rr * p.W + cc) * p.x_sp + csw * 8 : csw * 8;
  }
  const half_t* xn = p.x + (long)n * p.x_sn;

  // ---- the tile whose DMA is being issued: (pf_ty, pf_tx) advance by `stride` tiles without a division
  int pf_iy0 = 0, pf_ix0 = 0, pf_ch = 0;
  bool pf_interior = false;
  const half_t* pf_base = xn;
  auto issue_prep = [&](int ty, int tx, int ch) {
    pf_iy0 = ty * TH10 - 1;
    pf_ix0 = tx * TW10 - 1;
    pf_ch = ch;
    pf_interior = pf_iy0 >= 0 && pf_ix0 >= 0 && pf_iy0 + TIH10 <= p.H && pf_ix0 + TIW10 <= p.W;
    pf_base = xn + ((long)pf_iy0 * p.W + pf_ix0) * p.x_sp + ch * CK10;    // only dereferenced when interior
  };
  auto issue_one = [&](int j, unsigned dst_buf) {            // j is a compile-time constant at every call site
    const half_t* src = pf_base + it_off[j];
    if (!pf_interior) {                  // uniform branch: border tiles (6 % at 1080p) clamp per lane
      const int q = 16 * (j * NW10 + wave) + (lane >> 2);
      const int rr = q / TIW10;
      const int iy = pf_iy0 + rr, ix = pf_ix0 + (q - rr * TIW10);
      const bool ok = q < NPIX10 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      src = ok ? xn + ((long)iy * p.W + ix) * p.x_sp + pf_ch * CK10 + csw * 8 : e.zeros;
    }
    glds16_10(src, dst_buf + (j * NW10 + wave) * 1024);
  };

  // ---- prologue: first tile's DMA, then bias + resident weights (compiler-tracked loads, younger than the DMA)
  const int tile0 = p.reverse ? e.ntiles - 1 - first : first;   // alternate launches walk the raster backwards (tdvc_conv2d)
  const int tstep = p.reverse ? -stride : stride;
  int ty = tile0 / p.tiles_x, tx = tile0 - ty * p.tiles_x;      // the tile being computed
  issue_prep(ty, tx, 0);
#pragma unroll
  for (int j = 0; j < DMA10; ++j) issue_one(j, lds0);

  if (tid < 64) bias_s[tid] = p.bias[blockIdx.y * 64 + tid];
  {
    constexpr int nslices = nchunks * 9;             // slice (ch, t) = 4 KB [mt 2][s2 2][lane 64][8 halves]
    for (int i = tid; i < nslices * 256; i += NTHR10) {
      const int sl = i >> 8, u = i & 255;            // u = q*64 + lane, q = mt*2 + s2
      const int qq = u >> 6, ln = u & 63;
      const half_t* src = p.w + ((((long)(cb * 2 + (qq >> 1)) * nslices + sl) * 2 + (qq & 1)) * 64 + ln) * 8;
      *reinterpret_cast<half8*>(smem + wslice10(sl) + u * 16) = *reinterpret_cast<const half8*>(src);
    }
  }

  // ---- B-fragment read offsets (tile buffer 0, s2 = 0): input row ir of this wave, column r + dx, chunk slot hh
  int bq[NT10 + 2][3];
#pragma unroll
  for (int ir = 0; ir < NT10 + 2; ++ir)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int q = (wave * NT10 + ir) * TIW10 + r + dx;
      bq[ir][dx] = q * 64 + ((hh ^ ((q >> 2) & 3)) << 4);
    }

  __syncthreads();                       // bias and weights are visible (no DMA-aware wait here: see top of stage)

  // Accumulators are never re-initialised: the first MFMA of every accumulator chain in a tile (stage 0, group 0, dy of its
  // first visit) takes the bias vector as its C operand.  bias16[mt] holds, per lane, the 16 biases of the accumulator rows
  // (channels mt*32 + 8g + 4hh + i) and stays in registers for the whole launch.
  f32x16 acc[2][NT10];
  f32x16 bias16[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + mt * 32 + 8 * g + hh * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) bias16[mt][4 * g + i] = b4[i];
    }

  // ---- epilogue state.  Every CU moves ~10 B per cycle to / from HBM: a tile's 64 KB of residual reads and 64 KB of
  // stores issued as one burst after the matrix phase cost 6-7 k cycles each in which no MFMA runs (stamps).
  // So the residuals of tile i are fetched into registers at the START of its second stage (4.6 k cycles of cover), the
  // epilogue only packs / transposes / adds into `pend`, and the 16 stores of tile i go out four per matrix group under
  // the FIRST stage of tile i+1.  All vector-memory waits are then vmcnt(0) at a stage top, where the youngest
  // operations are DMA pieces issued >= one group (768 cycles) after the last store: no hand-counted wait is left.
  const int chunk = lane & 7, prow = lane >> 3;
  const int co = cb * 64 + chunk * 8;
  const bool ch_ok = co < p.y.C && co < ((p.cout + 63) & ~63);
  const int ccl = ch_ok ? co : 0;
  half8 pend[NT10][4];                   // finished output rows of the previous tile (this lane's 8 channels of 4 pixels per row)
  int pend_opix[NT10][4];
  unsigned pend_mask = 0;                // bit (j * 4 + k): store (row j, pixel group k); 0 = nothing pending
  half8 r1[NT10][4], r2[NT10][4];
  int cur_opix[NT10][4];
  unsigned cur_mask = 0;
  half_t* yb = reinterpret_cast<half_t*>(p.y.p) + (long)n * p.y.sn + ccl;
  auto store_pending = [&](int j) {      // row j of the pending tile: 4 full-line stores
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (pend_mask & (1u << (j * 4 + k))) *reinterpret_cast<half8*>(yb + (long)pend_opix[j][k] * p.y.sp) = pend[j][k];
  };

  for (int ti = 0; ti < my_tiles; ++ti) {
    const bool last_tile = ti + 1 == my_tiles;
    int nty = ty, ntx = tx + tstep;      // the next tile of this workgroup
    while (ntx >= p.tiles_x) { ntx -= p.tiles_x; ++nty; }
    while (ntx < 0) { ntx += p.tiles_x; --nty; }

    auto stage = [&](auto CH) {
      constexpr int ch = decltype(CH)::value;
      constexpr unsigned bsel = ch ? BUF1_10 : 0;
      ST10(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's DMA pieces of the stage (and every older store / load) are done
      ST10(1);
      raw_barrier10();                   // every wave's pieces landed; every wave is done with the other buffer
      ST10(2);
      const bool have_next = ch == 0 || !last_tile;
      if (ch == 0) issue_prep(ty, tx, 1);
      else if (have_next) issue_prep(nty, ntx, 0);
      const unsigned nbuf = lds0 + (bsel ^ BUF1_10);
      if constexpr (ch == 1) {           // this tile's output geometry + residual prefetch (consumed by the epilogue below)
        const int oy = ty * TH10 + wave * NT10, ox_first = tx * TW10;
        const bool full = (ty + 1) * TH10 <= p.Ho && (tx + 1) * TW10 <= p.Wo;
        if (full) {                      // wave-uniform: 94 % of the tiles at 1080p -- two adds per pixel group, no compares
          const int base = oy * p.Wo + ox_first + prow;
#pragma unroll
          for (int j = 0; j < NT10; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) cur_opix[j][k] = ch_ok ? base + j * p.Wo + k * 8 : 0;
          cur_mask = ch_ok ? 0xFFFFu : 0u;
        } else {
          cur_mask = 0;
#pragma unroll
          for (int j = 0; j < NT10; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int ox = ox_first + k * 8 + prow;
              const bool ok = ch_ok && oy + j < p.Ho && ox < p.Wo;
              cur_opix[j][k] = ok ? (oy + j) * p.Wo + ox : 0;
              cur_mask |= ok ? (1u << (j * 4 + k)) : 0u;
            }
        }
      }
      // the residual rows of this tile, all at the start of its second stage, when this wave has nothing outstanding (just
      // after vmcnt(0)): one row per matrix group was tried and is SLOWER (14.2 k -> 20.6 k cycles for stage + epilogue) -- behind
      // four DMA pieces and eight loads the next loads wait ~3 k cycles at issue: a wave keeps only ~12-16 vector-memory
      // instructions in flight
      if constexpr (ch == 1 && NRES >= 1) {
        const half_t* rb = reinterpret_cast<const half_t*>(p.res.p) + (long)n * p.res.sn + ccl;
#pragma unroll
        for (int j = 0; j < NT10; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) r1[j][k] = *reinterpret_cast<const half8*>(rb + (long)cur_opix[j][k] * p.res.sp);
      }
      if constexpr (ch == 1 && NRES >= 2) {
        const half_t* rb = reinterpret_cast<const half_t*>(p.res2.p) + (long)n * p.res2.sn + ccl;
#pragma unroll
        for (int j = 0; j < NT10; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) r2[j][k] = *reinterpret_cast<const half8*>(rb + (long)cur_opix[j][k] * p.res2.sp);
      }

      // matrix phase: 6 groups (s2, dx) of 24 MFMAs; group g+1's 12 fragment reads are issued under group g's MFMAs
      const unsigned char* tb = smem + bsel;
      const unsigned char* wlo = smem + (ch == 0 ? WLO10 : wslice10(9)) + lane * 16;                          // taps 0..5 of this chunk
      const unsigned char* whi = smem + (ch == 0 ? wslice10(6) - 6 * WSL10 : wslice10(9)) + lane * 16;        // taps 6..8
      half8 fa[2][3][2], fb[2][NT10 + 2];
      auto load_group = [&](int g, int buf) {
        const int s2 = g / 3, dx = g - 3 * s2;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int t = dy * 3 + dx;
          const unsigned char* wt = (t < 6 ? wlo : whi) + t * WSL10;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) fa[buf][dy][mt] = *reinterpret_cast<const half8*>(wt + (mt * 2 + s2) * 1024);
        }
#pragma unroll
        for (int ir = 0; ir < NT10 + 2; ++ir) fb[buf][ir] = *reinterpret_cast<const half8*>(tb + (bq[ir][dx] ^ (s2 * 32)));
      };
      load_group(0, 0);
      __builtin_amdgcn_sched_barrier(0);   // group 0's own 12 reads go out back to back (one latency), not read-by-read
                                           // between its MFMAs: the interleave pattern below is for the NEXT group's reads
#pragma unroll
      for (int g = 0; g < 6; ++g) {
        if (g + 1 < 6) load_group(g + 1, (g + 1) & 1);
#pragma unroll
        for (int ir = 0; ir < NT10 + 2; ++ir)
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int nt = ir - dy;
            if (nt >= 0 && nt < NT10) {
              // the chain of acc[.][nt] starts at (stage 0, group 0) with its smallest dy, i.e. dy == 0 (ir == nt)
              const bool first = ch == 0 && g == 0 && dy == 0;
#pragma unroll
              for (int mt = 0; mt < 2; ++mt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][dy][mt], fb[g & 1][ir], first ? bias16[mt] : acc[mt][nt], 0, 0, 0);
            }
          }
        // pin the software pipeline: one fragment read of group g+1 per two MFMAs of group g
        if (g + 1 < 6) {
#pragma unroll
          for (int k = 0; k < 12; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          }
        }
        if constexpr (ch == 0) {           // the previous tile's output: one row (4 stores) per group, all older than
          if (g < NT10 && pend_mask) store_pending(g);     // the group-4 DMA pieces the next stage top waits for
        }
        if (have_next && g < 5) {          // 10 DMA pieces of the next stage, two per group
          issue_one(2 * g, nbuf);
          issue_one(2 * g + 1, nbuf);
        }
        ST10(8 + g);
      }
      ST10(3);
    };
    stage(std::integral_constant<int, 0>{});
    stage(std::integral_constant<int, 1>{});

    {
      constexpr int ch = 1;
      raw_barrier10();                     // all waves finished reading tile buffer 1: it becomes epilogue scratch
      ST10(4);
      // epilogue: pack (fp16, packed activation) -> wave-private LDS rows -> 8 channels x 4 pixels per lane -> + residuals -> pend
      constexpr int EPS = 144;
      unsigned char* ew = smem + BUF1_10 + wave * (2 * EROW10);
      const half_t sl = (half_t)p.slope;
      const half2v sl2 = {sl, sl};
      const bool act = p.slope != 1.f;     // wave-uniform: no activation -> no packed math at all
      const bool relu = p.slope == 0.f;    // ReLU: one packed max against zero (half of the 64-channel layers)
      const half2v zero2 = {(half_t)0.f, (half_t)0.f};
#pragma unroll
      for (int j0 = 0; j0 < NT10; j0 += 2) {
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              half2v lo = {(half_t)acc[mt][j][4 * g + 0], (half_t)acc[mt][j][4 * g + 1]};
              half2v hi = {(half_t)acc[mt][j][4 * g + 2], (half_t)acc[mt][j][4 * g + 3]};
              if (relu) {
                lo = __builtin_elementwise_max(lo, zero2);
                hi = __builtin_elementwise_max(hi, zero2);
              } else if (act) {
                lo = __builtin_elementwise_max(lo, lo * sl2);
                hi = __builtin_elementwise_max(hi, hi * sl2);
              }
              half4 o = {lo[0], lo[1], hi[0], hi[1]};
              *reinterpret_cast<half4*>(ew + (j & 1) * EROW10 + r * EPS + (mt * 32 + 8 * g + 4 * hh) * 2) = o;
            }
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            half8 v = *reinterpret_cast<const half8*>(ew + (j & 1) * EROW10 + (k * 8 + prow) * EPS + chunk * 16);
            if constexpr (NRES >= 1) v = v + r1[j][k];
            if constexpr (NRES >= 2) v = v + r2[j][k];
            pend[j][k] = v;
            pend_opix[j][k] = cur_opix[j][k];
          }
      }
      pend_mask = cur_mask;
      ST10(5);
      if constexpr (STAMP) {
        if (ti == 1 && lane == 0) {          // one record per wave: [block][wave][24 stamps]
          const int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
          if (bid * 4 + 3 < stamp_cap) for (int i = 0; i < 24; ++i) stamps[((long)bid * 4 + wave) * 24 + i] = stv[i];
        }
      }
    }
    ty = nty;
    tx = ntx;
  }
#pragma unroll
  for (int j = 0; j < NT10; ++j) store_pending(j);           // the last tile's rows
}

}  // namespace

extern "C" void tdvc_debug_set_stamp_buffer_v10(void* buf, int cap_blocks) { g_stamp10 = (long long*)buf; g_stamp10_cap = cap_blocks; }

static bool g_v10_enabled = true;
// tests and A/B benchmarks switch the kernel off to send the same layers to conv_mfma_v7
extern "C" void tdvc_debug_enable_conv_v10(int enable) { g_v10_enabled = enable != 0; }

bool conv_v10_eligible(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo) {
  static const bool off = getenv("TDVC_CONV_NO_V10") != nullptr || getenv("TDVC_CONV_V1") != nullptr;
  if (off || !g_v10_enabled) return false;
  bool taps33 = d->ntaps == 9 && d->kh == 3 && d->kw == 3 && d->pad == 1;
  for (int t = 0; taps33 && t < 9; ++t) taps33 = d->tap_dy[t] == t / 3 && d->tap_dx[t] == t % 3;
  return taps33 && d->ck == 32 && d->stride == 1 && d->cout >= 64 && d->x.C == 64 && !d->s2d &&
         !d->square_input && (long)Ho * Wo >= 8192 && convk::conv_is_lean(p);
}

int launch_conv_v10(const ConvParams& p, int cout_blocks, int N, hipStream_t st) {
  const void* zeros = nullptr;
  if (const int zrc = tdvc_scratch_pages(&zeros, nullptr)) return zrc;
  ConvParams q = p;
  q.tiles_x = (p.Wo + TW10 - 1) / TW10;
  const int tiles_y = (p.Ho + TH10 - 1) / TH10;
  V10Extra e;
  e.ntiles = q.tiles_x * tiles_y;
  e.zeros = reinterpret_cast<const half_t*>(zeros);
  q.slope = convk::conv_simple_slope(p);
  int gx = 256 / (cout_blocks * N);
  if (gx < 1) gx = 1;
  if (gx > e.ntiles) gx = e.ntiles;
  dim3 grid(gx, cout_blocks, N);
  auto go = [&](auto kern, bool stamp) -> int {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) { tdvc_set_error("conv v10: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    hipLaunchKernelGGL(kern, grid, dim3(NTHR10), LDS10, st, q, e, stamp ? g_stamp10 : (long long*)nullptr, stamp ? g_stamp10_cap : 0);
    return 0;
  };
  const int nres = (p.res.p ? 1 : 0) + (p.res2.p ? 1 : 0);
  if (nres == 1 && !p.res.p) { q.res = q.res2; q.res2 = null_fmap(); }        // a single residual is always `res`
  int rc;
  if (g_stamp10) rc = nres == 0 ? go(&conv_mfma_v10_kernel<0, true>, true) : nres == 1 ? go(&conv_mfma_v10_kernel<1, true>, true) : go(&conv_mfma_v10_kernel<2, true>, true);
  else rc = nres == 0 ? go(&conv_mfma_v10_kernel<0, false>, false) : nres == 1 ? go(&conv_mfma_v10_kernel<1, false>, false) : go(&conv_mfma_v10_kernel<2, false>, false);
  if (rc) return rc;
  return tdvc_launch_status("tdvc_conv2d(v10)");
}
