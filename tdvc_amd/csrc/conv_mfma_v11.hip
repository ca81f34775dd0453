// conv_mfma_v11 — 3x3, stride 1, pad 1, Cin a multiple of 32 with Cin >= 128 (the 128->128 / 128->64 convs of the coders, MCNet
// and the in-loop filter: 7.7 ms per 1080p frame on conv_mfma_v3 at 0.27-0.29 of the MFMA peak), Cout >= 64, lean epilogue
// (fp16 NHWC or PixelShuffle(2) store, no GDN).
//
// These layers are compute-bound (575 FLOP/B at 128->128) but cannot be weight-stationary: 64 couts x 128 cin x 9 taps is
// 144 KB.  v3 stages tile AND weights of every 32-channel stage global -> VGPR -> LDS (measured: 4.3 k cycles of load-issue
// back-pressure per 5 k-cycle stage).  v11 keeps v7's pipeline and lets the LDS-DMA carry both:
//   * a stage = one 32-channel chunk of a 16x32-px tile (39 KB halo image, XOR-swizzled as in v7) + the 64-cout weight
//     slice of that chunk (36 KB = 2 cout tiles x 18 k-steps x 1 KB, already in MFMA fragment order in global memory, so
//     a DMA piece IS a fragment); two stage buffers (2 x 76 KB); stage S+1 streams in during stage S's matrix phase:
//     75 pieces, waves 0-3 the tile, waves 4-7 the weights (affine slot addressing), hidden behind the partner wave of each SIMD;
//   * 8 waves = 2 cout tiles x 4 row groups; a wave computes 32 couts x (4 rows x 32 px): four accumulators, and for a
//     fixed (k-half, dx) the B fragment of input row ir serves output rows ir, ir-1, ir-2: 3 A + 6 B reads per 12 MFMAs
//     (0.75 LDS reads per MFMA; v3 / v7: 1.0);
//   * accumulators start from the bias; lean packed-fp16 epilogue through the finished tile
//     buffer: 32 couts = 64-byte half lines, 2 stores per row, 8 per full tile -- the count the top-of-stage
//     `s_waitcnt vmcnt(8)` lets stay in flight (checked in the listing at build time, Makefile / check_asm.py).
// LDS map (bytes): [0, 40K) tile 0 | [40K, 76K) weights 0 | [80K, 120K) tile 1 | [120K, 156K) weights 1 | [156K, +256) bias.
#include "conv_common.h"

using convk::ConvParams;

namespace {

constexpr int TH11 = 16, TW11 = 32, NT11 = 4, NW11 = 8, CK11 = 32, NTHR11 = 512;
constexpr int TIW11 = TW11 + 2, TIH11 = TH11 + 2, NPIX11 = TIW11 * TIH11;   // 34 x 18 = 612 halo pixels
constexpr int TPIECES11 = (NPIX11 + 15) / 16;                               // 39 tile pieces of 16 pixels x 64 B
constexpr int WPIECES11 = 2 * 18;                                           // 36 weight pieces (cout tile, k-step)
constexpr int PIECES11 = TPIECES11 + WPIECES11;                             // 75 per stage
constexpr int DMA11 = (PIECES11 + NW11 - 1) / NW11;                         // 10 slots per wave
constexpr int TILE0_11 = 0, W0_11 = 40 * 1024, BUFSTEP11 = 80 * 1024, MISC11 = 156 * 1024;
constexpr int LDS11 = MISC11 + 256;
constexpr int EPS11 = 80, EROW11 = 32 * EPS11;                              // transposed half-line rows (32 px x 64 B + pad)
static_assert(TPIECES11 * 1024 <= W0_11 && W0_11 + WPIECES11 * 1024 <= BUFSTEP11, "stage buffer layout");
static_assert(NW11 * 2 * EROW11 <= W0_11, "epilogue scratch aliases a tile buffer");

struct V11Extra {
  int ntiles;
  int experiment;           // 0 in every product launch (see g_v11_experiment)
  const half_t* zeros;      // >= 16 bytes of zeros: the DMA source of out-of-image halo pixels
};

__device__ __forceinline__ void glds16_11(const half_t* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void raw_barrier11() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

static long long* g_stamp11 = nullptr;
static int g_stamp11_cap = 0;
static int g_v11_experiment = 0;     // diagnostic builds only (wrong results): 1 = no weight DMA, 2 = no DMA at all, 3 = one fragment read per group

// Measured and dropped (round 3, tools/ab_v11.py, interleaved rounds in one process): the next stage's 10 DMA slots under matrix
// groups 0-2 (4 + 3 + 3) instead of two per group under groups 0-4 -- 157.3 against 156.0 us at 128 -> 128 @544x960, 321 against
// 311 us at 128 -> 64 @1088x1920: the stage-top wait is 130 of 8 500 cycles per stage, there is nothing to gain there.  As a
// RUN-TIME switch the variant made this kernel, which sits at 247 of 256 VGPRs, spill 18 vector and 47 scalar registers (137
// instead of 105 us per launch in the frame): switches in this kernel are template parameters.
// Also measured and dropped (round 3): the hypothesis that both waves of a SIMD sit in their two LDS-DMA pieces at the end of
// every matrix group TOGETHER (stamps: the matrix phase of a stage takes 7.9-8.3 k cycles for 4.6 k cycles of MFMA on the SIMD;
// a piece costs its issuing wave 100-185 cycles).  Letting waves 4-7 issue their pieces in the MIDDLE of the group (after 6 of
// its 12 MFMAs) was 2-3 % SLOWER on four layer shapes (162.3 against 158.3 us at 128 -> 128 @544x960): the asm statement in
// the middle of the group splits the scheduling region and the pinned read / MFMA interleave degenerates into read pairs.
// Measured and dropped (round 3, tools/bench_v11_res.py, two builds back to back on one box): the residual rows of the epilogue
// requested two output rows ahead of their use (a ring of two rows; all four at once spill 13 registers that are then reloaded
// inside the matrix phase): 167.5 / 167.9 against 172.7 / 167.5 us at 128 -> 128 @544x960 with one residual -- no difference.
template <bool STAMP = false>
__global__ __launch_bounds__(NTHR11, 1) void conv_mfma_v11_kernel(const ConvParams p, const V11Extra e, long long* stamps = nullptr, int stamp_cap = 0) {
  long long stv[16];
  if constexpr (STAMP) { for (int i = 0; i < 16; ++i) stv[i] = 0; }
  // instrumented: the LAST stage of the workgroup's second tile (stamps 0..5) and the stage before it (stamps 8..11)
#define ST11(i) do { if constexpr (STAMP) { if (S == 2 * nchunks - 1) stv[i] = clock64(); else if (S == 2 * nchunks - 2 && (i) < 4) stv[8 + (i)] = clock64(); } } while (0)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* bias_s = reinterpret_cast<float*>(smem + MISC11);   // 64 floats
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave >> 2, rg = wave & 3;                   // cout tile (32 channels), row group (4 rows)
  const int hh = lane >> 5, r = lane & 31;
  const int cb = blockIdx.y, n = blockIdx.z;
  const int nchunks = p.nchunks;

  int first, stride, my_tiles;                               // XCD-aware: one contiguous band of tiles per L2
  convk::xcd_tile_walk(e.ntiles, first, stride, my_tiles);
  const int nstages = my_tiles * nchunks;
  if (nstages <= 0) return;

  // ---- DMA roles.  Waves 0-3 move the TILE: slot j of wave w is piece u = j * 4 + w (10 slots, u < 39): halo pixels 16u ..
  // 16u + 15; this lane moves 16-byte slot (lane & 3) of pixel 16u + (lane >> 2), logical chunk slot ^ ((q >> 2) & 3).
  // Waves 4-7 move the WEIGHTS: slot j of wave 4 + w is fragment (cout tile w & 1, k-step 2j + (w >> 1)), 9 slots: source and
  // destination are affine in j with wave-constant bases (a piece is lane-linear: the fragment as it is read), so a slot
  // costs one 64-bit add and the M0 write -- the mixed assignment of the first version spent ~35 scalar instructions and
  // 5 v_readlane (spilled SGPRs) per slot.
  const bool tile_role = wave < 4;                           // wave-uniform
  const int w3 = wave & 3;
  const int csw = (lane & 3) ^ ((lane >> 4) & 3);
  const int q0 = 16 * w3 + (lane >> 2);                      // halo pixel of this lane in slot 0; slot j: + 64 j
  const int xrow = p.W * p.x_sp;                             // elements per image row (the per-lane slot offsets are recomputed per
                                                             // slot: ten of them in registers spilled the 256-VGPR budget)
  const half_t* xn = p.x + (long)n * p.x_sn;
  // weights of cout block cb: [cout tile 2][chunk][k-step 18][lane 64][8]; this wave's fragment column
  const half_t* wsrc0 = p.w + ((long)(cb * 2 + (w3 & 1)) * nchunks * 18 + (w3 >> 1)) * 512 + lane * 8;    // + (ch * 18 + 2j) * 512
  const unsigned wdst0 = W0_11 + ((w3 & 1) * 18 + (w3 >> 1)) * 1024;                                       // + 2j * 1024

  int pf_iy0 = 0, pf_ix0 = 0, pf_ch = 0;
  bool pf_interior = false;
  const half_t* pf_base = xn;
  const half_t* pf_w = wsrc0;
  auto issue_prep = [&](int S) {
    const int tile_i = S / nchunks, ch = S - tile_i * nchunks;
    pf_ch = ch;
    pf_w = wsrc0 + (long)ch * 18 * 512;
    if (tile_role) {
      const int tile = p.reverse ? e.ntiles - 1 - (first + tile_i * stride) : first + tile_i * stride;
      const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
      pf_iy0 = ty * TH11 - 1;
      pf_ix0 = tx * TW11 - 1;
      pf_interior = pf_iy0 >= 0 && pf_ix0 >= 0 && pf_iy0 + TIH11 <= p.H && pf_ix0 + TIW11 <= p.W;
      pf_base = xn + ((long)pf_iy0 * p.W + pf_ix0) * p.x_sp + ch * CK11;  // only dereferenced when interior
    }
  };
  auto issue_one = [&](int j, unsigned dst_buf) {            // j is a compile-time constant at every call site
    if constexpr (STAMP) { if (e.experiment == 2 || (e.experiment == 1 && !tile_role)) return; }
    if (tile_role) {
      const int u = j * 4 + w3;
      if (u < TPIECES11) {
        const int q = q0 + 64 * j;
        const int rr = q / TIW11, cc = q - rr * TIW11;
        const half_t* src = pf_base + (q < NPIX11 ? rr * xrow + cc * p.x_sp : 0) + csw * 8;
        if (!pf_interior) {              // uniform branch: border tiles clamp per lane
          const int iy = pf_iy0 + rr, ix = pf_ix0 + cc;
          const bool ok = q < NPIX11 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
          src = ok ? xn + ((long)iy * p.W + ix) * p.x_sp + pf_ch * CK11 + csw * 8 : e.zeros;
        }
        glds16_11(src, dst_buf + TILE0_11 + u * 1024);
      }
    } else if (j < 9) {
      glds16_11(pf_w + j * 1024, dst_buf + wdst0 + j * 2048);
    }
  };

  // ---- prologue: first stage's DMA, then the bias (a compiler-tracked load, younger than the DMA)
  issue_prep(0);
#pragma unroll
  for (int j = 0; j < DMA11; ++j) issue_one(j, lds0);
  if (tid < 64) bias_s[tid] = p.bias[cb * 64 + tid];

  // ---- B-fragment read offsets (tile buffer 0, s2 = 0): input row ir of this wave's row group, column r + dx, chunk slot hh
  int bq[NT11 + 2][3];
#pragma unroll
  for (int ir = 0; ir < NT11 + 2; ++ir)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int q = (rg * NT11 + ir) * TIW11 + r + dx;
      bq[ir][dx] = TILE0_11 + q * 64 + ((hh ^ ((q >> 2) & 3)) << 4);
    }
  const int aoff = W0_11 + mt * 18 * 1024 + lane * 16;       // + (tap * 2 + s2) * 1024

  __syncthreads();                       // bias visible (no DMA-aware wait here: see top of stage)
  f32x16 acc[NT11];
  bool stores_in_flight = false;         // the previous stage ended with exactly 8 epilogue stores (full tile)
  for (int S = 0; S < nstages; ++S) {
    const int tile_i = S / nchunks, ch = S - tile_i * nchunks;
    const unsigned bofs = (S & 1) ? BUFSTEP11 : 0;
    // This wave's DMA pieces of stage S have landed: they are older than the (at most 8) epilogue stores.
    ST11(0);
    if (stores_in_flight) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ST11(1);
    raw_barrier11();                     // every wave's pieces landed; every wave is done with the other buffer
    ST11(2);
    const bool have_next = S + 1 < nstages;
    if (have_next) issue_prep(S + 1);
    const unsigned nbuf = lds0 + (bofs ^ BUFSTEP11);
    const unsigned char* sb = smem + bofs;

    if (ch == 0) {                       // a new tile: the four accumulators start from the bias, re-read from LDS (a bias vector
      f32x16 b16;                        // kept in 16 registers for the whole launch spilled the 256-VGPR budget; a per-MFMA
#pragma unroll                           // select between bias and accumulator cost 190 v_cndmask per STAGE)
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + mt * 32 + 8 * g + hh * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) b16[4 * g + i] = b4[i];
      }
#pragma unroll
      for (int nt = 0; nt < NT11; ++nt) acc[nt] = b16;
    }
    // matrix phase: 6 groups (s2, dx) of 12 MFMAs; group g+1's 9 fragment reads are issued under group g's MFMAs
    half8 fa[2][3], fb[2][NT11 + 2];
    auto load_group = [&](int g, int buf) {
      const int s2 = g / 3, dx = g - 3 * s2;
      if constexpr (STAMP) {
        if (e.experiment == 3) {         // diagnostic: a single LDS read per group, the other fragments alias it
          const half8 v = *reinterpret_cast<const half8*>(sb + aoff + (dx * 2 + s2) * 1024);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) fa[buf][dy] = v;
#pragma unroll
          for (int ir = 0; ir < NT11 + 2; ++ir) fb[buf][ir] = v;
          return;
        }
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) fa[buf][dy] = *reinterpret_cast<const half8*>(sb + aoff + ((dy * 3 + dx) * 2 + s2) * 1024);
#pragma unroll
      for (int ir = 0; ir < NT11 + 2; ++ir) fb[buf][ir] = *reinterpret_cast<const half8*>(sb + (bq[ir][dx] ^ (s2 * 32)));
    };
    load_group(0, 0);
    __builtin_amdgcn_sched_barrier(0);   // group 0's own reads go out back to back
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      if (g + 1 < 6) load_group(g + 1, (g + 1) & 1);
#pragma unroll
      for (int ir = 0; ir < NT11 + 2; ++ir) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int nt = ir - dy;
          if (nt >= 0 && nt < NT11) {
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][dy], fb[g & 1][ir], acc[nt], 0, 0, 0);
          }
        }
      }
      if (g + 1 < 6) {                   // pin the software pipeline: the next group's 9 reads between this group's MFMAs
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      }
      if (have_next && g < 5) {          // 10 DMA slots of the next stage, two per group
        issue_one(2 * g, nbuf);
        issue_one(2 * g + 1, nbuf);
      }
    }
    ST11(3);
    stores_in_flight = false;
    if (ch != nchunks - 1) continue;

    // ---- epilogue: 32 couts x 4 rows of this wave, transposed through the finished tile buffer
    const int tile = p.reverse ? e.ntiles - 1 - (first + tile_i * stride) : first + tile_i * stride;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const bool full = (ty + 1) * TH11 <= p.Ho && (tx + 1) * TW11 <= p.Wo;
    raw_barrier11();                     // all waves finished reading this stage's buffers: the tile buffer becomes scratch
    ST11(4);
    {
      unsigned char* ew = smem + bofs + TILE0_11 + wave * (2 * EROW11);
      const int chunk = lane & 3, prow = lane >> 2;          // 8 channels (16 B) of pixel prow (+16 for the second pass)
      const int co = cb * 64 + mt * 32 + chunk * 8;
      // PixelShuffle(2) store (sub-pixel convs): the host packs the rows as (i*2+j)*cq + c, so a 32-row tile of this wave
      // lies inside one sub-pixel (cq % 32 == 0, eligibility): output pixel (2 oy + i, 2 ox + j), channels pc ..
      int pc = co, sub_y = 0, sub_x = 0, mul = 1, PW = p.Wo;
      if (p.out_mode == TDVC_OUT_SHUFFLE2) {
        const int cq = p.cout >> 2;
        const int sub = co / cq;
        pc = co - sub * cq;
        sub_y = sub >> 1; sub_x = sub & 1; mul = 2; PW = 2 * p.Wo;
      }
      const bool ch_ok = pc < p.y.C && co < ((p.cout + 63) & ~63);
      const int cc = ch_ok ? pc : 0;
      const int oy0 = ty * TH11 + rg * NT11, ox0 = tx * TW11;
      const bool has1 = p.res.p != nullptr, has2 = p.res2.p != nullptr;    // wave-uniform
      const half_t sl = (half_t)p.slope;
      const half2v sl2 = {sl, sl}, zero2 = {(half_t)0.f, (half_t)0.f};
      const bool act = p.slope != 1.f, relu = p.slope == 0.f;
      half_t* yb = reinterpret_cast<half_t*>(p.y.p) + (long)n * p.y.sn + cc;
      const half_t* rb1 = reinterpret_cast<const half_t*>(p.res.p) + (long)n * p.res.sn + cc;
      const half_t* rb2 = reinterpret_cast<const half_t*>(p.res2.p) + (long)n * p.res2.sn + cc;
#pragma unroll
      for (int j = 0; j < NT11; ++j) {
        int opix[2];
        bool ok[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int ox = ox0 + k * 16 + prow;
          ok[k] = full ? ch_ok : (ch_ok && oy0 + j < p.Ho && ox < p.Wo);
          opix[k] = ok[k] ? (mul * (oy0 + j) + sub_y) * PW + mul * ox + sub_x : 0;
        }
        half8 r1[2], r2[2];
        if (has1) {
#pragma unroll
          for (int k = 0; k < 2; ++k) r1[k] = *reinterpret_cast<const half8*>(rb1 + (long)opix[k] * p.res.sp);
        }
        if (has2) {
#pragma unroll
          for (int k = 0; k < 2; ++k) r2[k] = *reinterpret_cast<const half8*>(rb2 + (long)opix[k] * p.res2.sp);
        }
        unsigned char* er = ew + (j & 1) * EROW11;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          half2v lo = {(half_t)acc[j][4 * g + 0], (half_t)acc[j][4 * g + 1]};
          half2v hi = {(half_t)acc[j][4 * g + 2], (half_t)acc[j][4 * g + 3]};
          if (relu) {
            lo = __builtin_elementwise_max(lo, zero2);
            hi = __builtin_elementwise_max(hi, zero2);
          } else if (act) {
            lo = __builtin_elementwise_max(lo, lo * sl2);
            hi = __builtin_elementwise_max(hi, hi * sl2);
          }
          const half4 o = {lo[0], lo[1], hi[0], hi[1]};
          *reinterpret_cast<half4*>(er + r * EPS11 + (8 * g + 4 * hh) * 2) = o;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          half8 v = *reinterpret_cast<const half8*>(er + (k * 16 + prow) * EPS11 + chunk * 16);
          if (has1) v = v + r1[k];
          if (has2) v = v + r2[k];
          if (ok[k]) *reinterpret_cast<half8*>(yb + (long)opix[k] * p.y.sp) = v;
        }
      }
    }
    stores_in_flight = full;
    ST11(5);
    if constexpr (STAMP) {
      if (S == 2 * nchunks - 1 && lane == 0) {   // one record per wave: [block][wave][16 stamps]
        const int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (bid * 8 + 7 < stamp_cap) for (int i = 0; i < 16; ++i) stamps[((long)bid * 8 + wave) * 16 + i] = stv[i];
      }
    }
  }
}

}  // namespace

extern "C" void tdvc_debug_set_stamp_buffer_v11(void* buf, int cap_blocks) { g_stamp11 = (long long*)buf; g_stamp11_cap = cap_blocks; }
extern "C" void tdvc_debug_set_v11_experiment(int mode) { g_v11_experiment = mode; }

static bool g_v11_enabled = true;
// tests and A/B benchmarks switch the kernel off to send the same layers to conv_mfma_v3
extern "C" void tdvc_debug_enable_conv_v11(int enable) { g_v11_enabled = enable != 0; }

bool conv_v11_eligible(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo) {
  static const bool off = getenv("TDVC_CONV_NO_V11") != nullptr || getenv("TDVC_CONV_V1") != nullptr;
  static const int min_cin = getenv("TDVC_V11_MIN_CIN") ? atoi(getenv("TDVC_V11_MIN_CIN")) : 128;
  if (off || !g_v11_enabled) return false;
  bool taps33 = d->ntaps == 9 && d->kh == 3 && d->kw == 3 && d->pad == 1;
  for (int t = 0; taps33 && t < 9; ++t) taps33 = d->tap_dy[t] == t / 3 && d->tap_dx[t] == t % 3;
  // The counted `vmcnt(8)` at the top of a stage assumes that EVERY wave issued its 8 epilogue stores: a narrower output view
  // (y.C at or below cout - 32, or cout / 4 - 32 for the sub-pixel store) would leave whole waves without stores and their
  // weight DMA pieces in flight across the barrier -- such launches go to the stage-pipelined kernel instead.
  // A wave covers 32 consecutive packed rows (cout % 64 == 0 here; sub-pixel store: inside one sub-pixel, cq % 32 == 0): it
  // stores as soon as its first 8-channel chunk is inside the view.
  const bool all_waves_store = d->y.C > (p.out_mode == TDVC_OUT_SHUFFLE2 ? (d->cout >> 2) : d->cout) - 32;
  return taps33 && all_waves_store && d->ck == 32 && d->stride == 1 && d->cout >= 64 && (d->cout % 64) == 0 && (d->x.C % 32) == 0 && d->x.C >= min_cin && !d->s2d &&
         !d->square_input && (long)Ho * Wo >= 8192 &&
         (convk::conv_is_lean(p) || (convk::conv_is_simple(p) && !p.gdn && p.out_mode == TDVC_OUT_SHUFFLE2 && ((d->cout >> 2) % 32) == 0));
}

int launch_conv_v11(const ConvParams& p, int cout_blocks, int N, hipStream_t st) {
  const void* zeros = nullptr;
  if (const int zrc = tdvc_scratch_pages(&zeros, nullptr)) return zrc;
  ConvParams q = p;
  q.tiles_x = (p.Wo + TW11 - 1) / TW11;
  const int tiles_y = (p.Ho + TH11 - 1) / TH11;
  V11Extra e;
  e.ntiles = q.tiles_x * tiles_y;
  e.zeros = reinterpret_cast<const half_t*>(zeros);
  e.experiment = g_v11_experiment;
  q.slope = convk::conv_simple_slope(p);
  int gx = 256 / (cout_blocks * N);
  if (gx < 1) gx = 1;
  if (gx > e.ntiles) gx = e.ntiles;
  dim3 grid(gx, cout_blocks, N);
  static TdvcPerDeviceFlag attr_flags;
  bool& attr_done = attr_flags.flag();
  if (!attr_done) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v11_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err == hipSuccess)
      err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v11_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) { tdvc_set_error("conv v11: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    attr_done = true;
  }
  if (g_stamp11) hipLaunchKernelGGL(conv_mfma_v11_kernel<true>, grid, dim3(NTHR11), LDS11, st, q, e, g_stamp11, g_stamp11_cap);
  else hipLaunchKernelGGL(conv_mfma_v11_kernel<false>, grid, dim3(NTHR11), LDS11, st, q, e, (long long*)nullptr, 0);
  return tdvc_launch_status("tdvc_conv2d(v11)");
}
