// conv_row — 3x3 / stride 1 / pad 1 convs with Cin = 128 (the 128 -> 128 / 64 / 256 / 512 layers of the two coders, MCNet and the
// in-loop filter: 52 launches, 5.5 ms of a 1080p frame on conv_mfma_v11 at 0.37-0.41 of the MFMA peak), lean epilogue or
// PixelShuffle(2) store, up to two fp16 residuals.
//
// What held v11 back (profiles/r03_v11_stamps.txt): its weights travel through LDS with every tile (36 KB per 32-channel stage),
// two waves of a SIMD fight for the issue port in a 7.9 k-cycle matrix phase that holds 4.6 k cycles of MFMA, and a 6-7 k-cycle
// tile epilogue overlaps nothing.  This kernel is conv_pair's structure (conv_pair.hip) for ONE conv with twice the K:
//
//   * weights in REGISTERS.  8 waves, two per SIMD; wave w owns output channels [16 w, +16) of the workgroup's block of 128 and
//     keeps their 9 taps x 4 chunks of 32 input channels as 36 v_mfma_f32_16x16x32_f16 A fragments (144 VGPRs), gathered once per
//     launch from the layer's standard packed blob (conv_backward.hip: pack_indexed_item, ck = 32).  LDS holds activations only.
//   * row streaming.  A workgroup walks a strip of 32 output columns top to bottom.  Per step one input row (34 px x 256 B, by
//     LDS-DMA into an 8-row ring, PF rows ahead) is consumed: its B fragments (16 px x 32 channels) feed the three live output
//     rows (dy = 0, 1, 2: rotating accumulators); the finished row goes to an LDS staging ring as fp16 (bias from the
//     accumulator chain's first MFMA, activation as packed fp16 math) and BI steps later leaves as full 256-byte lines with the
//     residual rows added.  No tile epilogue, no halo recompute, 4 extra steps per strip segment.
//   * roles.  Every wave computes (72 MFMAs per step).  Waves 0-3 also issue the LDS-DMA (9 pieces of 1 KB per row: pieces w,
//     w + 4 and, in turn, piece 8), with an exactly counted `s_waitcnt vmcnt`; waves 4-7 also store (two 16-byte items per thread
//     and step) and pre-load the residual rows.  The split is what keeps the counted wait exact: the compiler's own waits for a
//     residual load know nothing of inline-asm DMA and would drain a loader wave's row prefetch at every use.
//   * one s_barrier per BI = 2 steps; per step and SIMD 144 MFMAs = 2 304 cycles against ~600 of barrier / first-fragment latency
//     / pack (conv_pair: 1 152 against the same overheads).
//
// LDS image of a ring row: pixel-major, 256 B per pixel, the 16-byte slot c of pixel q stored at position c ^ F(q), F(q) = q & 12 |
// swap of bits 0 and 1 of q.  A B fragment is read by ds_read_b128 with lane (r, kb) = pixel p0 + SIGMA(r), slot 4 kc + kb; the
// instruction is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, ...), i.e. eight pixels with slot s and the other eight
// with slot s ^ 1.  With SIGMA sending lanes {0-3, 12-15} to the even and lanes {4-11} to the odd pixels of the 16-pixel block the
// two sets are the parity classes of the window, F pairs q with q ^ 2 (same parity), and every group reads 16 different 16-byte
// bank slots for all three windows dx = 0, 1, 2 (exhaustive check: tools/check_row_swizzle.py).  For the DMA-written rows the XOR
// sits on the SOURCE address.  The accumulators' columns follow SIGMA, so the pack writes pixel 16 cb + SIGMA(r).
#include <type_traits>

#include "conv_common.h"

using convk::ConvParams;

namespace {

// Two geometries, one code (template parameter NKC = 32-channel chunks of the input):
//   NKC = 4 (Cin = 128): strips of 32 columns, 8 waves = 8 blocks of 16 output channels (128 per workgroup pass), 72 MFMAs per wave and step;
//   NKC = 2 (Cin = 64):  strips of 64 columns, 8 waves = 4 blocks of 16 output channels (64 per pass) x 2 halves of the strip, 36 MFMAs per
//                        wave and step, 128-byte ring pixels (swizzle term (q >> 1) & 7: the eight same-parity pixels of a window sit in
//                        one half of the 256-byte bank row and take eight different positions).
// Both: a wave computes two 16-pixel column blocks, a ring row is 9 DMA pieces of 1 KB, a staging row 8 KB = 512 items of 16 bytes.
template <int NKC> struct RowGeo {
  static constexpr int SW = NKC == 4 ? 32 : 64;          // output columns per strip
  static constexpr int NCG = NKC == 4 ? 8 : 4;           // blocks of 16 output channels per workgroup pass
  static constexpr int CO = 16 * NCG;                    // output channels per pass
  static constexpr int PXB = NKC * 64;                   // bytes per ring pixel
  static constexpr int LPP = PXB / 16;                   // 16-byte positions (= DMA lanes) per ring pixel
  static constexpr int SLOTS = CO / 8;                   // 16-byte positions per staging pixel
};
constexpr int RW_PIECES = 9;                    // 1-KB DMA pieces per ring row (34 of 36 / 66 of 72 pixel slots used)
constexpr int RW_ROWB = RW_PIECES * 1024;
constexpr int RW_SROW = 8192;                   // staging row bytes: SW x CO x 2
constexpr int RW_MAXCO = 128;
static_assert(RowGeo<4>::SW * RowGeo<4>::CO * 2 == RW_SROW && RowGeo<2>::SW * RowGeo<2>::CO * 2 == RW_SROW, "staging row");
static_assert((RowGeo<4>::SW + 2) * RowGeo<4>::PXB <= RW_ROWB && (RowGeo<2>::SW + 2) * RowGeo<2>::PXB <= RW_ROWB, "ring row");
// ---- synchronisation geometry (step k consumes input row k; a workgroup barrier closes every BI-th step; between two barriers
// two waves are at most BI - 1 steps apart and always in the same barrier interval).
constexpr int RW_BI = 2;
constexpr int RW_XRING = 8;                     // input ring rows
constexpr int RW_PF = 6;                        // the DMA of row k + PF is issued in step k
constexpr int RW_HD = 4;                        // land_wait() after step s lets the DMA issued in steps s - HD + 1 .. s stay in flight
constexpr int RW_SRING = 2 * RW_BI;             // staging rows: written in step k, stored in step k + BI
constexpr int RW_X0 = 0, RW_S0 = RW_XRING * RW_ROWB;
constexpr int RW_B0 = RW_S0 + RW_SRING * RW_SROW;           // 128 bias floats of the workgroup's cout block
constexpr int RW_LDS = RW_B0 + RW_MAXCO * 4;
// (1) input ring, write-after-read: the DMA issued in step k by the fastest loader overwrites row k + PF - XRING; the slowest wave
//     of the interval is at step >= k - (BI - 1) and reads exactly that step's row: k + PF - XRING < k - (BI - 1).
static_assert(RW_PF - RW_XRING < -(RW_BI - 1), "input ring: a DMA would land on a row a wave BI - 1 steps behind still reads (needs PF + BI <= XRING)");
// (2) landing: a loader's own pieces of row r (issued in step r - PF) are known to have landed after its land_wait() of a step
//     s >= r - PF + HD, the other waves learn it at the next barrier; the last barrier before step r closes a step >= r - BI.
static_assert(RW_PF >= RW_HD + RW_BI, "landing: row r must be past land_wait()'s history window before the last barrier in front of step r");
// (3) staging ring: row written in step k, stored in step k + BI (the next barrier interval), overwritten in step k + SRING.
static_assert(RW_SRING >= 2 * RW_BI && (RW_SRING & (RW_SRING - 1)) == 0 && (RW_XRING & (RW_XRING - 1)) == 0 && (RW_BI & (RW_BI - 1)) == 0, "rings");
static_assert(RW_LDS <= 160 * 1024, "LDS");
static_assert(RW_HD == 4, "land_wait(): four-step history");
constexpr int RW_NTHR = 512;
// compile-time A/B switches (tools/build_row_variants.sh): where a loader wave issues a row's DMA inside a step (0 = before its
// MFMAs, 1 = after the sixth of its twelve fragment groups, 2 = after the last), lean address arithmetic / constant counted wait
#ifndef RW_DMA_POS
#define RW_DMA_POS (NKC == 4 ? 1 : 2)      /* measured (tools/ab_row.py, r04): Cin = 128: 133 / 141 / 144 us for mid / top / end at 128 -> 128 @544x960; Cin = 64: 169 / 177 / 158 us at 64 -> 64 @1088x1920 */
#endif
#ifndef RW_LEAN
#define RW_LEAN 1
#endif

struct RowParams {
  const half_t* x; long x_sn; int x_sp;
  half_t* y; long y_sn; int y_sp;
  const half_t* res; long r1_sn; int r1_sp;
  const half_t* w;          // the layer's standard blob: [cout tile 32][chunk NKC][k-step 18][lane 64][8 halves]
  const float* bias;
  const half_t* zeros;      // >= 16 B of zeros: DMA source of out-of-image pixels, load source of masked residual items
  half_t* dump;             // >= 8 KB nobody reads: store target of items outside the image
  int N, H, W;
  int ncb;                  // cout blocks of CO channels
  int cq;                   // PixelShuffle(2) store: channels after the shuffle (cout / 4); 0 = plain NHWC
  int strips;               // SW-column strips per image row
  float slope;              // max(v, v * slope): 1 = none, 0 = ReLU
  int reverse;
};

template <int NKC> __device__ __forceinline__ int rw_f(int q) {                                          // ring-row swizzle term
  return NKC == 4 ? ((q & 12) | ((q & 1) << 1) | ((q >> 1) & 1)) : ((q >> 1) & 7);
}
__device__ __forceinline__ int rw_sigma(int r) { return r < 4 ? 2 * r : (r >= 12 ? 2 * r - 16 : 2 * r - 7); }   // fragment lane -> pixel of its 16-px block
__device__ __forceinline__ int rw_g(int q) { return (q >> 1) & 7; }                                      // staging-row swizzle term

__device__ __forceinline__ void rw_glds16(const half_t* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void rw_barrier(bool skip = false) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (!skip) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ACT: 0 none, 1 ReLU, 2 max(v, v * slope); NRES: 0 / 1 fp16 residual added in the store phase (no Cin = 128 layer of the model
// has two; such a launch stays on conv_mfma_v11); SHUF: PixelShuffle(2) store
template <int NKC, int ACT, int NRES, bool SHUF>
__global__ __launch_bounds__(RW_NTHR, 1) void conv_row_kernel(const RowParams p) {
  using G = RowGeo<NKC>;
  constexpr int RW_NKC = NKC, RW_SW = G::SW, RW_CO = G::CO, RW_PXB = G::PXB;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave < 4;                // wave-uniform roles: waves 0-3 issue the DMA, waves 4-7 store
  const int wq = wave & 3;
  const int cg = wave % G::NCG, cbp = wave / G::NCG;        // block of 16 output channels; pair of column blocks (2 cbp, 2 cbp + 1) of the strip
  const int r16 = lane & 15, kb = lane >> 4;

  // ---- workgroup -> (cout block, run of rows).  The launch's work is the flat sequence of (image, strip, row) = N * strips * H
  // rows of 32 output columns; a cout block's S workgroup slots cut it into S equal runs (a run that crosses a strip end is two
  // segments), so every workgroup gets the same number of rows whatever H and the strip count are (a segment pays 2 + BI extra
  // steps).  Workgroup b sits on XCD b % 8: an XCD's slots take consecutive runs (neighbouring strips share their column halo in
  // that XCD's L2); the cout block is fixed per workgroup (its weights are loaded once)
  const int nwg = (int)gridDim.x, b = (int)blockIdx.x;
  int cbk, pos_slot, nslots;
  if ((nwg & 7) == 0 && ((nwg >> 3) % p.ncb) == 0) {
    const int slot = b >> 3, per_xcd = (nwg >> 3) / p.ncb;
    cbk = slot % p.ncb;
    pos_slot = (b & 7) * per_xcd + slot / p.ncb;
    nslots = 8 * per_xcd;
  } else {
    cbk = b % p.ncb; pos_slot = b / p.ncb; nslots = nwg / p.ncb;
  }
  if (p.reverse) pos_slot = nslots - 1 - pos_slot;
  const long allrows = (long)p.N * p.strips * p.H;
  const long run_begin = allrows * pos_slot / nslots, run_end = allrows * (pos_slot + 1) / nslots;

  // ---- this wave's 16 output channels: 36 A fragments (tap, chunk) in registers for the whole launch.  Lane (r16, kb) of
  // fragment (tap, kc) holds w[cout][cin = 32 kc + 8 kb .. + 7][tap]: in the standard blob that is k-step tap * 2 + (kb >> 1),
  // half (kb & 1), row cout & 31 of cout tile cout >> 5
  half8 wf[9 * RW_NKC];
  {
    const int co = cbk * RW_CO + 16 * cg + r16;
    const half_t* wb = p.w + ((long)(co >> 5) * RW_NKC * 18 * 64 + (kb & 1) * 32 + (co & 31)) * 8 + (long)(kb >> 1) * 512;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kc = 0; kc < RW_NKC; ++kc) wf[tap * RW_NKC + kc] = *reinterpret_cast<const half8*>(wb + ((long)kc * 18 + tap * 2) * 512);
  }
  // the bias is the C operand of every accumulator chain's first MFMA: re-read from LDS per step (4 registers of a 256-register budget)
  if (tid < RW_CO) reinterpret_cast<float*>(smem + RW_B0)[tid] = p.bias[cbk * RW_CO + tid];
  const int boff = RW_B0 + (16 * cg + 4 * kb) * 4;     // C/D rows 4 kb + i of this wave's 16 channels

  // ---- per-lane LDS offsets
  int foff[3];                                 // B fragment (dx), chunk 0, column block 0; chunk kc: ^ (kc << 6); block 1: + 4096
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int q = 32 * cbp + dx + rw_sigma(r16), f = rw_f<NKC>(q);
    foff[dx] = q * RW_PXB + ((kb ^ f) << 4);            // (4 kc + kb) ^ f = (kb ^ f) ^ (kc << 2)
  }
  int doff[2];                                 // pack: pixel 16 cb + SIGMA(r16), channels 16 wave + 4 kb .. + 3 of the block
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int q = 32 * cbp + 16 * cb + rw_sigma(r16), c = 2 * cg + (kb >> 1);
    doff[cb] = q * (RW_CO * 2) + ((c ^ rw_g(q)) << 4) + 8 * (kb & 1);
  }
  // ---- everything below is instantiated once per ROLE (waves 0-3: loaders, waves 4-7: storers) inside one wave-uniform branch, so that a
  // role's private registers (DMA offsets and masks / output offsets and the prefetched residual row) share physical registers with
  // the other role's instead of adding up: the kernel lives within 256 registers with 144 of them holding weights
  auto run = [&](auto LOADERc) __attribute__((always_inline)) {
  constexpr bool LOADER = decltype(LOADERc)::value;
  // DMA items (loader waves): piece j of a row covers 1 KB of ring pixels (lane: position lane % LPP of pixel (64 / LPP) j + lane / LPP);
  // ring pixel q is image column c0 - 1 + q
  int soff[3], sq[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int piece = j == 2 ? 8 : wq + 4 * j;
    const int q = (64 / G::LPP) * piece + lane / G::LPP, c = (lane % G::LPP) ^ rw_f<NKC>(q);
    sq[j] = q;
    soff[j] = q * p.x_sp + c * 8;
  }
  // store items (storer waves): item it = (tid - 256) + 256 j: pixel it / SLOTS, position it % SLOTS of the staging row
  const int s_t = tid & 255;
  int s_px[2], s_co[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int it = s_t + 256 * j;
    s_px[j] = it / G::SLOTS;
    s_co[j] = cbk * RW_CO + (((it % G::SLOTS) ^ rw_g(it / G::SLOTS)) << 3);
  }
  int s_pc[2], s_sub[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if constexpr (SHUF) { s_sub[j] = s_co[j] / p.cq; s_pc[j] = s_co[j] - s_sub[j] * p.cq; }
    else { s_sub[j] = 0; s_pc[j] = s_co[j]; }
  }
  const int OW = SHUF ? 2 * p.W : p.W;         // output row pitch in pixels

  const half_t hs = (half_t)p.slope;
  const half4 sl4 = {hs, hs, hs, hs};
  f32x4 acc[3][2];
  half8 r1v[2] = {};

  for (long pos = run_begin; pos < run_end;) {
    const int sid = (int)(pos / p.H);          // (image, strip)
    const int ra = (int)(pos - (long)sid * p.H), rb = (int)min((long)p.H, ra + (run_end - pos));
    const int rows = rb - ra;
    pos += rows;
    const int n = sid / p.strips, strip = sid - n * p.strips;
    const int c0 = strip * RW_SW;
    const half_t* xn = p.x + (long)n * p.x_sn;

    bool colok[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) colok[j] = sq[j] < RW_SW + 2 && c0 - 1 + sq[j] >= 0 && c0 - 1 + sq[j] < p.W;
    // this thread's two output items: element offset of (row 0, its column, its channels) and validity
    int s_off[2];
    bool s_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = c0 + s_px[j];
      s_ok[j] = col < p.W;
      s_off[j] = SHUF ? (s_sub[j] >> 1) * OW + 2 * col + (s_sub[j] & 1) : col;       // pixel indices fit 32 bits (checked by the host)
    }
#if RW_LEAN
    // lean form: a row's base address is wave-uniform (scalar registers); a lane adds a 32-bit element offset.  Lanes outside the image
    // load their residual from the last valid column (the value goes to the dump page): no address selects on the loads.
    int y_lo[2], r_lo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int colc = min(c0 + s_px[j], p.W - 1);
      const int offc = SHUF ? (s_sub[j] >> 1) * OW + 2 * colc + (s_sub[j] & 1) : colc;
      y_lo[j] = s_off[j] * p.y_sp + s_pc[j];
      r_lo[j] = offc * p.r1_sp + s_pc[j];
    }
    const half_t* const yimg = p.y + (long)n * p.y_sn;
    const half_t* const rimg = NRES >= 1 ? p.res + (long)n * p.r1_sn : nullptr;
    const long yrow_stride = (long)(SHUF ? 2 : 1) * OW * p.y_sp, rrow_stride = (long)(SHUF ? 2 : 1) * OW * p.r1_sp;
    // loader: the row the next DMA reads, advanced by one image row per step
    const long xrow_stride = (long)p.W * p.x_sp;
#endif
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[s3][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // x row ra - 1 + kk into ring slot kk & 7: pieces wq, wq + 4 and, when it is this wave's turn, piece 8
    auto issue_row = [&](int kk) __attribute__((always_inline)) -> int {
      const int row = ra - 1 + kk;
      const bool rowok = row >= 0 && row < p.H;
#if RW_LEAN
      const half_t* base = xn + (long)(c0 - 1) * p.x_sp + row * xrow_stride;
#else
      const half_t* base = xn + ((long)row * p.W + (c0 - 1)) * p.x_sp;
#endif
      const unsigned dst = lds0 + RW_X0 + (kk & (RW_XRING - 1)) * RW_ROWB;
      rw_glds16((rowok && colok[0]) ? base + soff[0] : p.zeros, dst + wq * 1024);
      rw_glds16((rowok && colok[1]) ? base + soff[1] : p.zeros, dst + (wq + 4) * 1024);
      if (wq == (kk & 3)) {
        rw_glds16((rowok && colok[2]) ? base + soff[2] : p.zeros, dst + 8 * 1024);
        return 3;
      }
      return 2;
    };

    // ---- prologue: the first PF rows; everything landed and visible before step 0
    if constexpr (LOADER) {
#pragma unroll
      for (int kk = 0; kk < RW_PF; ++kk)
        if (kk <= rows + 1) issue_row(kk);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rw_barrier();

    const int K = (rows + 2 + RW_BI + RW_BI - 1) & ~(RW_BI - 1);     // steps, a whole number of barrier intervals
    unsigned hist = 0;                         // DMA instructions this wave issued in each of the last four steps (one byte each)
    auto land_wait = [&](int nvm) __attribute__((always_inline)) {
#if RW_LEAN
      // every loader issues 2 + 2 + 2 + 3 = 9 instructions over any four consecutive issuing steps (piece 8 goes round the four
      // waves), so the count left in flight is the constant 9 while rows are being issued; shorter histories (the first steps of
      // a segment) have fewer outstanding than that and pass without waiting; once issuing has stopped the wait is vmcnt(0)
      if (nvm) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
#endif
      hist = (hist << 8) | (unsigned)nvm;
      const unsigned sum = (hist & 0xFFu) + ((hist >> 8) & 0xFFu) + ((hist >> 16) & 0xFFu) + (hist >> 24);
      switch (sum) {
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    };

    // One step k: every wave consumes x row i = ra - 1 + k: output rows i + 1 (started), i, i - 1 (finished -> staging row k);
    // storer waves first send the row finished BI steps ago (staging row k - BI, image row ra + k - BI - 2) to memory.
    auto step = [&](auto PHc, auto FULLc, int k) __attribute__((always_inline)) {
      constexpr int PH = decltype(PHc)::value;       // k % 3: accumulator row started here = (PH + 1) % 3, continued = PH, finished = (PH + 2) % 3
      constexpr bool FULL = decltype(FULLc)::value;
      constexpr int SN = (PH + 1) % 3, SM = PH, SD = (PH + 2) % 3;
      if constexpr (!LOADER) {
        const int srow = ra + k - RW_BI - 2;
        if (FULL || (srow >= ra && srow < rb)) {
          const unsigned char* sb = smem + RW_S0 + ((k - RW_BI) & (RW_SRING - 1)) * RW_SROW;
#if RW_LEAN
          half_t* const yrow = const_cast<half_t*>(yimg) + srow * yrow_stride;      // wave-uniform
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            half8 yv = *reinterpret_cast<const half8*>(sb + (s_t + 256 * j) * 16);
            if constexpr (NRES >= 1) yv = yv + r1v[j];
            // no branch around the store: behind a branch hipcc drains vmcnt(0) at the join, i.e. in front of the residual loads below
            // (measured: 128 -> 128 + residual @544x960 145 -> 175 us); lanes outside the image store into the dump page instead
            *reinterpret_cast<half8*>(s_ok[j] ? yrow + y_lo[j] : p.dump + (s_t + 256 * j) * 8) = yv;
          }
#else
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            half8 yv = *reinterpret_cast<const half8*>(sb + (s_t + 256 * j) * 16);
            if constexpr (NRES >= 1) yv = yv + r1v[j];
            const long opix = (SHUF ? 2 * srow : srow) * OW + s_off[j];
            half_t* dst = s_ok[j] ? p.y + (long)n * p.y_sn + opix * p.y_sp + s_pc[j] : p.dump + (s_t + 256 * j) * 8;
            *reinterpret_cast<half8*>(dst) = yv;
          }
#endif
        }
        if constexpr (NRES >= 1) {
          const int nrow = srow + 1;
          const bool rok = FULL || (nrow >= ra && nrow < rb);
#if RW_LEAN
          const half_t* const rrow = rimg + (rok ? nrow : ra) * rrow_stride;        // wave-uniform; a row outside the segment reads a valid one
#pragma unroll
          for (int j = 0; j < 2; ++j) r1v[j] = *reinterpret_cast<const half8*>(rrow + r_lo[j]);
#else
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const long opix = (SHUF ? 2 * nrow : nrow) * OW + s_off[j];
            const bool ok = rok && s_ok[j];
            r1v[j] = *reinterpret_cast<const half8*>(ok ? p.res + (long)n * p.r1_sn + opix * p.r1_sp + s_pc[j] : p.zeros);
          }
#endif
        }
      }
      int nvm = 0;
      if (FULL || k <= rows + 1) {
        const unsigned char* xb = smem + RW_X0 + (k & (RW_XRING - 1)) * RW_ROWB;
        // 12 groups g = (kc, dx) of 6 MFMAs (3 output rows x 2 column blocks) on two B fragments; the fragments of group g + 2
        // are requested before group g's MFMAs (three fragment pairs in flight: 24 registers)
        constexpr int FBN = (NKC == 4 && NRES == 1) ? 2 : 3;      // fragment pairs in flight (the residual variants of the 128-channel
        half8 fb[FBN][2];                                           // geometry sit at the 256-register limit: one pair less)
        auto load_group = [&](int g, int buf) __attribute__((always_inline)) {
          const int kc = g / 3, dx = g - 3 * kc;
          const unsigned char* b0 = xb + (foff[dx] ^ (kc << 6));
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) fb[buf][cb] = *reinterpret_cast<const half8*>(b0 + cb * 16 * RW_PXB);
        };
        const f32x4 bias4 = *reinterpret_cast<const f32x4*>(smem + boff);
        load_group(0, 0);
        if (FBN == 3) load_group(1, 1);
        if (RW_DMA_POS == 0 && LOADER && k + RW_PF <= rows + 1) nvm = issue_row(k + RW_PF);
#pragma unroll
        for (int g = 0; g < 3 * RW_NKC; ++g) {
          const int kc = g / 3, dx = g - 3 * kc;
          if (g + FBN - 1 < 3 * RW_NKC) load_group(g + FBN - 1, (g + FBN - 1) % FBN);
          // the loader's DMA in the MIDDLE of its MFMA stream: its ~100 instructions of address work and the pieces' issue time run
          // while the matrix pipe works off this wave's queued MFMAs (at the top of the step both waves of a SIMD did their role
          // work side by side in front of an idle pipe)
          if (RW_DMA_POS == 1 && g == 3 * RW_NKC / 2 && LOADER && k + RW_PF <= rows + 1) nvm = issue_row(k + RW_PF);
#pragma unroll
          for (int dyo = 0; dyo < 3; ++dyo) {
            const int dy = 2 - dyo;            // the finishing row first: its chain ends four MFMAs before the step's last one
            const int slot = dy == 2 ? SD : (dy == 1 ? SM : SN);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
              const bool first = dy == 0 && g == 0;
              acc[slot][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[(dy * 3 + dx) * RW_NKC + kc], fb[g % FBN][cb],
                                                                     first ? bias4 : acc[slot][cb], 0, 0, 0);
            }
          }
        }
        if (RW_DMA_POS == 2 && LOADER && k + RW_PF <= rows + 1) nvm = issue_row(k + RW_PF);
        {
          // finished row i - 1 -> fp16, activation, into staging row k (rows outside the segment are never stored)
          unsigned char* sb = smem + RW_S0 + (k & (RW_SRING - 1)) * RW_SROW;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            const f32x4 v = acc[SD][cb];
            half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            if constexpr (ACT == 1) {
              const half4 z = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
              h = __builtin_elementwise_max(h, z);
            } else if constexpr (ACT == 2) {
              h = __builtin_elementwise_max(h, h * sl4);
            }
            *reinterpret_cast<half4*>(sb + doff[cb]) = h;
          }
        }
      }
      if constexpr (LOADER) land_wait(nvm);
      rw_barrier((k & (RW_BI - 1)) != RW_BI - 1);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    auto edge_steps = [&](int k, int kend) __attribute__((always_inline)) {   // k % 3 == 0
      for (; k < kend; k += 3) {
        step(I0{}, std::false_type{}, k);
        if (k + 1 < kend) step(I1{}, std::false_type{}, k + 1);
        if (k + 2 < kend) step(I2{}, std::false_type{}, k + 2);
      }
    };
    constexpr int KF = ((RW_BI + 2 + 2) / 3) * 3;       // first steady-state step: a multiple of 3 >= BI + 2
    edge_steps(0, KF);
    int k = KF;
    for (; k + 2 <= rows + 1; k += 3) {
      step(I0{}, std::true_type{}, k);
      step(I1{}, std::true_type{}, k + 1);
      step(I2{}, std::true_type{}, k + 2);
    }
    edge_steps(k, K);
    rw_barrier();                              // the next job's prologue overwrites ring rows the slowest wave may still read
  }
  };
  if (loader) run(std::true_type{});
  else run(std::false_type{});
}

}  // namespace

static int g_row_enabled = 3;
// tests and A/B benchmarks: bit 0 = the Cin = 128 layers (off: conv_mfma_v11 takes them), bit 1 = the Cin = 64 layers (off: conv_mfma_v10)
extern "C" void tdvc_debug_enable_conv_row(int enable) { g_row_enabled = enable; }

bool conv_row_eligible(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo) {
  static const bool off = getenv("TDVC_CONV_NO_ROW") != nullptr || getenv("TDVC_CONV_V1") != nullptr;
  static const bool off64 = getenv("TDVC_CONV_NO_ROW64") != nullptr;
  if (off || !g_row_enabled) return false;
  bool taps33 = d->ntaps == 9 && d->kh == 3 && d->kw == 3 && d->pad == 1;
  for (int t = 0; taps33 && t < 9; ++t) taps33 = d->tap_dy[t] == t / 3 && d->tap_dx[t] == t % 3;
  const bool shuf = p.out_mode == TDVC_OUT_SHUFFLE2;
  const int ych = shuf ? (d->cout >> 2) : d->cout;       // channels of an output pixel
  const bool c128 = (g_row_enabled & 1) && d->x.C == 128 && (d->cout % RowGeo<4>::CO) == 0 &&
                    (convk::conv_is_lean(p) || (convk::conv_is_simple(p) && !p.gdn && shuf && (ych % 8) == 0));
  const bool c64 = d->x.C == 64 && (d->cout % RowGeo<2>::CO) == 0 && convk::conv_is_lean(p) && !off64 && (g_row_enabled & 2);
  return taps33 && (c128 || c64) && d->ck == 32 && d->stride == 1 && !d->s2d && !d->square_input && d->bias &&
         d->y.C >= ych && (!d->res.p || d->res.C >= ych) && (!d->res2.p || d->res2.C >= ych) && !(d->res.p && d->res2.p) &&
         (long)Ho * Wo >= 8192 && Ho >= 16;
}

template <int NKC>
static int launch_conv_row_t(const ConvParams& p, int N, hipStream_t st) {
  using G = RowGeo<NKC>;
  const void* zeros = nullptr;
  void* dump = nullptr;
  if (const int zrc = tdvc_scratch_pages(&zeros, &dump)) return zrc;
  RowParams q;
  q.x = p.x; q.x_sn = p.x_sn; q.x_sp = p.x_sp;
  q.y = reinterpret_cast<half_t*>(p.y.p); q.y_sn = p.y.sn; q.y_sp = p.y.sp;
  q.res = reinterpret_cast<const half_t*>(p.res.p); q.r1_sn = p.res.p ? p.res.sn : 0; q.r1_sp = p.res.p ? p.res.sp : 0;
  q.w = p.w; q.bias = p.bias;
  q.zeros = reinterpret_cast<const half_t*>(zeros);
  q.dump = reinterpret_cast<half_t*>(dump);
  q.N = N; q.H = p.H; q.W = p.W;
  q.ncb = p.cout / G::CO;
  const bool shuf = p.out_mode == TDVC_OUT_SHUFFLE2;
  q.cq = shuf ? p.cout >> 2 : 0;
  q.slope = convk::conv_simple_slope(p);
  q.reverse = p.reverse;
  q.strips = (p.W + G::SW - 1) / G::SW;
  // 256 workgroups (one per CU), 256 / ncb slots per cout block; fewer when the launch has fewer rows than slots
  const long allrows = (long)N * q.strips * p.H;
  int per = 256 / q.ncb;
  if (per < 1) per = 1;
  if (allrows < per) per = (int)allrows;
  const int grid = per * q.ncb;
  if (!p.res.p && p.res2.p) {                  // a single residual always travels as `res`
    q.res = reinterpret_cast<const half_t*>(p.res2.p); q.r1_sn = p.res2.sn; q.r1_sp = p.res2.sp;
  }
  const bool has1 = q.res != nullptr;
  const int act = q.slope == 1.f ? 0 : (q.slope == 0.f ? 1 : 2);
  auto go = [&](auto kern) -> int {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RW_LDS);
    if (err != hipSuccess) { tdvc_set_error("conv_row: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(RW_NTHR), RW_LDS, st, q);
    return 0;
  };
  const int nres = has1 ? 1 : 0;
  int rc = TDVC_EINVAL;
#define RW_CASE(A, R, S) if (act == A && nres == R && shuf == S) rc = go(&conv_row_kernel<NKC, A, R, S>);
  RW_CASE(0, 0, false) RW_CASE(0, 1, false) RW_CASE(1, 0, false) RW_CASE(1, 1, false) RW_CASE(2, 0, false) RW_CASE(2, 1, false)
  if constexpr (NKC == 4) {
    RW_CASE(0, 0, true) RW_CASE(0, 1, true) RW_CASE(1, 0, true) RW_CASE(1, 1, true) RW_CASE(2, 0, true) RW_CASE(2, 1, true)
  }
#undef RW_CASE
  if (rc) return rc;
  return tdvc_launch_status("tdvc_conv2d(conv_row)");
}

int launch_conv_row(const ConvParams& p, int N, hipStream_t st) {
  return p.Cin == 128 ? launch_conv_row_t<4>(p, N, st) : launch_conv_row_t<2>(p, N, st);
}
