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
//
// Geometries (RowGeo): the same code runs the Cin = 128 -> 128 k layers (32-column strips), the single Cin = 64 -> 64 k layers (64-column
// strips, two halves of the strip per block of 16 output channels), the Cin = 128 -> 64 layers (64-column strips, 17-piece rows in a 6-row
// ring) and the space-to-depth form of the 3x3 STRIDE-2 convs with 64 input channels (a 2x2 conv over (H/2, W/2, 256): 512-byte ring
// pixels whose four 128-byte parity blocks the DMA gathers from two image rows; the 7 of 16 (tap, parity) blocks outside the 3x3 window
// hold zero weights and are skipped at compile time: 18 fragments per wave, two live output rows).
#include <type_traits>

#include "conv_common.h"

using convk::ConvParams;

namespace {

enum { GEO_C128 = 0, GEO_C64 = 1, GEO_C128W = 2, GEO_S2D64 = 3 };
// NKC: 32-channel chunks of a ring pixel; KH x KW: window (pad 1 at the top / left, and for 3x3 at the bottom / right); SW: output columns
// per strip; NCG: blocks of 16 output channels per workgroup pass (a wave computes two 16-pixel column blocks; 8 / NCG waves share a block);
// PIECES: 1-KB DMA pieces per ring row; XRING / PF / HD: ring rows, rows the DMA runs ahead, steps of DMA history land_wait() leaves in flight;
// DMA_POS: where in a step a loader issues the next row (0 top, 1 after half of its fragment groups, 2 end: measured per geometry).
template <int GID> struct RowGeo;
template <> struct RowGeo<GEO_C128> { static constexpr int NKC = 4, KH = 3, KW = 3, SW = 32, NCG = 8, PIECES = 9, XRING = 8, PF = 6, HD = 4, DMA_POS = 1, CORIG = 128; static constexpr bool S2D = false; };
template <> struct RowGeo<GEO_C64> { static constexpr int NKC = 2, KH = 3, KW = 3, SW = 64, NCG = 4, PIECES = 9, XRING = 8, PF = 6, HD = 4, DMA_POS = 2, CORIG = 64; static constexpr bool S2D = false; };
template <> struct RowGeo<GEO_C128W> { static constexpr int NKC = 4, KH = 3, KW = 3, SW = 64, NCG = 4, PIECES = 17, XRING = 6, PF = 4, HD = 2, DMA_POS = 2, CORIG = 128; static constexpr bool S2D = false; };
template <> struct RowGeo<GEO_S2D64> { static constexpr int NKC = 8, KH = 2, KW = 2, SW = 32, NCG = 8, PIECES = 17, XRING = 6, PF = 4, HD = 2, DMA_POS = 1, CORIG = 64; static constexpr bool S2D = true; };
template <int GID> struct RowDerived : RowGeo<GID> {
  using G = RowGeo<GID>;
  static constexpr int CO = 16 * G::NCG;                 // output channels per pass
  static constexpr int PXB = G::NKC * 64;                // bytes per ring pixel
  static constexpr int LPP = PXB / 16;                   // 16-byte positions (= DMA lanes) per ring pixel
  static constexpr int SLOTS = CO / 8;                   // 16-byte positions per staging pixel
  static constexpr int ROWB = G::PIECES * 1024;          // ring row bytes
  static constexpr int STEPS = (G::KH * G::KW * 4 + 1) / 2;      // k-steps per chunk of the standard packed blob
  static constexpr int NP = G::PIECES / 4 + 1;           // DMA instructions a loader may issue per row: PIECES / 4 + the odd piece
  static constexpr int X0 = 0, S0 = G::XRING * ROWB, B0 = S0 + 4 * 8192, LDS = B0 + 512;
  static_assert(G::SW * CO * 2 == 8192, "a staging row is 8 KB = 512 items of 16 bytes");
  static_assert((G::SW + G::KW - 1) * PXB <= ROWB && (G::PIECES % 4) == 1, "ring row");
  static_assert(G::SW / 16 == 2 * (8 / G::NCG), "a wave computes two column blocks");
  // (1) input ring, write-after-read: the DMA issued in step k by the fastest loader overwrites row k + PF - XRING; the slowest wave
  //     of the interval is at step >= k - (BI - 1) and reads exactly that step's row: k + PF - XRING < k - (BI - 1).
  // (2) landing: a loader's own pieces of row r (issued in step r - PF) are known to have landed after its land_wait() of a step
  //     s >= r - PF + HD, the other waves learn it at the next barrier; the last barrier before step r closes a step >= r - BI.
  static_assert(G::PF + 2 <= G::XRING && G::PF >= G::HD + 2, "ring conditions (BI = 2)");
  static_assert(G::HD == 4 || G::HD == 2, "land_wait(): four-step history with the odd piece going round the loaders, or two steps with a fixed owner");
  static_assert(LDS <= 160 * 1024, "LDS");
};
// ---- synchronisation geometry (step k consumes input row k; a workgroup barrier closes every BI-th step; between two barriers
// two waves are at most BI - 1 steps apart and always in the same barrier interval).
constexpr int RW_BI = 2;
constexpr int RW_SRING = 2 * RW_BI;             // staging rows: written in step k, stored in step k + BI (the next barrier interval), overwritten in step k + SRING
constexpr int RW_SROW = 8192;
constexpr int RW_NTHR = 512;
#ifndef RW_DMA_POS
#define RW_DMA_POS G::DMA_POS      /* measured (tools/ab_row.py, r04): Cin = 128: 133 / 141 / 144 us for mid / top / end at 128 -> 128 @544x960; Cin = 64: 169 / 177 / 158 us at 64 -> 64 @1088x1920 */
#endif

struct RowParams {
  const half_t* x; long x_sn; int x_sp;
  half_t* y; long y_sn; int y_sp;
  const half_t* res; long r1_sn; int r1_sp;
  const half_t* w;          // the layer's standard blob: [cout tile 32][chunk NKC][k-step STEPS][lane 64][8 halves]
  const float* bias;
  const half_t* zeros;      // >= 16 B of zeros: DMA source of out-of-image pixels
  half_t* dump;             // >= 8 KB nobody reads: store target of items outside the image
  int N, H, W;              // OUTPUT rows / columns (= the input's, or half of them in the space-to-depth geometry)
  int Win;                  // input image row pitch in pixels (W, or 2 W)
  int ncb;                  // cout blocks of CO channels
  int cq;                   // PixelShuffle(2) store: channels after the shuffle (cout / 4); 0 = plain NHWC
  int strips;               // SW-column strips per image row
  float slope;              // max(v, v * slope): 1 = none, 0 = ReLU
  int reverse;
};

template <int PXB> __device__ __forceinline__ int rw_f(int q) {                                          // ring-row swizzle term
  return PXB == 128 ? ((q >> 1) & 7) : ((q & 12) | ((q & 1) << 1) | ((q >> 1) & 1));
}
__device__ __forceinline__ int rw_sigma(int r) { return r < 4 ? 2 * r : (r >= 12 ? 2 * r - 16 : 2 * r - 7); }   // fragment lane -> pixel of its 16-px block
__device__ __forceinline__ int rw_g(int q) { return (q >> 1) & 7; }                                      // staging-row swizzle term

__device__ __forceinline__ void rw_glds16(const half_t* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// ring slot of row kk: a mask for the 8-row rings; the 6-row rings divide by a constant -- on the scalar unit (kk is wave-uniform; left to
// itself hipcc did the multiply-high in vector registers, three more live registers in kernels at the 256-register limit)
template <int XRING> __device__ __forceinline__ int rw_slot(int kk) {
  if constexpr ((XRING & (XRING - 1)) == 0) return kk & (XRING - 1);
  else return __builtin_amdgcn_readfirstlane(kk) % XRING;
}
__device__ __forceinline__ void rw_barrier(bool skip = false) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (!skip) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// which (tap row dy, tap column dx, chunk kc) fragments exist: all of them for a dense window; in the space-to-depth geometry chunk kc
// belongs to parity block (py, px) = bits of kc / (CORIG / 32), and virtual tap (dy, dx) of that block is the original tap
// (2 dy + py - 1, 2 dx + px - 1): outside the 3x3 window (zero weights) when dy == 0 && py == 0 or dx == 0 && px == 0
template <int GID> constexpr bool rw_frag(int dy, int dx, int kc) {
  using G = RowGeo<GID>;
  if (!G::S2D) return true;
  const int q = kc / (G::CORIG / 32), py = q >> 1, px = q & 1;
  return (dy == 1 || py == 1) && (dx == 1 || px == 1);
}
template <int GID> constexpr bool rw_group(int dx, int kc) {       // a fragment group (dx, kc) exists when any of its tap rows does
  for (int dy = 0; dy < RowGeo<GID>::KH; ++dy)
    if (rw_frag<GID>(dy, dx, kc)) return true;
  return false;
}
template <int GID> constexpr int rw_ngroups() {
  int n = 0;
  for (int g = 0; g < RowGeo<GID>::KW * RowGeo<GID>::NKC; ++g) n += rw_group<GID>(g % RowGeo<GID>::KW, g / RowGeo<GID>::KW);
  return n;
}
template <int GID> constexpr int rw_group_at(int i) {              // the i-th existing group, as kc * KW + dx
  int n = 0;
  for (int g = 0; g < RowGeo<GID>::KW * RowGeo<GID>::NKC; ++g)
    if (rw_group<GID>(g % RowGeo<GID>::KW, g / RowGeo<GID>::KW)) {
      if (n == i) return g;
      ++n;
    }
  return -1;
}
template <int GID> constexpr int rw_first_group_of_row0() {        // the first group in which the row STARTED in a step (dy = 0) has a fragment
  for (int i = 0; i < rw_ngroups<GID>(); ++i) {
    const int g = rw_group_at<GID>(i);
    if (rw_frag<GID>(0, g % RowGeo<GID>::KW, g / RowGeo<GID>::KW)) return i;
  }
  return -1;
}

// ACT: 0 none, 1 ReLU, 2 max(v, v * slope); NRES: 0 / 1 fp16 residual added in the store phase (no such layer of the model has two; a launch
// with two stays on conv_mfma_v11 / v10); SHUF: PixelShuffle(2) store
template <int GID, int ACT, int NRES, bool SHUF>
__global__ __launch_bounds__(RW_NTHR, 1) void conv_row_kernel(const RowParams p) {
  using G = RowDerived<GID>;
  constexpr int NKC = G::NKC, KH = G::KH, KW = G::KW, SW = G::SW, CO = G::CO, PXB = G::PXB, XRING = G::XRING, PF = G::PF, NP = G::NP;
  constexpr int NG = rw_ngroups<GID>(), G0 = rw_first_group_of_row0<GID>();
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave < 4;                // wave-uniform roles: waves 0-3 issue the DMA, waves 4-7 store
  const int wq = wave & 3;
  const int cg = wave % G::NCG, cbp = wave / G::NCG;        // block of 16 output channels; pair of column blocks (2 cbp, 2 cbp + 1) of the strip
  const int r16 = lane & 15, kb = lane >> 4;

  // ---- workgroup -> (cout block, run of rows).  The launch's work is the flat sequence of (image, strip, row) = N * strips * H
  // rows of SW output columns; a cout block's S workgroup slots cut it into S equal runs (a run that crosses a strip end is two
  // segments), so every workgroup gets the same number of rows whatever H and the strip count are (a segment pays KH - 1 + BI extra
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

  // ---- this wave's 16 output channels: its A fragments (tap, chunk) in registers for the whole launch.  Lane (r16, kb) of
  // fragment (tap, kc) holds w[cout][cin = 32 kc + 8 kb .. + 7][tap]: in the standard blob that is k-step tap * 2 + (kb >> 1),
  // half (kb & 1), row cout & 31 of cout tile cout >> 5
  half8 wf[KH * KW * NKC];
  {
    const int co = cbk * CO + 16 * cg + r16;
    const half_t* wb = p.w + ((long)(co >> 5) * NKC * G::STEPS * 64 + (kb & 1) * 32 + (co & 31)) * 8 + (long)(kb >> 1) * 512;
#pragma unroll
    for (int tap = 0; tap < KH * KW; ++tap)
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc)
        if (rw_frag<GID>(tap / KW, tap % KW, kc))
          wf[tap * NKC + kc] = *reinterpret_cast<const half8*>(wb + ((long)kc * G::STEPS + tap * 2) * 512);
  }
  // the bias is the C operand of every accumulator chain's first MFMA: re-read from LDS per step (4 registers of a 256-register budget)
  if (tid < CO) reinterpret_cast<float*>(smem + G::B0)[tid] = p.bias[cbk * CO + tid];
  const int boff = G::B0 + (16 * cg + 4 * kb) * 4;     // C/D rows 4 kb + i of this wave's 16 channels

  // ---- per-lane LDS offsets
  int foff[KW];                                // B fragment (dx), chunk 0, column block 0; chunk kc: ^ (kc << 6); block 1: + 16 pixels
#pragma unroll
  for (int dx = 0; dx < KW; ++dx) {
    const int q = 32 * cbp + dx + rw_sigma(r16), f = rw_f<PXB>(q);
    foff[dx] = q * PXB + ((kb ^ f) << 4);              // (4 kc + kb) ^ f = (kb ^ f) ^ (kc << 2)
  }
  int doff[2];                                 // pack: pixel 16 cb + SIGMA(r16), channels 16 cg + 4 kb .. + 3 of the block
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int q = 32 * cbp + 16 * cb + rw_sigma(r16), c = 2 * cg + (kb >> 1);
    doff[cb] = q * (CO * 2) + ((c ^ rw_g(q)) << 4) + 8 * (kb & 1);
  }
  // ---- everything below is instantiated once per ROLE (waves 0-3: loaders, waves 4-7: storers) inside one wave-uniform branch, so that a
  // role's private registers (DMA offsets and masks / output offsets and the prefetched residual row) share physical registers with
  // the other role's instead of adding up: the kernel lives within 256 registers with 144 of them holding weights
  auto run = [&](auto LOADERc) __attribute__((always_inline)) {
  constexpr bool LOADER = decltype(LOADERc)::value;
  // DMA items (loader waves): piece j of a row covers 1 KB of ring pixels (lane: position lane % LPP of pixel (64 / LPP) j + lane / LPP);
  // ring pixel q is image column c0 - 1 + q.  Loader wq sends pieces wq, wq + 4, ...; the odd piece (the last) goes round the four loaders
  // (HD = 4) or stays with loader 0 (HD = 2): either way a loader's count over HD consecutive steps is a constant.  Space-to-depth:
  // position c of ring pixel q is channels 8 (c % (CORIG / 8)) .. of image pixel (2 row + py, 2 (c0 - 1 + q) + px), (py, px) = c / (CORIG / 8)
  // Dense geometries: piece wq + 4 j starts 16 (PXB = 256) or 32 (PXB = 128) ring pixels behind piece wq + 4 (j - 1), a multiple of the swizzle
  // period: one lane offset serves all of a loader's regular pieces (+ a wave-uniform stride), a second one the odd piece.
  constexpr int SRC = G::S2D ? NP : 2;         // distinct per-lane source offsets
  constexpr int QSTEP = 4 * (64 / G::LPP);     // ring pixels between a loader's consecutive regular pieces
  static_assert(G::S2D || (QSTEP % 16) == 0, "regular pieces of a loader share their swizzle terms");
  int soff[SRC], sq[SRC];
#pragma unroll
  for (int j = 0; j < SRC; ++j) {
    const int piece = j == SRC - 1 ? G::PIECES - 1 : wq + 4 * j;
    const int q = (64 / G::LPP) * piece + lane / G::LPP, c = (lane % G::LPP) ^ rw_f<PXB>(q);
    sq[j] = q;
    if constexpr (G::S2D) {
      const int par = c / (G::CORIG / 8), within = c % (G::CORIG / 8);
      soff[j] = (2 * q + (par & 1)) * p.x_sp + (par >> 1) * p.Win * p.x_sp + within * 8;
    } else {
      soff[j] = q * p.x_sp + c * 8;
    }
  }
  // store items (storer waves): item it = (tid - 256) + 256 j: pixel it / SLOTS, position it % SLOTS of the staging row
  const int s_t = tid & 255;
  int s_px[2], s_co[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int it = s_t + 256 * j;
    s_px[j] = it / G::SLOTS;
    s_co[j] = cbk * CO + (((it % G::SLOTS) ^ rw_g(it / G::SLOTS)) << 3);
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
  f32x4 acc[KH][2];
  half8 r1v[2] = {};

  for (long pos = run_begin; pos < run_end;) {
    const int sid = (int)(pos / p.H);          // (image, strip)
    const int ra = (int)(pos - (long)sid * p.H), rb = (int)min((long)p.H, ra + (run_end - pos));
    const int rows = rb - ra;
    pos += rows;
    const int n = sid / p.strips, strip = sid - n * p.strips;
    const int c0 = strip * SW;
    const half_t* xn = p.x + (long)n * p.x_sn;
    const int last_in = rows + KH - 2;         // the last step that consumes an input row

    unsigned colok = 0;                        // bit j: this lane's pixel of its j-th piece is one of the row's SW + KW - 1 and inside the image
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int q = G::S2D ? sq[j] : (j == NP - 1 ? sq[1] : sq[0] + QSTEP * j);
      colok |= (q < SW + KW - 1 && c0 - 1 + q >= 0 && c0 - 1 + q < p.W) ? (1u << j) : 0u;
    }
    // this thread's two output items: a row's base address is wave-uniform (scalar registers), a lane adds a 32-bit element offset.
    // Lanes outside the image load their residual from the last valid column (the value goes to the dump page): no address selects
    // on the loads.
    bool s_ok[2];
    int y_lo[2], r_lo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = c0 + s_px[j], colc = min(col, p.W - 1);
      s_ok[j] = col < p.W;
      const int off = SHUF ? (s_sub[j] >> 1) * OW + 2 * col + (s_sub[j] & 1) : col;       // pixel indices fit 32 bits (checked by the host)
      const int offc = SHUF ? (s_sub[j] >> 1) * OW + 2 * colc + (s_sub[j] & 1) : colc;
      y_lo[j] = off * p.y_sp + s_pc[j];
      r_lo[j] = offc * p.r1_sp + s_pc[j];
    }
    const half_t* const yimg = p.y + (long)n * p.y_sn;
    const half_t* const rimg = NRES >= 1 ? p.res + (long)n * p.r1_sn : nullptr;
    const long yrow_stride = (long)(SHUF ? 2 : 1) * OW * p.y_sp, rrow_stride = (long)(SHUF ? 2 : 1) * OW * p.r1_sp;
    const long xrow_stride = (long)(G::S2D ? 2 : 1) * p.Win * p.x_sp;       // one ring row = one image row (two in the space-to-depth geometry)
    const half_t* const xcol = xn + (long)(G::S2D ? 2 : 1) * (c0 - 1) * p.x_sp;
#pragma unroll
    for (int s3 = 0; s3 < KH; ++s3)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[s3][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ring row ra - 1 + kk into ring slot kk % XRING; returns the number of DMA instructions issued
    auto issue_row = [&](int kk) __attribute__((always_inline)) -> int {
      const int row = ra - 1 + kk;
      const bool rowok = row >= 0 && row < p.H;
      const half_t* base = xcol + row * xrow_stride;
      const unsigned dst = lds0 + G::X0 + rw_slot<XRING>(kk) * G::ROWB;
      const unsigned okb = rowok ? colok : 0u;
      if constexpr (G::S2D || NP <= 3) {
#pragma unroll
        for (int j = 0; j < NP - 1; ++j) {
          const half_t* src = G::S2D ? base + soff[j] : base + (long)(QSTEP * j) * p.x_sp + soff[0];
          rw_glds16(((okb >> j) & 1u) ? src : p.zeros, dst + (wq + 4 * j) * 1024);
        }
      } else {
        // four regular pieces per loader (the 17-piece rows): a rolled loop, so that the four source addresses are formed one at a time
        // (unrolled, hipcc forms them all up front: 10 more live registers in a kernel that sits at the 256-register limit)
#pragma nounroll
        for (int j = 0; j < NP - 1; ++j) {
          const half_t* src = base + (long)(QSTEP * j) * p.x_sp + soff[0];
          rw_glds16(((okb >> j) & 1u) ? src : p.zeros, dst + (wq + 4 * j) * 1024);
        }
      }
      if (G::HD == 4 ? wq == (kk & 3) : wq == 0) {
        rw_glds16(((okb >> (NP - 1)) & 1u) ? base + soff[SRC - 1] : p.zeros, dst + (G::PIECES - 1) * 1024);
        return NP;
      }
      return NP - 1;
    };

    // ---- prologue: the first PF rows; everything landed and visible before step 0
    if constexpr (LOADER) {
#pragma unroll
      for (int kk = 0; kk < PF; ++kk)
        if (kk <= last_in) issue_row(kk);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rw_barrier();

    const int K = (rows + KH - 1 + RW_BI + RW_BI - 1) & ~(RW_BI - 1);     // steps, a whole number of barrier intervals
    // A loader's count of instructions left in flight is a constant while rows are being issued: HD = 4: the odd piece goes round the four
    // loaders, 4 (NP - 1) + 1 over any four consecutive steps; HD = 2: loader 0 owns it, 2 NP, the others 2 (NP - 1).  Shorter histories
    // (the first steps of a segment) have fewer outstanding than that and pass without waiting; once issuing has stopped the wait is vmcnt(0).
    auto land_wait = [&](int nvm) __attribute__((always_inline)) {
      if (!nvm) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
      if constexpr (G::HD == 4) {
        static_assert(G::HD != 4 || NP == 3, "vmcnt(9)");
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      } else {
        static_assert(G::HD != 2 || NP == 5, "vmcnt(10) / vmcnt(8)");
        if (wq == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
    };

    // One step k: every wave consumes ring row i = ra - 1 + k: tap row dy of it belongs to output row i + 1 - dy (started: dy = 0, finished:
    // dy = KH - 1 -> staging row k); storer waves first send the row finished BI steps ago (staging row k - BI, image row
    // ra + k - BI - (KH - 1)) to memory.
    auto step = [&](auto PHc, auto FULLc, int k) __attribute__((always_inline)) {
      constexpr int PH = decltype(PHc)::value;       // k % KH: tap row dy accumulates into slot (PH + KH - dy) % KH
      constexpr bool FULL = decltype(FULLc)::value;
      constexpr int SD = (PH + 1) % KH;              // the finishing row: dy = KH - 1
      if constexpr (!LOADER) {
        const int srow = ra + k - RW_BI - (KH - 1);
        if (FULL || (srow >= ra && srow < rb)) {
          const unsigned char* sb = smem + G::S0 + ((k - RW_BI) & (RW_SRING - 1)) * RW_SROW;
          half_t* const yrow = const_cast<half_t*>(yimg) + srow * yrow_stride;      // wave-uniform
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            half8 yv = *reinterpret_cast<const half8*>(sb + (s_t + 256 * j) * 16);
            if constexpr (NRES >= 1) yv = yv + r1v[j];
            // no branch around the store: behind a branch hipcc drains vmcnt(0) at the join, i.e. in front of the residual loads below
            // (measured: 128 -> 128 + residual @544x960 145 -> 175 us); lanes outside the image store into the dump page instead
            *reinterpret_cast<half8*>(s_ok[j] ? yrow + y_lo[j] : p.dump + (s_t + 256 * j) * 8) = yv;
          }
        }
        if constexpr (NRES >= 1) {
          const int nrow = srow + 1;
          const bool rok = FULL || (nrow >= ra && nrow < rb);
          const half_t* const rrow = rimg + (rok ? nrow : ra) * rrow_stride;        // wave-uniform; a row outside the segment reads a valid one
#pragma unroll
          for (int j = 0; j < 2; ++j) r1v[j] = *reinterpret_cast<const half8*>(rrow + r_lo[j]);
        }
      }
      int nvm = 0;
      if (FULL || k <= last_in) {
        const unsigned char* xb = smem + G::X0 + rw_slot<XRING>(k) * G::ROWB;
        // NG groups (kc, dx) of up to 2 KH MFMAs (KH output rows x 2 column blocks) on two B fragments; the fragments of group g + FBN - 1
        // are requested before group g's MFMAs (FBN fragment pairs in flight)
        constexpr int FBN = (PXB == 256 && (NRES == 1 || GID == GEO_C128W)) ? 2 : 3;      // the 128-channel geometries sit at the 256-register limit
        half8 fb[FBN][2];
        auto load_group = [&](int gi, int buf) __attribute__((always_inline)) {
          const int g = rw_group_at<GID>(gi), kc = g / KW, dx = g - KW * kc;
          const unsigned char* b0 = xb + (foff[dx] ^ (kc << 6));
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) fb[buf][cb] = *reinterpret_cast<const half8*>(b0 + cb * 16 * PXB);
        };
        const f32x4 bias4 = *reinterpret_cast<const f32x4*>(smem + boff);
        load_group(0, 0);
        if (FBN == 3) load_group(1, 1);
        if (RW_DMA_POS == 0 && LOADER && k + PF <= last_in) nvm = issue_row(k + PF);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
          const int g = rw_group_at<GID>(gi), kc = g / KW, dx = g - KW * kc;
          if (gi + FBN - 1 < NG) load_group(gi + FBN - 1, (gi + FBN - 1) % FBN);
          // the loader's DMA in the MIDDLE of its MFMA stream: its ~100 instructions of address work and the pieces' issue time run
          // while the matrix pipe works off this wave's queued MFMAs (at the top of the step both waves of a SIMD did their role
          // work side by side in front of an idle pipe)
          if (RW_DMA_POS == 1 && gi == NG / 2 && LOADER && k + PF <= last_in) nvm = issue_row(k + PF);
#pragma unroll
          for (int dyo = 0; dyo < KH; ++dyo) {
            const int dy = KH - 1 - dyo;       // the finishing row first: its chain ends before the step's last MFMAs
            if (!rw_frag<GID>(dy, dx, kc)) continue;
            const int slot = (PH + KH - dy) % KH;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
              const bool first = dy == 0 && gi == G0;
              acc[slot][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[(dy * KW + dx) * NKC + kc], fb[gi % FBN][cb],
                                                                     first ? bias4 : acc[slot][cb], 0, 0, 0);
            }
          }
        }
        if (RW_DMA_POS == 2 && LOADER && k + PF <= last_in) nvm = issue_row(k + PF);
        {
          // finished row -> fp16, activation, into staging row k (rows outside the segment are never stored)
          unsigned char* sb = smem + G::S0 + (k & (RW_SRING - 1)) * RW_SROW;
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
    auto steps_of = [&](auto FULLc, int k0, int kend) __attribute__((always_inline)) {    // up to KH steps from k0 (k0 % KH == 0), while < kend
      step(std::integral_constant<int, 0>{}, FULLc, k0);
      if (k0 + 1 < kend) step(std::integral_constant<int, 1 % KH>{}, FULLc, k0 + 1);
      if constexpr (KH == 3) {
        if (k0 + 2 < kend) step(std::integral_constant<int, 2 % KH>{}, FULLc, k0 + 2);
      }
    };
    constexpr int KF = ((RW_BI + KH - 1 + KH - 1) / KH) * KH;       // first steady-state step: a multiple of KH >= BI + KH - 1
    int k = 0;
    for (; k < KF; k += KH) steps_of(std::false_type{}, k, KF);
    for (; k + KH - 1 <= last_in; k += KH) steps_of(std::true_type{}, k, k + KH);
    for (; k < K; k += KH) steps_of(std::false_type{}, k, K);
    rw_barrier();                              // the next job's prologue overwrites ring rows the slowest wave may still read
  }
  };
  if (loader) run(std::true_type{});
  else run(std::false_type{});
}

}  // namespace

static int g_row_enabled = 15;
// tests and A/B benchmarks: bit 0 = the Cin = 128 -> 128 k layers (off: conv_mfma_v11 takes them), bit 1 = Cin = 64 (off: conv_mfma_v10),
// bit 2 = Cin = 128 -> 64 (2k + 1) (off: conv_mfma_v11), bit 3 = the space-to-depth form of the stride-2 convs (off: conv_mfma_v3)
extern "C" void tdvc_debug_enable_conv_row(int enable) { g_row_enabled = enable; }

static bool row_off() {
  static const bool off = getenv("TDVC_CONV_NO_ROW") != nullptr || getenv("TDVC_CONV_V1") != nullptr;
  return off || !g_row_enabled;
}

// -> geometry id, or -1
int conv_row_geometry(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo) {
  if (row_off() || d->ck != 32 || d->square_input || !d->bias || (long)Ho * Wo < 8192 || Ho < 16 || (d->res.p && d->res2.p)) return -1;
  const bool shuf = p.out_mode == TDVC_OUT_SHUFFLE2;
  const int ych = shuf ? (d->cout >> 2) : d->cout;       // channels of an output pixel
  if (d->y.C < ych || (d->res.p && d->res.C < ych) || (d->res2.p && d->res2.C < ych)) return -1;
  if (d->s2d) {                                  // the virtual 2x2 / stride 1 / pad 1 conv over the space-to-depth view of a 3x3 stride-2 conv
    const bool ok = (g_row_enabled & 8) && d->x.C == 64 && (d->cout % 128) == 0 && convk::conv_is_lean(p) && d->kh == 2 && d->kw == 2 && d->ntaps == 4 &&
                    d->tap_dy[0] == 0 && d->tap_dx[0] == 0 && d->tap_dy[1] == 0 && d->tap_dx[1] == 1 && d->tap_dy[2] == 1 && d->tap_dx[2] == 0 &&
                    d->tap_dy[3] == 1 && d->tap_dx[3] == 1;
    return ok ? GEO_S2D64 : -1;
  }
  bool taps33 = d->ntaps == 9 && d->kh == 3 && d->kw == 3 && d->pad == 1 && d->stride == 1;
  for (int t = 0; taps33 && t < 9; ++t) taps33 = d->tap_dy[t] == t / 3 && d->tap_dx[t] == t % 3;
  if (!taps33) return -1;
  const bool lean = convk::conv_is_lean(p);
  if (d->x.C == 128 && (d->cout % 128) == 0 && (g_row_enabled & 1) && (lean || (convk::conv_is_simple(p) && !p.gdn && shuf && (ych % 8) == 0))) return GEO_C128;
  if (d->x.C == 128 && (d->cout % 64) == 0 && (g_row_enabled & 4) && lean) return GEO_C128W;
  if (d->x.C == 64 && (d->cout % 64) == 0 && (g_row_enabled & 2) && lean) return GEO_C64;
  return -1;
}

template <int GID>
static int launch_conv_row_t(const ConvParams& p, int N, hipStream_t st) {
  using G = RowDerived<GID>;
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
  q.N = N; q.H = p.Ho; q.W = p.Wo; q.Win = p.W;
  q.ncb = p.cout / G::CO;
  const bool shuf = p.out_mode == TDVC_OUT_SHUFFLE2;
  q.cq = shuf ? p.cout >> 2 : 0;
  q.slope = convk::conv_simple_slope(p);
  q.reverse = p.reverse;
  q.strips = (q.W + G::SW - 1) / G::SW;
  // 256 workgroups (one per CU), 256 / ncb slots per cout block; fewer when the launch has fewer rows than slots
  const long allrows = (long)N * q.strips * q.H;
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
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (err != hipSuccess) { tdvc_set_error("conv_row: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(RW_NTHR), G::LDS, st, q);
    return 0;
  };
  const int nres = has1 ? 1 : 0;
  int rc = TDVC_EINVAL;
#define RW_CASE(A, R, S) if (act == A && nres == R && shuf == S) rc = go(&conv_row_kernel<GID, A, R, S>);
  RW_CASE(0, 0, false) RW_CASE(0, 1, false) RW_CASE(1, 0, false) RW_CASE(1, 1, false) RW_CASE(2, 0, false) RW_CASE(2, 1, false)
  if constexpr (GID == GEO_C128) {
    RW_CASE(0, 0, true) RW_CASE(0, 1, true) RW_CASE(1, 0, true) RW_CASE(1, 1, true) RW_CASE(2, 0, true) RW_CASE(2, 1, true)
  }
#undef RW_CASE
  if (rc) return rc;
  return tdvc_launch_status("tdvc_conv2d(conv_row)");
}

int launch_conv_row(int geo, const ConvParams& p, int N, hipStream_t st) {
  switch (geo) {
    case GEO_C128: return launch_conv_row_t<GEO_C128>(p, N, st);
    case GEO_C64: return launch_conv_row_t<GEO_C64>(p, N, st);
    case GEO_C128W: return launch_conv_row_t<GEO_C128W>(p, N, st);
    default: return launch_conv_row_t<GEO_S2D64>(p, N, st);
  }
}
