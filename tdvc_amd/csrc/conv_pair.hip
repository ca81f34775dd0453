// conv_pair — two chained 3x3 / stride 1 / pad 1 / 64 -> 64 convs in ONE launch, the intermediate map never leaving the CU:
//     y = act2(conv2(act1(conv1(x)))) [+ x] [+ res2]
// (`Res_Block`, main/utils/utils.py:43-56: ReLU between, identity added; the conv pairs of flownet.py / pnet.py with
// LeakyReLU on both).  As two launches of conv_mfma_v10 the pair moves 5 maps through HBM (x, t, t, x, y: 640 B per pixel)
// and each launch sits on the per-CU memory rate with its MFMA pipes half idle ("What bounds the 64-channel
// 3x3 class").  Fusing through LDS needs both weight sets on chip (2 x 72 KB) next to the tiles: they do not fit in 160 KB
// of LDS -- but they fit in REGISTERS once the work is split by output channel and by conv:
//
//   * 8 waves, two per SIMD.  Waves 0-3 compute conv1, waves 4-7 conv2; wave w of a group owns output channels
//     [16 (w & 3), +16) of its conv and keeps their weights as v_mfma_f32_16x16x32_f16 A fragments: 9 taps x 2 chunks of 32
//     input channels = 18 fragments = 72 VGPRs, loaded once per launch.  LDS holds activations only.  A SIMD hosts one wave
//     of each group: while one waits (LDS latency, a vector-memory instruction being accepted, the pack of a finished row)
//     the other issues MFMAs.  (A first version with 4 waves holding both weight sets lost a third of its cycles to exactly
//     those waits: a wave alone on its SIMD pays every one of them.)
//   * row streaming.  A workgroup walks a strip of 30 output columns top to bottom (the numbers of this header are the 30-column
//     geometry's, PairGeo<2>; the 62-column geometry of round 4, PairGeo<4>, is described where it is defined below).  Per step one
//     input row (34 px, by LDS-DMA into a 16-row ring, 8 rows ahead) enters conv1: its B fragments (16 px x 32 channels) feed the three live
//     rows of t (dy = 0, 1, 2: three rotating accumulator rows).  The finished row of t (32 px) is written to a small LDS
//     ring as fp16 -- what the unfused path stores to HBM -- and one step later enters conv2 the same way; the finished
//     row of y picks up the identity from the input ring, goes through an LDS staging row and leaves as full 128-byte
//     lines.  No tile epilogue, no halo recompute in y (2 of 32 conv2 columns are waste: 97 % useful MFMAs); the vertical
//     halo costs 5 extra steps per strip segment.
//   * per step and SIMD: 72 MFMAs (1152 cycles), 24 ds_read_b128, one s_barrier per BI = 2 steps; per CU 4.4 KB in + 3.8 KB out per step =
//     7 B per cycle at full MFMA rate -- under the ~10 B per cycle a CU can move.
//   * HBM traffic of the pair: x once (+ 13 % column halo, mostly L2 hits) and y once: 256 B per pixel instead of 640.
//
// LDS image of a ring row: pixel-major, 128 B per pixel (64 channels), the 16-byte chunk c of pixel q stored at slot
// c ^ G[(q >> 1) & 7], G = {0, 1, 2, 4, 5, 6, 2, 6}.  A B fragment is read by ds_read_b128 with lane = (pixel p0 + (l & 15),
// channel block l >> 4); the instruction is served in four groups of 16 lanes, each of which holds the 16 pixels once, the
// middle eight with the neighbouring channel block (c ^ 1).  The plain (q >> 1) & 7 is conflict-free only for p0 = 0 mod 4
// (the dx = 0 tap); the taps dx = 1, 2 shift the window and two pairs of lanes meet in one 16-byte bank slot (rocprofv3:
// SQ_LDS_BANK_CONFLICT = 40 % of SQ_LDS_IDX_ACTIVE).  G is one of the 720 tables (exhaustive search) that keep all three
// windows conflict-free; it costs the few 8-byte C/D-layout accesses (t / staging writes, identity read) a second pass.
// For the DMA-written input rows the XOR sits on the SOURCE address.
#include <type_traits>

#include "conv_common.h"

namespace {

// Two strip geometries (PairGeo<NCB>, NCB = 16-column blocks a wave computes per row):
//   NCB = 2: 30 output columns per strip, 34-pixel ring rows (5 DMA pieces), 16-row input ring 8 rows ahead -- rounds 2 and 3;
//   NCB = 4 (round 4): 62 output columns per strip, 66-pixel ring rows (9 pieces), 10-row input ring 4 rows ahead.  Per step and SIMD
//            144 MFMAs (2 304 cycles) stand against the same barrier / first-fragment latency / pack overheads as 72 did (the measure that
//            took conv_row past 0.50 of the peak), the column halo falls from 13 % to 6 %, and a 1920-column map is 31 strips
//            (30 x 62 + 60).  The ring is as deep as 160 KB of LDS allow: 10 x 9 KB + t ring 4 x 9 KB + staging ring 4 x 8 KB = 158 KB.
// ---- synchronisation geometry.  A "step" k consumes one input row; a workgroup barrier closes every BI-th step, so between
// two barriers any two waves are at most BI - 1 steps apart and always in the same barrier interval.  Every ring / landing
// condition the kernel relies on is one named inequality below (row numbers in step units: "row r" = what step r consumes).
constexpr int BI = 2;                        // row steps per workgroup barrier (a power of two)
constexpr int TRING = 2 * BI;                // t ring rows: conv1 writes row k - 1 in step k, conv2 reads row k - 1 - BI
constexpr int SRING = 2 * BI;                // staging ring rows: conv2 writes row k in step k, stores row k - BI
constexpr int T_LAG = 1 + BI;                // conv2 consumes t row r in step r + T_LAG (r is written in step r + 1)
constexpr int ID_LAG = 2 + BI;               // conv2 reads the identity (x ring row k - ID_LAG) in step k
template <int NCB> struct PairGeoBase;
// XRING: input ring rows; PF: the DMA of row k + PF is ISSUED in step k (conv1 waves); HD: land_wait() after step s lets the DMA issued in
// steps s - HD + 1 .. s stay in flight
template <> struct PairGeoBase<2> { static constexpr int PIECES = 5, XRING = 16, PF = 8, HD = 4; };
template <> struct PairGeoBase<4> { static constexpr int PIECES = 9, XRING = 10, PF = 4, HD = 2; };
template <int NCB> struct PairGeo : PairGeoBase<NCB> {
  using G = PairGeoBase<NCB>;
  static constexpr int PW = 16 * NCB - 2;            // output columns per strip
  static constexpr int NPX = PW + 4;                 // input pixels per ring row
  static constexpr int ROWB = G::PIECES * 1024;      // ring row: 8 pixel slots of 128 B per 1-KB DMA piece
  static constexpr int TROWB = ROWB;                 // t ring row: 16 NCB pixels used; conv2's two waste columns read two more (never stored)
  static constexpr int SROW = 16 * NCB * 128;        // staging row
  static constexpr int X0 = 0, T0 = G::XRING * ROWB, S0 = T0 + TRING * TROWB;
  static constexpr int LDS = S0 + SRING * SROW;
  static_assert(NPX * 128 <= ROWB, "ring row holds the strip's input pixels");
  // (1) input ring, write-after-read.  The DMA issued in step k by the fastest conv1 wave overwrites row k + PF - XRING; the
  //     slowest wave of the interval is at step >= k - (BI - 1) and its oldest x-ring read is the identity row ID_LAG behind it:
  //     k + PF - XRING < k - (BI - 1) - ID_LAG.
  static_assert(G::PF - G::XRING < -(BI - 1) - ID_LAG, "input ring: a DMA would land on a row whose identity a conv2 wave BI - 1 steps behind has not read (needs PF + 2 BI + 2 <= XRING)");
  // (2) input ring, read-after-write (landing).  A wave's own pieces of row r (issued in step r - PF) are only known to have
  //     landed after its land_wait() of a step s >= r - PF + HD, and other waves' pieces only after a barrier behind that wait.
  //     The last barrier before step r closes a step kb >= r - BI: r - PF + HD <= r - BI.
  static_assert(G::PF >= G::HD + BI, "landing: row r must be past land_wait()'s history window before the last barrier in front of step r");
  // (3) the counted wait: conv1 waves issue nothing but DMA, a fixed number per step (NCB = 2: 1 + 1 + 1 + 2 over any four consecutive
  //     steps; NCB = 4: 2 per step, wave 0 three), so HD steps of history are a constant the hardware counter (63) holds.
  static_assert((NCB == 2 && G::HD == 4) || (NCB == 4 && G::HD == 2), "land_wait(): constant counts for these two histories");
  // (6) the identity row k - ID_LAG was consumed by conv1 in step k - ID_LAG (so it has landed) and is still in the ring by (1).
  static_assert(ID_LAG >= 1 && ID_LAG <= G::XRING - G::PF - BI, "identity row inside the input ring");
  static_assert(LDS <= 160 * 1024, "rings inside 160 KB of LDS");
};
// (4) t ring.  In one barrier interval [m BI, m BI + BI - 1] conv1 writes rows m BI - 1 .. m BI + BI - 2 and conv2 reads rows
//     m BI - T_LAG .. m BI + BI - 1 - T_LAG: 2 BI consecutive rows when T_LAG = 1 + BI, which must not alias; a row is read BI
//     steps after it was written and overwritten BI steps after it was read: a barrier lies between either pair.
static_assert(T_LAG == 1 + BI && TRING >= 2 * BI && (TRING & (TRING - 1)) == 0, "t ring: BI rows being written + BI rows being read");
// (5) staging ring: row k written in step k (each conv2 wave its 16 channels), stored by thread items in step k + BI,
//     overwritten in step k + SRING: a barrier between write and store needs the BI lag, one between store and overwrite SRING - BI >= BI.
static_assert(SRING >= 2 * BI && (SRING & (SRING - 1)) == 0, "staging ring: BI rows being written + BI rows being stored");
static_assert((BI & (BI - 1)) == 0, "power-of-two barrier interval");
// BI = 4 (2.5 % faster in isolation) fails (1)+(2) with HD = 4: PF >= 8 and PF <= 6.  With HD = 2 / PF = 6 both hold and the
// rings take 155 648 B; measured gain < 1 % on the frame's four-slice launches, so BI = 2 ships.
constexpr int NTHR = 512;
// round 4 (from conv_row.hip, tools/ab_row.py): a conv1 wave issues the next row's DMA pieces after the first third of its MFMAs instead
// of in front of them -- at the top of a step both waves of a SIMD did their non-matrix work side by side in front of an idle matrix pipe --
// and waits with a constant count
#ifndef PAIR_DMA_MID
#define PAIR_DMA_MID 1
#endif
// NCB = 4: fragment groups in flight (this one + PAIR_FBN - 1 being read) and the group behind which a conv1 wave issues its DMA
#ifndef PAIR_FBN
#define PAIR_FBN 2
#endif
#ifndef PAIR_MID_G
#define PAIR_MID_G 1
#endif

struct PairParams {
  const half_t* x; long x_sn; int x_sp;
  half_t* y; long y_sn; int y_sp;
  const half_t* res2; long r2_sn; int r2_sp;
  const half_t* w;          // [conv 2][wave 4][tap*2 + chunk 18][lane 64][8] halves (tdvc_pack_conv_pair_weights)
  const float* bias;        // [2][64]
  const half_t* zeros;      // >= 16 B of zeros: DMA source of out-of-image pixels
  half_t* dump;             // 4 KB nobody reads: store target of the lanes outside a strip (keeps the store branch-free)
  int N, H, W;
  int strips, segs, seg_rows, jobs;
  float slope1, slope2;     // max(v, v * slope): 1 = none, 0 = ReLU, else LeakyReLU
  int experiment;           // timing diagnostics (tdvc_debug_set_pair_experiment): 1 no DMA after the prologue, 2 stores to the dump line, 4 no barrier
};

// swizzle term of pixel q of a ring row (see the LDS image note at the top)
__device__ __forceinline__ int swz(int q) { return (int)((0x62654210u >> (4 * ((q >> 1) & 7))) & 7u); }

__device__ __forceinline__ void glds16(const half_t* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void pair_barrier(bool skip = false) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (!skip) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <int A>
__device__ __forceinline__ half4 actk(half4 v, half4 sl) {
  if constexpr (A == 1) {
    const half4 z = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    return __builtin_elementwise_max(v, z);
  } else if constexpr (A == 2) {
    return __builtin_elementwise_max(v, v * sl);
  } else {
    return v;
  }
}

// ring slot of row kk: a mask for the 16-row ring, a division by a constant for the 10-row ring -- on the scalar unit (kk is wave-uniform)
template <int XRING> __device__ __forceinline__ int pair_slot(int kk) {
  if constexpr ((XRING & (XRING - 1)) == 0) return kk & (XRING - 1);
  else return __builtin_amdgcn_readfirstlane(kk) % XRING;
}

// A1 / A2: activation after conv1 / conv2: 0 none, 1 ReLU, 2 max(v, v * slope) (LeakyReLU; slope 1 = none)
template <int NCB, int A1, int A2, bool ADDX, bool RES2, bool STAMP = false>
__global__ __launch_bounds__(NTHR, 1) void conv_pair_kernel(const PairParams p, long long* stamps = nullptr, int stamp_cap = 0) {
  using G = PairGeo<NCB>;
  constexpr int PW = G::PW, ROWB = G::ROWB, TROWB = G::TROWB, SROW = G::SROW, XRING = G::XRING, PF = G::PF, X0 = G::X0, T0 = G::T0, S0 = G::S0;
  constexpr int NF = 6 * NCB;                  // B fragments of a row
  constexpr int NIT = NCB / 2;                 // 16-byte store items per conv2 thread and row (256 threads, 8 items per pixel)
  long long st_busy = 0, st_vm = 0, st_bar = 0, st_n = 0;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;    // group 0: conv1, group 1: conv2; wq: the wave's block of 16 output channels
  const int r16 = lane & 15, kb = lane >> 4;

  // ---- this wave's 16 output channels of ITS conv: 18 A fragments in registers for the whole launch
  half8 wf[18];
#pragma unroll
  for (int f = 0; f < 18; ++f) wf[f] = *reinterpret_cast<const half8*>(p.w + ((((long)(grp * 4 + wq) * 18 + f) * 64 + lane) * 8));
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + 64 * grp + 16 * wq + 4 * kb);   // C/D rows 4 kb + i of this wave's block

  // ---- per-lane LDS offsets inside a ring row
  int foff[3][2];                              // B fragment (dx, channel chunk) of column block 0: pixel dx + (l & 15); block cb sits
#pragma unroll                                 // 16 cb pixels = 2048 cb B further (16 pixels do not change the swizzle term)
  for (int dx = 0; dx < 3; ++dx) {
    const int q = dx + r16, sw = swz(q);
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) foff[dx][kc] = q * 128 + (((4 * kc + kb) ^ sw) << 4);
  }
  // C/D layout (pixel l & 15, channels 16 wq + 4 kb ..+3) of column block 0: t / staging write, identity read (output column yi sits at
  // input pixel yi + 2); block cb again 2048 cb B further
  const int cdc = 2 * wq + (kb >> 1);
  const int doff = r16 * 128 + ((cdc ^ swz(r16)) << 4) + 8 * (kb & 1);
  const int roff = (r16 + 2) * 128 + ((cdc ^ swz(r16 + 2)) << 4) + 8 * (kb & 1);
  // DMA items (conv1 waves): piece j of a row covers pixels 8 j .. 8 j + 7 (lane: slot lane & 7 of pixel 8 j + (lane >> 3)).
  // NCB = 2: every step wave wq sends piece wq of the row PF steps ahead; piece 4 (pixels 32, 33) goes round the four waves.
  // NCB = 4: wave wq sends pieces wq and wq + 4, wave 0 also piece 8 (pixels 64, 65).
  constexpr int PLAST = G::PIECES - 1;         // the odd piece
  int soff_own, soff_hi = 0, soff_last;
  {
    const int q = 8 * wq + (lane >> 3), c = (lane & 7) ^ swz(q);
    soff_own = q * p.x_sp + c * 8;
    if constexpr (NCB == 4) {
      const int qh = 32 + q, ch = (lane & 7) ^ swz(qh);
      soff_hi = qh * p.x_sp + ch * 8;
    }
    const int ql = 8 * PLAST + (lane >> 3), cl = (lane & 7) ^ swz(ql);
    soff_last = ql * p.x_sp + cl * 8;
  }
  // store items (conv2 waves): item u of thread t = tid - 256 is (output column (t >> 3) + 32 u, slot t & 7)
  const int s_t = tid & 255;
  const int s_yi = s_t >> 3, s_c = (s_t & 7) ^ swz(s_yi);

  // ---- XCD-aware job walk: workgroup b sits on XCD b % 8; an XCD takes a contiguous range of jobs (neighbouring strips
  // of one row segment share their column halo in that XCD's L2)
  const int nwg = (int)gridDim.x, b = (int)blockIdx.x;
  int jfirst, jstep, jend;
  if ((nwg & 7) == 0) {
    const int per = (p.jobs + 7) >> 3;
    jfirst = (b & 7) * per + (b >> 3);
    jstep = nwg >> 3;
    jend = min(p.jobs, ((b & 7) + 1) * per);
  } else {
    jfirst = b; jstep = nwg; jend = p.jobs;
  }

  // activations as max(v, v * slope) in packed fp16 (slope 1: none, 0: ReLU), what the two-launch path computes
  const half_t hs1 = (half_t)p.slope1, hs2 = (half_t)p.slope2;
  const half4 sl1 = {hs1, hs1, hs1, hs1}, sl2 = {hs2, hs2, hs2, hs2};
  // The two roles are instantiated as the two branches of ONE wave-uniform `if` around the whole job loop: written as `if (grp == 0)`
  // inside a step, every per-lane value of either role (DMA source offsets and column flags, store items and residual rows) stays live
  // in both -- 20 to 60 spilled registers in the NCB = 4 kernel, and a spill is a scratch access the counted vmcnt knows nothing of.
  // Both roles execute the same sequence of barriers.
  auto run = [&](auto ROLEc) __attribute__((always_inline)) {
  constexpr int ROLE = decltype(ROLEc)::value;       // 0: conv1 wave, 1: conv2 wave
  f32x4 acc[3][NCB];                           // rotating accumulator rows (of t in group 0, of y in group 1)
  half8 fb[NF];                                // B fragments of the row being consumed
  half8 r2v[NIT] = {};

  for (int job = jfirst; job < jend; job += jstep) {
    const int n = job / (p.strips * p.segs);
    const int rem = job - n * (p.strips * p.segs);
    const int seg = rem / p.strips, strip = rem - seg * p.strips;
    const int c0 = strip * PW, ra = seg * p.seg_rows, rb = min(p.H, ra + p.seg_rows);
    const int rows = rb - ra;
    const half_t* xn = p.x + (long)n * p.x_sn;
    half_t* yn = p.y + (long)n * p.y_sn;

    bool col_own, col_hi = false, col_last;    // this lane's pixel of its pieces is inside the image (and one of the strip's)
    {
      const int q = 8 * wq + (lane >> 3), ql = 8 * PLAST + (lane >> 3);
      col_own = c0 - 2 + q >= 0 && c0 - 2 + q < p.W;
      if constexpr (NCB == 4) col_hi = c0 - 2 + 32 + q < p.W;
      col_last = ql < G::NPX && c0 - 2 + ql < p.W;
    }
    // t columns outside the image are the zero padding of conv2
    unsigned tmask[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int col = c0 - 1 + 16 * cb + r16;
      tmask[cb] = (col >= 0 && col < p.W) ? 0xFFFFFFFFu : 0u;
    }
    bool s_ok[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) s_ok[u] = s_yi + 32 * u < PW && c0 + s_yi + 32 * u < p.W && !(p.experiment & 2);
    // rows above the segment never finish a valid chain; start them from zero rather than from whatever the registers hold
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[s3][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // x row ra - 2 + kk into its ring slot: this wave's pieces (conv1 waves only); returns whether anything was issued
    auto issue_row = [&](int kk) __attribute__((always_inline)) -> int {
      const int row = ra - 2 + kk;
      const bool rowok = row >= 0 && row < p.H;
      const half_t* base = xn + ((long)row * p.W + (c0 - 2)) * p.x_sp;
      const unsigned dst = lds0 + X0 + pair_slot<XRING>(kk) * ROWB;
      glds16((rowok && col_own) ? base + soff_own : p.zeros, dst + wq * 1024);
      if constexpr (NCB == 4) {
        glds16((rowok && col_hi) ? base + soff_hi : p.zeros, dst + (wq + 4) * 1024);
        if (wq == 0) glds16((rowok && col_last) ? base + soff_last : p.zeros, dst + PLAST * 1024);
      } else {
        if (wq == (kk & 3)) glds16((rowok && col_last) ? base + soff_last : p.zeros, dst + PLAST * 1024);
      }
      return 1;
    };
    auto load_frags = [&](unsigned rowbase) __attribute__((always_inline)) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          const unsigned char* b0 = smem + (rowbase + foff[dx][kc]);      // one address per (dx, kc); the column block is an
#pragma unroll                                                           // immediate offset of the read
          for (int cb = 0; cb < NCB; ++cb) fb[(cb * 3 + dx) * 2 + kc] = *reinterpret_cast<const half8*>(b0 + cb * 2048);
        }
    };
    // the 18 NCB MFMAs of one row: fragment (cb, dx, kc) x taps (dy, dx); SN / SM / SD: accumulator rows that start here (dy 0,
    // bias as the C operand), continue (dy 1) and finish (dy 2).
    // NCB = 2: all 12 fragments are read up front; dy 2 is issued first so that the pack of the finished row (fin) overlaps the rest.
    // NCB = 4: 24 fragments next to 72 weight and 48 accumulator registers do not fit (65-100 registers spilled): the row goes by fragment
    //          GROUP (dx, kc) -- four fragments, 12 MFMAs -- with the next group's reads in flight under this group's MFMAs; the finished
    //          row is packed behind the last group.  Each accumulator still receives its (dx, kc) terms in the same order: the sums are
    //          bit-identical to the NCB = 2 kernel's.
    auto row_mfmas = [&](unsigned rowbase, auto SNc, auto SMc, auto SDc, auto fin, auto mid) __attribute__((always_inline)) {
      constexpr int SN = decltype(SNc)::value, SM = decltype(SMc)::value, SD = decltype(SDc)::value;
      if constexpr (NCB == 4) {
        half8 fg[PAIR_FBN][NCB];
        auto loadg = [&](int g, half8 (&dst)[NCB]) __attribute__((always_inline)) {
          const unsigned char* b0 = smem + (rowbase + foff[g >> 1][g & 1]);
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) dst[cb] = *reinterpret_cast<const half8*>(b0 + cb * 2048);
        };
#pragma unroll
        for (int g = 0; g < PAIR_FBN - 1; ++g) loadg(g, fg[g]);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
          const int dx = g >> 1, kc = g & 1;
          if (g + PAIR_FBN - 1 < 6) loadg(g + PAIR_FBN - 1, fg[(g + PAIR_FBN - 1) % PAIR_FBN]);
#pragma unroll
          for (int dyo = 0; dyo < 3; ++dyo) {
            const int dy = 2 - dyo;
            const int slot = dy == 2 ? SD : (dy == 1 ? SM : SN);
            const bool first = dy == 0 && g == 0;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
              acc[slot][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[(dy * 3 + dx) * 2 + kc], fg[g % PAIR_FBN][cb], first ? bias4 : acc[slot][cb], 0, 0, 0);
          }
          if (g == PAIR_MID_G) mid();
        }
        fin();
      } else {
        load_frags(rowbase);
#pragma unroll
        for (int dyo = 0; dyo < 3; ++dyo) {
          const int dy = 2 - dyo;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
              for (int cb = 0; cb < NCB; ++cb) {
                const int slot = dy == 2 ? SD : (dy == 1 ? SM : SN);
                const bool first = dy == 0 && dx == 0 && kc == 0;
                acc[slot][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[(dy * 3 + dx) * 2 + kc], fb[(cb * 3 + dx) * 2 + kc],
                                                                       first ? bias4 : acc[slot][cb], 0, 0, 0);
              }
          if (dy == 2) fin();
          if (dyo == 0) mid();
        }
      }
    };

    // ---- prologue: the first PF rows; everything landed and visible before step 0
    if constexpr (ROLE == 0) {
#pragma unroll
      for (int kk = 0; kk < PF; ++kk)
        if (kk <= rows + 3) issue_row(kk);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pair_barrier();

    const int K = (rows + 4 + 2 * BI + BI - 1) & ~(BI - 1);      // row steps, a whole number of barrier intervals
    // Before a step's barrier: everything this wave sent more than HD steps ago has landed (NCB = 2: row k + 4 is read from step
    // k + 4 on); the DMA of the last HD steps -- a constant number: conv1 waves issue no other vector-memory operation -- may stay in
    // flight.  Vector memory operations of a wave complete in order on this architecture (one counter for loads, stores and LDS-DMA).
    auto land_wait = [&](int nvm) __attribute__((always_inline)) {
      if (!nvm) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // issuing has stopped (segment tail)
      else if constexpr (NCB == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // 1 + 1 + 1 + 2 over any four issuing steps; shorter histories (first steps) have fewer in flight
      else if (wq == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                  // two steps x three pieces
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                               // two steps x two pieces
    };
    // One row step k.  conv1 waves consume x row i = ra - 2 + k: t rows i + 1 (started), i, i - 1 (finished, into the t ring).
    // conv2 waves consume t row i - 1 - BI: y rows i - BI (started), i - 1 - BI, i - 2 - BI (finished, + identity, into a staging
    // row); the y row finished BI row steps ago goes to memory.  The workgroup barrier closes every BI-th row step: each group
    // reads only rows that were finished before the last barrier, and the 2 BI-row t and staging rings hold exactly the BI
    // rows being written and the BI being read.  Between barriers the waves run free: the step time follows the average wave,
    // not the slowest one of every step (stamps without the barrier: 16 % fewer cycles per step).
    // FULL: the steady state (4 + 2 BI <= k <= rows + 2), where every part is active: straight-line code per group.
    auto step = [&](auto PHc, auto FULLc, int k) __attribute__((always_inline)) {
      constexpr int PH = decltype(PHc)::value;       // k % 3: which accumulator row is new / mid / done
      constexpr bool FULL = decltype(FULLc)::value;
      using S0c = std::integral_constant<int, PH>;
      using S1c = std::integral_constant<int, (PH + 1) % 3>;
      using S2c = std::integral_constant<int, (PH + 2) % 3>;
      const int i = ra - 2 + k;
      long long t0 = 0, t1 = 0, t2 = 0;
      if constexpr (STAMP && FULL) t0 = clock64();       // lgkmcnt is 0 here (pair_barrier): the read costs nothing
      if constexpr (ROLE == 0) {
        // ---- conv1 wave: t row new = slot (PH + 1) % 3, mid = PH, done = (PH + 2) % 3
        int nvm = 0;
        if (FULL || k <= rows + 3) {
          row_mfmas(X0 + pair_slot<XRING>(k) * ROWB, S1c{}, S0c{}, S2c{}, [&]() __attribute__((always_inline)) {
            // t row i - 1 -> fp16, activation, zero outside the image, into the t ring
            const int trow = i - 1;
            const unsigned rowm = (trow >= 0 && trow < p.H) ? 0xFFFFFFFFu : 0u;
            unsigned char* tb = smem + T0 + ((k - 1) & (TRING - 1)) * TROWB + doff;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
              const f32x4 v = acc[(PH + 2) % 3][cb];
              half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
              h = actk<A1>(h, sl1);
              typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
              u32x2 u = __builtin_bit_cast(u32x2, h);
              const unsigned m = rowm & tmask[cb];
              u[0] &= m; u[1] &= m;
              *reinterpret_cast<u32x2*>(tb + cb * 2048) = u;
            }
          }, [&]() __attribute__((always_inline)) {
            if (k + PF <= rows + 3 && !(p.experiment & 1)) nvm = issue_row(k + PF);
          });
        }
        if constexpr (STAMP && FULL) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          t1 = clock64();
        }
        land_wait(nvm);
      } else {
        // ---- conv2 wave: new = (PH + 2) % 3, mid = (PH + 1) % 3, done = PH
        const int srow = ra + k - 4 - 2 * BI;  // the y row finished BI row steps ago (before the last barrier): staging row -> memory
        if (FULL || (srow >= ra && srow < rb)) {
#pragma unroll
          for (int u = 0; u < NIT; ++u) {
            half8 yv = *reinterpret_cast<const half8*>(smem + S0 + ((k - BI) & (SRING - 1)) * SROW + (s_t + 256 * u) * 16);
            if constexpr (RES2) yv = yv + r2v[u];
            half_t* dst = s_ok[u] ? yn + ((long)srow * p.W + c0 + s_yi + 32 * u) * p.y_sp + s_c * 8 : p.dump + (s_t + 256 * u) * 8;
            *reinterpret_cast<half8*>(dst) = yv;
          }
        }
        if constexpr (RES2) {
          const int nrow = srow + 1;
#pragma unroll
          for (int u = 0; u < NIT; ++u) {
            const bool ok = s_ok[u] && (FULL || (nrow >= ra && nrow < rb));
            const half_t* src = ok ? p.res2 + (long)n * p.r2_sn + ((long)nrow * p.W + c0 + s_yi + 32 * u) * p.r2_sp + s_c * 8 : p.zeros;
            r2v[u] = *reinterpret_cast<const half8*>(src);
          }
        }
        if (FULL || (k >= BI + 2 && k <= rows + 3 + BI)) {
          // t row i - 1 - BI, finished before the last barrier
          row_mfmas(T0 + ((k - T_LAG) & (TRING - 1)) * TROWB, S2c{}, S1c{}, S0c{}, [&]() __attribute__((always_inline)) {
            // y row i - 2 - BI -> fp16, activation, + identity (that x row is still in the ring), into the staging row
            unsigned char* sb = smem + S0 + (k & (SRING - 1)) * SROW + doff;
            const unsigned char* xb = smem + X0 + pair_slot<XRING>(k - ID_LAG) * ROWB + roff;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
              const f32x4 v = acc[PH][cb];
              half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
              h = actk<A2>(h, sl2);
              if constexpr (ADDX) h = h + *reinterpret_cast<const half4*>(xb + cb * 2048);
              *reinterpret_cast<half4*>(sb + cb * 2048) = h;
            }
          }, []() {});
        }
        if constexpr (STAMP && FULL) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          t1 = clock64();
        }
      }
      if constexpr (STAMP && FULL) t2 = clock64();
      pair_barrier((k & (BI - 1)) != BI - 1 || (p.experiment & 4) != 0);     // one barrier per BI row steps
      if constexpr (STAMP && FULL) {
        const long long t3 = clock64();
        st_busy += t1 - t0; st_vm += t2 - t1; st_bar += t3 - t2; st_n += 1;
      }
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
    constexpr int KF = ((4 + 2 * BI + 2) / 3) * 3;      // first steady-state step: a multiple of 3 (accumulator rotation) >= 4 + 2 BI
    edge_steps(0, KF);
    int k = KF;
    for (; k + 2 <= rows + 2; k += 3) {
      step(I0{}, std::true_type{}, k);
      step(I1{}, std::true_type{}, k + 1);
      step(I2{}, std::true_type{}, k + 2);
    }
    edge_steps(k, K);
  }
  };
  if (grp == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
  if constexpr (STAMP) {
    if (lane == 0 && (int)blockIdx.x < stamp_cap) {
      long long* o = stamps + ((long)blockIdx.x * 8 + wave) * 8;
      o[0] = st_busy; o[1] = st_vm; o[2] = st_bar; o[3] = 0; o[4] = st_n;
    }
  }
}

}  // namespace

static bool g_pair_enabled = true;
static int g_pair_experiment = 0;
static int g_pair_geometry = 0;
static long long* g_pair_stamps = nullptr;
static int g_pair_stamp_cap = 0;
// diagnostics: [workgroup][wave][8] int64 = {busy, vmcnt wait, barrier wait, conv1 phase, steps} cycle sums over the steady-state steps
extern "C" void tdvc_debug_set_stamp_buffer_pair(void* buf, int cap_blocks) { g_pair_stamps = (long long*)buf; g_pair_stamp_cap = cap_blocks; }
extern "C" void tdvc_debug_set_pair_experiment(int e) { g_pair_experiment = e; }
// tests and A/B benchmarks: 2 / 4 = 30- / 62-column strips for every launch, 0 = chosen per width
extern "C" void tdvc_debug_set_pair_geometry(int ncb) { g_pair_geometry = ncb; }
// tests and A/B benchmarks: 0 makes tdvc_conv_pair_supported() answer no (callers then run the two convs separately)
extern "C" void tdvc_debug_enable_conv_pair(int enable) { g_pair_enabled = enable != 0; }

extern "C" int64_t tdvc_conv_pair_packed_bytes(void) { return 2L * 4 * 18 * 64 * 8 * 2; }

// Host-side packing: two fp32 [64][64][3][3] (OIHW) weights -> the per-wave A fragments of v_mfma_f32_16x16x32_f16:
// dst[conv][wave][tap * 2 + chunk][lane][j] = w[cout = 16 wave + (lane & 15)][cin = 32 chunk + 8 (lane >> 4) + j][tap]
extern "C" int tdvc_pack_conv_pair_weights(const float* w1_oihw, const float* w2_oihw, uint16_t* dst) {
  TDVC_CHECK(w1_oihw && w2_oihw && dst, "tdvc_pack_conv_pair_weights: null argument");
  for (int cv = 0; cv < 2; ++cv) {
    const float* w = cv ? w2_oihw : w1_oihw;
    for (int wv = 0; wv < 4; ++wv)
      for (int f = 0; f < 18; ++f)
        for (int ln = 0; ln < 64; ++ln)
          for (int j = 0; j < 8; ++j) {
            const int co = 16 * wv + (ln & 15), ci = 32 * (f & 1) + 8 * (ln >> 4) + j, t = f >> 1;
            const _Float16 h = (_Float16)w[((long)co * 64 + ci) * 9 + t];
            uint16_t bits;
            memcpy(&bits, &h, 2);
            dst[((((long)(cv * 4 + wv) * 18 + f) * 64) + ln) * 8 + j] = bits;
          }
  }
  return TDVC_OK;
}

// x rows are re-read as halo after neighbouring y rows were written: x and y may share a buffer only as disjoint channel
// windows of the same pixels (cat buffers); any other overlap of the two address ranges is refused
static bool pair_buffers_ok(const tdvc_fmap& x, const tdvc_fmap& y) {
  const uintptr_t xa = (uintptr_t)x.p, ya = (uintptr_t)y.p;
  const uintptr_t xe = xa + 2 * ((uintptr_t)(x.N - 1) * x.sn + ((uintptr_t)x.H * x.W - 1) * x.sp + 64);
  const uintptr_t ye = ya + 2 * ((uintptr_t)(y.N - 1) * y.sn + ((uintptr_t)y.H * y.W - 1) * y.sp + 64);
  if (!(xa < ye && ya < xe)) return true;
  const uintptr_t gap = xa > ya ? xa - ya : ya - xa;
  return x.sp == y.sp && x.sn == y.sn && gap >= 128 && gap + 128 <= 2 * (uintptr_t)x.sp;
}

extern "C" int tdvc_conv_pair_supported(const tdvc_conv_pair_desc* d) {
  static const bool off = getenv("TDVC_NO_CONV_PAIR") != nullptr;
  if (off || !g_pair_enabled || !d) return 0;
  const tdvc_fmap &x = d->x, &y = d->y;
  const bool ok = fmap_ok16(x) && fmap_ok16(y) && x.C == 64 && y.C == 64 && x.N == y.N && x.H == y.H && x.W == y.W && d->w && d->bias &&
                  (!d->res2.p || (fmap_ok16(d->res2) && d->res2.C == 64 && d->res2.N == x.N && d->res2.H == x.H && d->res2.W == x.W)) &&
                  (long)x.H * x.W >= 8192 && x.H >= 16 && (x.p == y.p /* geometry-only query: y not allocated yet */ || pair_buffers_ok(x, y)) &&
                  (d->act1 != TDVC_ACT_LRELU || (d->slope1 >= 0.f && d->slope1 <= 1.f)) && (d->act2 != TDVC_ACT_LRELU || (d->slope2 >= 0.f && d->slope2 <= 1.f));
  return ok ? 1 : 0;
}

extern "C" int tdvc_conv_pair(const tdvc_conv_pair_desc* d, void* stream) {
  TDVC_CHECK(d, "tdvc_conv_pair: null descriptor");
  TDVC_CHECK(fmap_ok16(d->x) && fmap_ok16(d->y) && d->x.C == 64 && d->y.C == 64, "tdvc_conv_pair: x and y must be fp16 maps of 64 channels");
  TDVC_CHECK(d->x.N == d->y.N && d->x.H == d->y.H && d->x.W == d->y.W, "tdvc_conv_pair: x / y geometry mismatch");
  TDVC_CHECK(d->w && d->bias, "tdvc_conv_pair: weights / bias missing");
  TDVC_CHECK(!d->res2.p || (fmap_ok16(d->res2) && d->res2.C == 64 && d->res2.N == d->x.N && d->res2.H == d->x.H && d->res2.W == d->x.W),
             "tdvc_conv_pair: res2 must be an fp16 map of the output geometry");
  TDVC_CHECK(pair_buffers_ok(d->x, d->y), "tdvc_conv_pair: x and y overlap (in-place operation is not supported: rows of x are re-read as halo)");
  TDVC_CHECK((d->act1 != TDVC_ACT_LRELU || (d->slope1 >= 0.f && d->slope1 <= 1.f)) && (d->act2 != TDVC_ACT_LRELU || (d->slope2 >= 0.f && d->slope2 <= 1.f)),
             "tdvc_conv_pair: LeakyReLU is computed as max(v, v * slope) and needs 0 <= slope <= 1");
  auto slope_of = [](int act, float slope) { return act == TDVC_ACT_NONE ? 1.f : (act == TDVC_ACT_RELU ? 0.f : slope); };
  TDVC_CHECK((d->act1 == TDVC_ACT_NONE || d->act1 == TDVC_ACT_RELU || d->act1 == TDVC_ACT_LRELU) &&
             (d->act2 == TDVC_ACT_NONE || d->act2 == TDVC_ACT_RELU || d->act2 == TDVC_ACT_LRELU), "tdvc_conv_pair: activations are none / ReLU / LeakyReLU");
  const void* zeros = nullptr;
  void* dump = nullptr;
  if (const int zrc = tdvc_scratch_pages(&zeros, &dump)) return zrc;
  PairParams p;
  p.x = reinterpret_cast<const half_t*>(d->x.p); p.x_sn = d->x.sn; p.x_sp = d->x.sp;
  p.y = reinterpret_cast<half_t*>(d->y.p); p.y_sn = d->y.sn; p.y_sp = d->y.sp;
  p.res2 = reinterpret_cast<const half_t*>(d->res2.p); p.r2_sn = d->res2.p ? d->res2.sn : 0; p.r2_sp = d->res2.p ? d->res2.sp : 0;
  p.w = reinterpret_cast<const half_t*>(d->w);
  p.bias = d->bias;
  p.zeros = reinterpret_cast<const half_t*>(zeros);
  p.experiment = g_pair_experiment;
  p.dump = reinterpret_cast<half_t*>(dump);
  p.N = d->x.N; p.H = d->x.H; p.W = d->x.W;
  p.slope1 = slope_of(d->act1, d->slope1);
  p.slope2 = slope_of(d->act2, d->slope2);
  // strip geometry: 62-column strips (NCB = 4) unless the narrow ones waste clearly fewer columns on this width (the wide steps are
  // worth ~1.12x: tools/bench_pair.py); tdvc_debug_set_pair_geometry() forces one
  auto waste = [&](int pw) { return (double)(((p.W + pw - 1) / pw) * (pw + 2)) / (double)p.W; };
  const int ncb = g_pair_geometry == 2 || g_pair_geometry == 4 ? g_pair_geometry : (waste(62) <= 1.12 * waste(30) ? 4 : 2);
  const int PWsel = 16 * ncb - 2;
  p.strips = (p.W + PWsel - 1) / PWsel;
  // row segments: fill 256 workgroups evenly; a segment pays 5 extra steps for its vertical halo
  const long base = (long)p.N * p.strips;
  int best = 1;
  double best_eff = 0.;
  for (int sg = 1; sg <= 64 && (p.H + sg - 1) / sg >= 16; ++sg) {
    const int sr = (p.H + sg - 1) / sg, nseg = (p.H + sr - 1) / sr;
    const long jobs = base * nseg;
    const double eff = (double)jobs / (double)(((jobs + 255) / 256) * 256) * sr / (sr + 5.0);
    if (eff > best_eff + 1e-9) { best_eff = eff; best = sg; }
  }
  p.seg_rows = (p.H + best - 1) / best;
  p.segs = (p.H + p.seg_rows - 1) / p.seg_rows;
  p.jobs = (int)(base * p.segs);
  const int grid = p.jobs < 256 ? p.jobs : 256;
  auto go = [&](auto kern, int lds) -> int {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) { tdvc_set_error("tdvc_conv_pair: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, reinterpret_cast<hipStream_t>(stream), p, g_pair_stamps, g_pair_stamp_cap);
    return 0;
  };
  // specialised: ReLU / none (Res_Block); everything else through the slope form
  const bool resblock = d->act1 == TDVC_ACT_RELU && d->act2 == TDVC_ACT_NONE;
  const int sel = (resblock ? 4 : 0) + (d->add_input ? 2 : 0) + (d->res2.p ? 1 : 0);
  auto launch = [&](auto ncbc) -> int {
    constexpr int NCB = decltype(ncbc)::value;
    constexpr int lds = PairGeo<NCB>::LDS;
    switch (sel) {
      case 0: return go(&conv_pair_kernel<NCB, 2, 2, false, false>, lds);
      case 1: return go(&conv_pair_kernel<NCB, 2, 2, false, true>, lds);
      case 2: return go(&conv_pair_kernel<NCB, 2, 2, true, false>, lds);
      case 3: return go(&conv_pair_kernel<NCB, 2, 2, true, true>, lds);
      case 4: return go(&conv_pair_kernel<NCB, 1, 0, false, false>, lds);
      case 5: return go(&conv_pair_kernel<NCB, 1, 0, false, true>, lds);
      case 6: return g_pair_stamps ? go(&conv_pair_kernel<NCB, 1, 0, true, false, true>, lds) : go(&conv_pair_kernel<NCB, 1, 0, true, false>, lds);
      default: return go(&conv_pair_kernel<NCB, 1, 0, true, true>, lds);
    }
  };
  const int rc = ncb == 4 ? launch(std::integral_constant<int, 4>{}) : launch(std::integral_constant<int, 2>{});
  if (rc) return rc;
  return tdvc_launch_status("tdvc_conv_pair");
}
