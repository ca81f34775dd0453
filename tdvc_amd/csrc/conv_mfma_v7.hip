// conv_mfma_v7 — the dominant layer class of the TDVC path: 3x3, stride 1, pad 1, Cin in {32, 64},
// Cout >= 64 (the 64->64 / 64->216 convs), transposed fp16 epilogue.
//
// In-kernel stamps of its register-staged predecessor showed the stage time is set inside
// the CU, not by HBM: staging the tile global -> VGPR -> ds_write costs two barriers per stage, a publish
// pass, address arithmetic per load, and the compiler drains every outstanding load (vmcnt(0)) around
// it.  v7 removes the staging:
//   * tiles arrive by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KB per instruction, no VGPR
//     destination), double-buffered: stage S+1 streams into the other buffer during stage S's matrix
//     phase and is retired by a COUNTED s_waitcnt vmcnt(N) + one raw s_barrier at the top of stage S+1;
//   * the DMA image is lane-linear (pixel-major, 64 B per pixel per 32-channel chunk); bank conflicts of
//     the B-fragment reads are removed by an XOR swizzle applied on the SOURCE address of the DMA
//     (16-byte slot c of pixel q holds channels 8*(c ^ ((q >> 2) & 3)) ..) and on the read address;
//   * one persistent 8-wave workgroup per CU, all weights of its 64 output channels resident in LDS
//     (72 KB), accumulators initialised from the bias, fragment reads software-pipelined one half tap ahead;
//   * one barrier per stage + one per tile (the finished tile buffer becomes the epilogue scratch).
// LDS map (bytes): [0, 40K) tile buffer 0 | [40K, 64K) weight slices 0..5 | [64K, 104K) tile buffer 1 |
// [104K, 152K) weight slices 6..17 | [152K, +512) bias, so that switching buffers is `addr ^ 0x10000`.
//
// The DMA is issued from inline asm (the compiler neither counts nor drains it); the only
// compiler-tracked vector-memory operations in the loop are the epilogue's residual / GDN loads and its
// stores, which are younger than the DMA they follow, so the compiler's own counted waits stay valid.
// The top-of-stage wait allows exactly the epilogue's 8 stores to stay in flight (full tiles) or drains.
#include <type_traits>

#include "conv_common.h"

using convk::ConvParams;

namespace {

constexpr int TH7 = 16, TW7 = 32, NT7 = 2, CK7 = 32, NTHR7 = 512;
constexpr int TIW7 = TW7 + 2, TIH7 = TH7 + 2, NPIX7 = TIW7 * TIH7;   // 34 x 18 = 612 halo pixels
constexpr int PIECES7 = (NPIX7 + 15) / 16;                           // 39 DMA pieces of 16 pixels x 64 B
constexpr int DMA7 = (PIECES7 + 7) / 8;                              // 5 per wave
constexpr int BUF1 = 0x10000, WLO7 = 40 * 1024, WHI7 = 104 * 1024, MISC7 = 152 * 1024;
constexpr int LDS7 = MISC7 + 512;
constexpr int WSL7 = 4096;
static_assert(DMA7 * 8 * 1024 <= WLO7, "tile buffer");
static_assert(8 * 32 * 144 <= WLO7, "epilogue scratch aliases a tile buffer");

struct V7Extra {
  int ntiles;
  const half_t* zeros;      // >= 16 bytes of zeros: the DMA source of out-of-image halo pixels
};

static long long* g_stamp7 = nullptr;
static int g_stamp7_cap = 0;

// weight slice sl (= chunk * 9 + tap) -> LDS byte offset
__device__ __host__ constexpr int wslice_off(int sl) { return sl < 6 ? WLO7 + sl * WSL7 : WHI7 + (sl - 6) * WSL7; }

__device__ __forceinline__ void glds16(const half_t* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void raw_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <bool STAMP = false>
__global__ __launch_bounds__(NTHR7, 1) void conv_mfma_v7_kernel(const ConvParams p, const V7Extra e, long long* stamps = nullptr, int stamp_cap = 0) {
  long long stv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ST7(i) do { if constexpr (STAMP) { if (S == 3) stv[i] = clock64(); } } while (0)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* bias_s = reinterpret_cast<float*>(smem + MISC7);   // 64 floats
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));   // LDS byte address of the array (low 32 bits of the flat address)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, r = lane & 31;
  const int cb = blockIdx.y, n = blockIdx.z;
  const int nchunks = p.nchunks;

  int first, stride, my_tiles;                             // XCD-aware: one contiguous band of tiles per L2
  convk::xcd_tile_walk(e.ntiles, first, stride, my_tiles);
  const int nstages = my_tiles * nchunks;
  if (nstages <= 0) return;

  // ---- per-lane DMA item: piece u = j * 8 + wave covers halo pixels q = 16u .. 16u + 15; this lane moves
  // 16-byte slot (lane & 3) of pixel q = 16u + (lane >> 2), which holds logical chunk c = slot ^ ((q >> 2) & 3).
  const int csw = (lane & 3) ^ ((lane >> 4) & 3);          // (q >> 2) & 3 == (lane >> 4) & 3 for every piece
  int it_off[DMA7];                                        // element offset from the tile's first halo pixel (interior tiles)
#pragma unroll
  for (int j = 0; j < DMA7; ++j) {
    const int q = 16 * (j * 8 + wave) + (lane >> 2);
    const int rr = q / TIW7, cc = q - rr * TIW7;
    it_off[j] = q < NPIX7 ? (rr * p.W + cc) * p.x_sp + csw * 8 : csw * 8;
  }
  const half_t* xn = p.x + (long)n * p.x_sn;

  int pf_iy0 = 0, pf_ix0 = 0, pf_ch = 0;
  bool pf_interior = false;
  const half_t* pf_base = xn;
  auto issue_prep = [&](int S) {
    const int tile_i = S / nchunks, ch = S - tile_i * nchunks;
    const int tile = p.reverse ? e.ntiles - 1 - (first + tile_i * stride) : first + tile_i * stride;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    pf_iy0 = ty * TH7 - 1;
    pf_ix0 = tx * TW7 - 1;
    pf_ch = ch;
    pf_interior = pf_iy0 >= 0 && pf_ix0 >= 0 && pf_iy0 + TIH7 <= p.H && pf_ix0 + TIW7 <= p.W;
    pf_base = xn + ((long)pf_iy0 * p.W + pf_ix0) * p.x_sp + ch * CK7;     // only dereferenced when interior
  };
  auto issue_one = [&](int j, unsigned dst_buf) {          // j is a compile-time constant at every call site
    const half_t* src = pf_base + it_off[j];
    if (!pf_interior) {                  // uniform branch: border tiles (6 % at 1080p) clamp per lane
      const int q = 16 * (j * 8 + wave) + (lane >> 2);
      const int rr = q / TIW7;
      const int iy = pf_iy0 + rr, ix = pf_ix0 + (q - rr * TIW7);
      const bool ok = q < NPIX7 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      src = ok ? xn + ((long)iy * p.W + ix) * p.x_sp + pf_ch * CK7 + csw * 8 : e.zeros;
    }
    glds16(src, dst_buf + (j * 8 + wave) * 1024);
  };

  // ---- prologue: first tile's DMA, then bias + resident weights (compiler-tracked loads, younger than the DMA)
  issue_prep(0);
#pragma unroll
  for (int j = 0; j < DMA7; ++j) issue_one(j, lds0);

  if (tid < 64) bias_s[tid] = p.bias[blockIdx.y * 64 + tid];
  {
    const int nslices = nchunks * 9;                 // slice (ch, t) = 4 KB [mt 2][s2 2][lane 64][8 halves]
    for (int i = tid; i < nslices * 256; i += NTHR7) {
      const int sl = i >> 8, u = i & 255;            // u = q*64 + lane, q = mt*2 + s2
      const int qq = u >> 6, ln = u & 63;
      const half_t* src = p.w + ((((long)(cb * 2 + (qq >> 1)) * nslices + sl) * 2 + (qq & 1)) * 64 + ln) * 8;
      *reinterpret_cast<half8*>(smem + wslice_off(sl) + u * 16) = *reinterpret_cast<const half8*>(src);
    }
  }

  // ---- B-fragment read offsets (tile buffer 0, s2 = 0): pixel q = (2*wave + nt + dy) * 34 + r + dx, chunk hh
  int bq[NT7][9];
#pragma unroll
  for (int nt = 0; nt < NT7; ++nt)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int q = (wave * NT7 + nt + t / 3) * TIW7 + r + (t % 3);
      bq[nt][t] = q * 64 + ((hh ^ ((q >> 2) & 3)) << 4);
    }

  __syncthreads();                       // bias and weights are visible (no DMA-aware wait here: see top of stage)

  f32x16 acc[2][NT7];
  auto init_acc = [&]() {                // accumulators start from the bias: LDS reads, no vector moves
#pragma unroll
    for (int nt = 0; nt < NT7; ++nt) {
      int o = hh * 4;
      asm volatile("" : "+v"(o));        // keep one read per accumulator (no register copies)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + mt * 32 + 8 * g + o);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[mt][nt][4 * g + i] = b4[i];
        }
    }
  };
  init_acc();

  bool stores_in_flight = false;         // the previous stage ended with exactly 8 epilogue stores (full tile)
  for (int S = 0; S < nstages; ++S) {
    const int tile_i = S / nchunks, ch = S - tile_i * nchunks;
    const unsigned bsel = (S & 1) ? BUF1 : 0;
    ST7(0);
    // This wave's DMA pieces of stage S have landed: they are older than the (at most 8) epilogue stores.
    if (stores_in_flight) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ST7(1);
    raw_barrier();                       // every wave's pieces landed; every wave is done with the other buffer
    ST7(2);
    const bool have_next = S + 1 < nstages;
    if (have_next) issue_prep(S + 1);
    const unsigned nbuf = lds0 + (bsel ^ BUF1);

    // matrix phase: 18 half taps, fragments double-buffered; one DMA piece of the next stage after each of
    // the first DMA7 half taps
    const unsigned char* tb = smem + bsel;
    const unsigned char* wlo = smem + (ch == 0 ? WLO7 : wslice_off(9) - 0 * WSL7) + lane * 16;              // taps 0..5 of this chunk
    const unsigned char* whi = smem + (ch == 0 ? wslice_off(6) - 6 * WSL7 : wslice_off(9)) + lane * 16;     // taps 6..8
    half8 fa[2][2], fb[2][NT7];
    auto frag = [&](int i, int buf) {
      const int t = i >> 1, s2 = i & 1;
      const unsigned char* wt = (t < 6 ? wlo : whi) + t * WSL7;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) fa[buf][mt] = *reinterpret_cast<const half8*>(wt + (mt * 2 + s2) * 1024);
#pragma unroll
      for (int nt = 0; nt < NT7; ++nt) fb[buf][nt] = *reinterpret_cast<const half8*>(tb + (bq[nt][t] ^ (s2 * 32)));
    };
    frag(0, 0);
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      if (i + 1 < 18) frag(i + 1, (i + 1) & 1);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT7; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i & 1][mt], fb[i & 1][nt], acc[mt][nt], 0, 0, 0);
      // pin the software pipeline: all four fragment reads of half tap i+1 are issued BEFORE the four MFMAs
      // of half tap i (left alone, hipcc sinks them behind the second MFMA and the next group stalls on them)
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      if (i < DMA7 && have_next) issue_one(i, nbuf);
    }
    ST7(3);
    stores_in_flight = false;
    if (ch != nchunks - 1) continue;

    const int tile = p.reverse ? e.ntiles - 1 - (first + tile_i * stride) : first + tile_i * stride;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const bool full = (ty + 1) * TH7 <= p.Ho && (tx + 1) * TW7 <= p.Wo;
    raw_barrier();                       // all waves finished reading this tile buffer: it becomes epilogue scratch
    ST7(4);
    convk::epilogue_simple_rows<NT7, true, 1>(p, acc, bias_s, smem + bsel + wave * (32 * 144), n, cb * 64,
                                           ty * TH7 + wave * NT7, tx * TW7, lane, false, full);
    init_acc();
    stores_in_flight = full;
    ST7(5);
    if constexpr (STAMP) {
      if (S == 3 && lane == 0) {             // one record per wave: [block][wave][8 stamps]
        const int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (bid * 8 + 7 < stamp_cap) for (int i = 0; i < 8; ++i) stamps[((long)bid * 8 + wave) * 8 + i] = stv[i];
      }
    }
  }
}

}  // namespace

extern "C" void tdvc_debug_set_stamp_buffer_v7(void* buf, int cap_blocks) { g_stamp7 = (long long*)buf; g_stamp7_cap = cap_blocks; }

bool conv_v7_eligible(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo) {
  static const bool off = getenv("TDVC_CONV_NO_V7") != nullptr || getenv("TDVC_CONV_V1") != nullptr;
  if (off) return false;
  bool taps33 = d->ntaps == 9 && d->kh == 3 && d->kw == 3 && d->pad == 1;
  for (int t = 0; taps33 && t < 9; ++t) taps33 = d->tap_dy[t] == t / 3 && d->tap_dx[t] == t % 3;
  // the counted vmcnt(8) of a full tile assumes every wave stored (see conv_v11_eligible); here a wave covers the 64 packed
  // rows of its cout block (sub-pixel store: inside one sub-pixel, cq % 64 == 0 by conv_is_simple)
  const bool all_waves_store = d->y.C > (p.out_mode == TDVC_OUT_SHUFFLE2 ? (d->cout >> 2) - 64 : ((d->cout + 63) / 64 - 1) * 64);
  return taps33 && all_waves_store && d->ck == 32 && d->stride == 1 && d->cout >= 64 && (d->x.C == 32 || d->x.C == 64) && !d->s2d &&
         !d->square_input && (long)Ho * Wo >= 8192 && convk::conv_is_simple(p);
}

int launch_conv_v7(const ConvParams& p, int cout_blocks, int N, hipStream_t st) {
  const void* zeros = nullptr;
  if (const int zrc = tdvc_scratch_pages(&zeros, nullptr)) return zrc;
  ConvParams q = p;
  q.tiles_x = (p.Wo + TW7 - 1) / TW7;
  const int tiles_y = (p.Ho + TH7 - 1) / TH7;
  V7Extra e;
  e.ntiles = q.tiles_x * tiles_y;
  e.zeros = reinterpret_cast<const half_t*>(zeros);
  q.slope = convk::conv_simple_slope(p);
  int gx = 256 / (cout_blocks * N);
  if (gx < 1) gx = 1;
  if (gx > e.ntiles) gx = e.ntiles;
  dim3 grid(gx, cout_blocks, N);
  auto go = [&](auto kern, bool stamp) -> int {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) { tdvc_set_error("conv v7: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    hipLaunchKernelGGL(kern, grid, dim3(NTHR7), LDS7, st, q, e, stamp ? g_stamp7 : (long long*)nullptr, stamp ? g_stamp7_cap : 0);
    return 0;
  };
  const int rc = g_stamp7 ? go(&conv_mfma_v7_kernel<true>, true) : go(&conv_mfma_v7_kernel<false>, false);
  if (rc) return rc;
  return tdvc_launch_status("tdvc_conv2d(v7)");
}
