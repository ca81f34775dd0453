// Modulated deformable convolution (DCNv2) for gfx950 — the motion-compensation operator
// (SURVEY §8 a11/a12).
//
// (1) tdvc_dcn_fused: fp16 NHWC, sampling + modulation + 64x576 contraction in ONE kernel.
//     The reference materialises `columns` in HBM (576 values / pixel = 4.8 GB at 1080p,
//     src/cuda/dcn_v2_cuda.cu:67-92); here the modulated samples of an 8x8 pixel tile live in a
//     25 KB LDS tile, three taps at a time, and are consumed as MFMA B fragments.
// (2) tdvc_dcn_v2_forward_f32: fp32 NCHW operator with `_ext.dcn_v2_forward` semantics.
#include "common.h"

#include <type_traits>

namespace {

struct DcnParams {
  const half_t* x; long x_sn; int x_sp; int H, W;
  const half_t* om; long om_sn; int om_sp;
  FMap y;
  const half_t* w; const float* bias;
  int G; int act; float slope; int round16;
  long npix;   // H*W
  const half_t* xp;   // group-planar copy of x ([N][G][H][W][8]) or null
};

// x (NHWC, 8G channels) -> [n][g][H][W][8]: 8 lanes read one pixel's 128-byte line, every group plane receives 128
// contiguous bytes per 8 pixels
__global__ void dcn_planarise_kernel(const half_t* x, long x_sn, int x_sp, long npix, int G, half_t* xp) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (i >= npix * G) return;
  const int g = (int)(i % G);
  const long pix = i / G;
  const half8 v = *reinterpret_cast<const half8*>(x + (long)n * x_sn + pix * x_sp + g * 8);
  *reinterpret_cast<half8*>(xp + (((long)n * G + g) * npix + pix) * 8) = v;
}

// Fused kernel, 3x3 / stride 1 / pad 1 / dilation 1, 8 channels per deformable group.
// One 256-thread workgroup = an 8x8 pixel tile x all 8G output channels.
//   sampling phase: thread (pixel = tid>>3 [+32], group = tid&7) -> the 8 lanes of a pixel read
//     one full 128-byte NHWC line per bilinear corner (coalesced), branch-free (clamped address,
//     zeroed weight) so all corner loads of a tap are in flight together; modulated samples go
//     to an LDS column tile of 3 taps x 8G channels per pixel (stride +16 B: conflict-free).
//   MFMA phase: wave (mt, nt) multiplies the 32-row weight tile mt with the 32-pixel half nt,
//     B fragments = 16-byte LDS reads (lane = pixel, k-chunk = group), 3 taps per barrier pair.
constexpr int DCN_TPX = 8, DCN_TPY = 8, DCN_TAPS_PER_CHUNK = 3;

__device__ __forceinline__ half8 sample8_bf(const half_t* xg, int H, int W, int sp, float h_im, float w_im, float mask) {
  // dcn_v2_im2col_cuda.cu:25-54 (bilinear, zero outside) and :180 (open-interval test), branch-free
  const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
  const float hc = fminf(fmaxf(h_im, -2.f), (float)H + 1.f), wc = fminf(fmaxf(w_im, -2.f), (float)W + 1.f);
  const float hf = floorf(hc), wf = floorf(wc);
  const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
  const float lh = hc - hf, lw = wc - wf, hh = 1.f - lh, hw = 1.f - lw;
  const bool hl = h_low >= 0 && h_low <= H - 1, hhg = h_high >= 0 && h_high <= H - 1;
  const bool wl = w_low >= 0 && w_low <= W - 1, whg = w_high >= 0 && w_high <= W - 1;
  // corner weights with the modulation mask folded in; the 8-channel combine is 4 fused multiply-adds per
  // channel straight from the fp16 samples (v_fma_mix_f32), ~40 vector instructions instead of ~100
  const float w1 = (inside && hl && wl) ? hh * hw * mask : 0.f, w2 = (inside && hl && whg) ? hh * lw * mask : 0.f;
  const float w3 = (inside && hhg && wl) ? lh * hw * mask : 0.f, w4 = (inside && hhg && whg) ? lh * lw * mask : 0.f;
  const int y0 = min(max(h_low, 0), H - 1), y1 = min(max(h_high, 0), H - 1);
  const int x0 = min(max(w_low, 0), W - 1), x1 = min(max(w_high, 0), W - 1);
  const int r0 = y0 * W, r1 = y1 * W;                     // pixel indices fit 32 bits (checked by the host)
  const half8 v1 = *reinterpret_cast<const half8*>(xg + (long)(r0 + x0) * sp);
  const half8 v2 = *reinterpret_cast<const half8*>(xg + (long)(r0 + x1) * sp);
  const half8 v3 = *reinterpret_cast<const half8*>(xg + (long)(r1 + x0) * sp);
  const half8 v4 = *reinterpret_cast<const half8*>(xg + (long)(r1 + x1) * sp);
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    o[j] = (half_t)__builtin_fmaf(w1, (float)v1[j], __builtin_fmaf(w2, (float)v2[j], __builtin_fmaf(w3, (float)v3[j], w4 * (float)v4[j])));
  return o;
}

__global__ __launch_bounds__(256) void dcn_fused_kernel(const DcnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char col[];
  const int G = p.G;
  const int PS = DCN_TAPS_PER_CHUNK * G * 16 + 16;       // LDS bytes per pixel (3 taps x 8G halves + pad)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, r = lane & 31;
  const int mt = wave & 1, nt = wave >> 1;
  const int tiles_x = (p.W + DCN_TPX - 1) / DCN_TPX;
  const int tile_id = tdvc_xcd_tile(blockIdx.x);
  const int tx = tile_id % tiles_x, ty = tile_id / tiles_x;
  const int n = blockIdx.y;
  const half_t* xn = p.x + (long)n * p.x_sn;

  // ---- sampling identity: two pixels per thread (pass 0: pixels 0..31, pass 1: 32..63)
  const int g = tid & 7;
  const bool g_on = g < G;
  const int gg = g_on ? g : 0;
  // gather source of this thread's group: the NHWC map (pixel stride x_sp: every 16-byte corner of every lane is a line
  // of its own -- the L1 takes one line per clock, profiles/r01_dcn_fused_1080p_pmc.txt) or the group-planar copy
  // [n][g][H][W][8] (pixel stride 8: the 8 lanes of a group sample 8 neighbouring pixels, whose corners share lines)
  const half_t* xg = p.xp ? p.xp + ((long)n * G + gg) * p.npix * 8 : xn + gg * 8;
  const int xsp = p.xp ? 8 : p.x_sp;
  int soy[2], sox[2];
  half2v off[2][9];
  float msk[2][9];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int pl = (tid >> 3) + 32 * ps;
    soy[ps] = ty * DCN_TPY + (pl >> 3);
    sox[ps] = tx * DCN_TPX + (pl & 7);
    const int cy = min(soy[ps], p.H - 1), cx = min(sox[ps], p.W - 1);
    const half_t* omp = p.om + (long)n * p.om_sn + ((long)cy * p.W + cx) * p.om_sp;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      off[ps][t] = *reinterpret_cast<const half2v*>(omp + gg * 18 + 2 * t);
      msk[ps][t] = __builtin_amdgcn_rcpf(1.f + __expf(-(float)omp[18 * G + gg * 9 + t]));    // sigmoid (dcn_v2_amp.py: mask = sigmoid(mask))
    }
  }

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int steps_per_tap = G / 2;
  const half_t* wbase = p.w + ((long)mt * 9 * steps_per_tap * 64 + lane) * 8;
  const unsigned char* bcol = col + (nt * 32 + r) * PS + hh * 16;

#pragma unroll
  for (int chunk = 0; chunk < 3; ++chunk) {             // fully unrolled: tap registers are selected at compile time
    if (chunk > 0) __syncthreads();
    if (g_on) {
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
        for (int tc = 0; tc < DCN_TAPS_PER_CHUNK; ++tc) {
          const half2v o2 = off[ps][chunk * 3 + tc];
          const float h_im = (float)(soy[ps] - 1 + chunk) + (float)o2[0];      // tap = chunk*3 + tc: dy = chunk, dx = tc
          const float w_im = (float)(sox[ps] - 1 + tc) + (float)o2[1];
          const half8 v = sample8_bf(xg, p.H, p.W, xsp, h_im, w_im, msk[ps][chunk * 3 + tc]);
          const int pl = (tid >> 3) + 32 * ps;
          *reinterpret_cast<half8*>(col + pl * PS + tc * G * 16 + gg * 16) = v;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int tc = 0; tc < DCN_TAPS_PER_CHUNK; ++tc) {
      const int tap = chunk * 3 + tc;
      for (int s2 = 0; s2 < steps_per_tap; ++s2) {
        const half8 a = *reinterpret_cast<const half8*>(wbase + (long)(tap * steps_per_tap + s2) * 512);
        const half8 b = *reinterpret_cast<const half8*>(bcol + tc * G * 16 + s2 * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      }
    }
  }

  const int pl = nt * 32 + r;
  const int oy = ty * DCN_TPY + (pl >> 3), ox = tx * DCN_TPX + (pl & 7);
  if (oy >= p.H || ox >= p.W) return;
  half_t* yp = reinterpret_cast<half_t*>(p.y.p) + (long)n * p.y.sn + ((long)oy * p.W + ox) * p.y.sp;
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int co = mt * 32 + 8 * gq + 4 * hh;
    if (co >= p.y.C) continue;
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + co);
    half4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = acc[4 * gq + i] + b4[i];
      if (p.round16) v = (float)(half_t)v;
      v = act_apply(v, p.act, p.slope);
      o[i] = (half_t)v;
    }
    *reinterpret_cast<half4*>(yp + co) = o;
  }
}

// ------------------------------------------------------------------------------------------
// dcn_lds: the same operator with the bilinear corners gathered from an LDS window instead of L1.
//
// dcn_fused_kernel is bound by the texture path: 4 corners x 9 taps x 8 groups of 16 bytes per pixel, every one a cache
// line of its own for the L1 (profiles/r01_dcn_fused_1080p_pmc.txt), behind two barriers per three taps.  Here
//   * a workgroup owns an 8 x 16 pixel tile and stages the (8+12) x (16+12) window of x around it in LDS once (80 KB for
//     the 64-channel map, row pitch 32 pixels, 16-byte group slots XOR-swizzled by the column so that the 16 lanes of one
//     ds_read_b128 phase -- neighbouring pixels, same group -- fall on 16 different slots);
//   * the sampling lane IS the MFMA B lane: wave w owns rows 2w, 2w+1 of the tile (32 pixels = the 32 columns of a
//     32x32x16 product), lane (r, hh) samples group 2*s2 + hh of pixel r at k-step (tap, s2), i.e. exactly its 8 k-values
//     of the B fragment: no column tile, no barrier between sampling and the matrix product;
//   * offsets / masks of the tile travel through the same LDS region first (coalesced 16-byte loads of the 432-byte
//     per-pixel record), each lane keeps its 4 groups x 9 taps in registers;
//   * a sample whose corners leave the window (more than 5 px from the tile's mean displacement) falls back to the global gather, per lane; a wave whose
//     lanes are all inside (the common case) takes a branch without it.
// Arithmetic and accumulation order are those of dcn_fused_kernel: the two kernels agree bit for bit
// (tests/test_dcn_gpu.py).
// ------------------------------------------------------------------------------------------
constexpr int DL_TY = 8, DL_TX = 16, DL_M = 6;
constexpr int DL_WH = DL_TY + 2 * DL_M, DL_WW = DL_TX + 2 * DL_M, DL_WP = 32;      // window rows, columns, row pitch (pixels)
constexpr int DL_LDS = DL_WH * DL_WP * 128;                                           // 81,920 B: two workgroups per CU
constexpr int DL_OM_REC = 27 * 8 * 2;                                                 // 432 B of offsets + masks per pixel
constexpr int DL_RED = ((DL_WH - 1) * DL_WP + DL_WW) * 128;                           // pitch padding of the last row: never a window pixel
static_assert(DL_TY * DL_TX * DL_OM_REC <= DL_RED, "the offset/mask tile is staged through the window region, below the reduction scratch");
static_assert(DL_WP >= DL_WW + 1 && DL_WP % 16 == 0, "row pitch: a multiple of 16 pixels keeps the two rows of a wave on disjoint slots");

struct DlSample {            // one (pixel, group, tap) sample in flight: corner weights, LDS addresses, image coordinates
  float w1, w2, w3, w4;
  int s00, s01, s10, s11;    // LDS byte offsets of the four corners (clamped into the window)
  int y0, y1, x0, x1;        // image coordinates (for the fallback gather)
  bool iw;                   // all four corners inside the window
};

__device__ __forceinline__ DlSample dl_prepare(int wy0, int wx0, int g, int H, int W, float h_im, float w_im, float mask) {
  // identical arithmetic to sample8_bf; only the source of the four corner vectors differs
  DlSample q;
  const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
  const float hc = fminf(fmaxf(h_im, -2.f), (float)H + 1.f), wc = fminf(fmaxf(w_im, -2.f), (float)W + 1.f);
  const float hf = floorf(hc), wf = floorf(wc);
  const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
  const float lh = hc - hf, lw = wc - wf, hh = 1.f - lh, hw = 1.f - lw;
  const bool hl = h_low >= 0 && h_low <= H - 1, hhg = h_high >= 0 && h_high <= H - 1;
  const bool wl = w_low >= 0 && w_low <= W - 1, whg = w_high >= 0 && w_high <= W - 1;
  q.w1 = (inside && hl && wl) ? hh * hw * mask : 0.f; q.w2 = (inside && hl && whg) ? hh * lw * mask : 0.f;
  q.w3 = (inside && hhg && wl) ? lh * hw * mask : 0.f; q.w4 = (inside && hhg && whg) ? lh * lw * mask : 0.f;
  q.y0 = min(max(h_low, 0), H - 1); q.y1 = min(max(h_high, 0), H - 1);
  q.x0 = min(max(w_low, 0), W - 1); q.x1 = min(max(w_high, 0), W - 1);
  const int a0 = q.y0 - wy0, a1 = q.y1 - wy0, b0 = q.x0 - wx0, b1 = q.x1 - wx0;  // window coordinates (a1 >= a0, b1 >= b0)
  q.iw = a0 >= 0 && a1 < DL_WH && b0 >= 0 && b1 < DL_WW;
  const int ca0 = q.iw ? a0 : 0, ca1 = q.iw ? a1 : 0, cb0 = q.iw ? b0 : 0, cb1 = q.iw ? b1 : 0;
  const int r0 = ca0 * (DL_WP * 128), r1 = ca1 * (DL_WP * 128);
  const int c0 = cb0 * 128 + ((g ^ ((cb0 >> 1) & 7)) << 4), c1 = cb1 * 128 + ((g ^ ((cb1 >> 1) & 7)) << 4);
  q.s00 = r0 + c0; q.s01 = r0 + c1; q.s10 = r1 + c0; q.s11 = r1 + c1;
  return q;
}

// The same for a tile whose whole window lies inside the image: a sample inside the window needs none of the image-border
// logic (every corner exists, the clamps are identities), so the common path is floor / fractions / four products and the
// addresses.  The window test is done on the floats (NaN fails it); a lane that fails is recomputed by dl_prepare.
__device__ __forceinline__ DlSample dl_prepare_interior(int wy0, int wx0, int g, float h_im, float w_im, float mask) {
  DlSample q;
  const float hf = floorf(h_im), wf = floorf(w_im);
  q.iw = (hf >= (float)wy0) & (hf <= (float)(wy0 + DL_WH - 2)) & (wf >= (float)wx0) & (wf <= (float)(wx0 + DL_WW - 2));
  const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
  q.w1 = hh * hw * mask; q.w2 = hh * lw * mask; q.w3 = lh * hw * mask; q.w4 = lh * lw * mask;
  const int a0 = q.iw ? (int)hf - wy0 : 0, b0 = q.iw ? (int)wf - wx0 : 0, b1 = b0 + 1;
  const int r0 = a0 * (DL_WP * 128);
  const int c0 = b0 * 128 + ((g ^ ((b0 >> 1) & 7)) << 4), c1 = b1 * 128 + ((g ^ ((b1 >> 1) & 7)) << 4);
  q.s00 = r0 + c0; q.s01 = r0 + c1; q.s10 = q.s00 + DL_WP * 128; q.s11 = q.s01 + DL_WP * 128;
  q.y0 = q.y1 = q.x0 = q.x1 = 0;
  return q;
}

// MODE 0: the operator.  MODE 1 / 2: timing-only builds for tools/ab_dcn.py (1: no sampling / matrix phase, 2: no window
// staging -- the window holds garbage, control flow and instruction stream unchanged); their output is meaningless.
template <int MODE>
__global__ __launch_bounds__(256, 2) void dcn_lds_kernel(const DcnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char win[];
  const int tid0 = threadIdx.x, lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int hh = lane >> 5, r = lane & 31;
  const int tiles_x = (p.W + DL_TX - 1) / DL_TX;
  const int tile_id = tdvc_xcd_tile(blockIdx.x);
  const int tx = tile_id % tiles_x, ty = tile_id / tiles_x;
  const int n = blockIdx.y;
  const half_t* xn = p.x + (long)n * p.x_sn;
  const int tid = tid0;

  // ---- phase 1: the tile's offset / mask records -> LDS -> registers
  // Staging indices are chosen so that everything that varies from piece to piece is wave-uniform (scalar ALU, immediate LDS
  // offsets): thread = (piece slot tid & 31 of the 27-piece record, pixel tid >> 5 of a group of 8); 16 groups of 8 pixels =
  // the tile, group k = row k >> 1, columns 8 (k & 1) .. + 7.  (The first version divided a flat piece index by 27 and by 28:
  // ~740 of the kernel's ~4 300 vector instructions per wave and tile.  Removing them moved the kernel from 724 to 716 us: it
  // is not bound by the instruction count.)  All 16 loads of a thread are in flight together.
  {
    const half_t* omn = p.om + (long)n * p.om_sn;
    const int ps = tid & 31, pj = tid >> 5;
    const bool pv = ps < 27;
    const int pcl = pv ? ps : 26;
    half8 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int cy = min(ty * DL_TY + (k >> 1), p.H - 1);                      // uniform
      const int cx = min(tx * DL_TX + 8 * (k & 1) + pj, p.W - 1);
      v[k] = *reinterpret_cast<const half8*>(omn + (long)cy * p.W * p.om_sp + (cx * p.om_sp + pcl * 8));
    }
    unsigned char* dst = win + pj * DL_OM_REC + pcl * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (pv) *reinterpret_cast<half8*>(dst + k * 8 * DL_OM_REC) = v[k];
  }
  __syncthreads();
  const int pl = (2 * wave + (r >> 4)) * DL_TX + (r & 15);
  const int oy = ty * DL_TY + 2 * wave + (r >> 4), ox = tx * DL_TX + (r & 15);
  half2v off[4][9];
  float msk[4][9];
  float sdy = 0.f, sdx = 0.f;
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) {
    const int g = 2 * s2 + hh;
    const unsigned char* rec = win + pl * DL_OM_REC;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      off[s2][t] = *reinterpret_cast<const half2v*>(rec + g * 36 + 4 * t);
      const half_t m = *reinterpret_cast<const half_t*>(rec + 288 + g * 18 + 2 * t);
      msk[s2][t] = __builtin_amdgcn_rcpf(1.f + __expf(-(float)m));
      sdy += __builtin_amdgcn_fmed3f((float)off[s2][t][0], -64.f, 64.f);      // a stray wild offset must not move the window
      sdx += __builtin_amdgcn_fmed3f((float)off[s2][t][1], -64.f, 64.f);
    }
  }
  // the window is centred on the tile's mean displacement (coherent motion moves the window, not the samples out of it);
  // the choice only decides which samples take the fallback gather, never a value
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) {
    sdy += __shfl_xor(sdy, sh);
    sdx += __shfl_xor(sdx, sh);
  }
  float* red = reinterpret_cast<float*>(win + DL_RED);
  if (lane == 0) { red[2 * wave] = sdy; red[2 * wave + 1] = sdx; }
  __syncthreads();
  const float inv = 1.f / (float)(DL_TY * DL_TX * 72);
  const int cy = __builtin_amdgcn_readfirstlane((int)rintf(fminf(fmaxf((red[0] + red[2] + red[4] + red[6]) * inv, -4096.f), 4096.f)));
  const int cx = __builtin_amdgcn_readfirstlane((int)rintf(fminf(fmaxf((red[1] + red[3] + red[5] + red[7]) * inv, -4096.f), 4096.f)));

  // ---- phase 2: the window of x -> LDS (swizzled)
  const int wy0 = ty * DL_TY - DL_M + cy, wx0 = tx * DL_TX - DL_M + cx;
  {
    // thread = (channel chunk tid & 7, column slot tid >> 3 of 32, 28 used); one window row per load, the row a scalar
    const int c = tid & 7, wx = tid >> 3;
    const bool wv = wx < DL_WW;
    const int ix = min(max(wx0 + (wv ? wx : 0), 0), p.W - 1);                  // outside the image: never sampled
    const int lane_off = ix * p.x_sp + c * 8;
    half8 v[DL_WH];
#pragma unroll
    for (int wy = 0; wy < DL_WH; ++wy) {
      const int iy = min(max(wy0 + wy, 0), p.H - 1);                           // uniform
      if constexpr (MODE != 2) v[wy] = *reinterpret_cast<const half8*>(xn + (long)iy * p.W * p.x_sp + lane_off);
      else v[wy] = half8{(half_t)(float)iy, (half_t)(float)ix, 0, 0, 0, 0, 0, 0};
    }
    unsigned char* dst = win + (wv ? wx : 0) * 128 + ((c ^ ((wx >> 1) & 7)) << 4);
#pragma unroll
    for (int wy = 0; wy < DL_WH; ++wy)
      if (wv) *reinterpret_cast<half8*>(dst + wy * DL_WP * 128) = v[wy];
  }
  __syncthreads();

  // ---- phase 3: sample + multiply, no barrier.  Software pipeline of depth one over the 36 k-steps (tap, group pair):
  // the corner reads and the two A fragments of step i+1 are in flight while step i is combined and multiplied.
  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
  DlSample q[2];
  constexpr int APF = 2;                                          // A fragments are fetched (from L2) two steps ahead
  half8 v[2][4], a[APF + 1][2];
  // A fragments: blob [cout tile 2][tap 9][step 4][lane 64][8 halves], read through a buffer descriptor (scalar base and
  // step offset, the lane offset the only vector operand: no 64-bit address arithmetic per load)
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.w), 0, 2 * 36 * 1024, 0x00020000);
  const int lane16 = lane * 16;
  float neg0 = -0.f;                                              // x * y == fma(x, y, -0) for every x, y: the opaque addend keeps
  asm volatile("" : "+v"(neg0));                                  // the first product of the combine on v_fma_mix (fp16 operand)
  auto issue_a = [&](int i) {
    a[i % (APF + 1)][0] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, i * 1024, 0));
    a[i % (APF + 1)][1] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, (36 + i) * 1024, 0));
  };
  auto issue = [&](auto interior, int i, int sl) {
    const int t = i >> 2, s2 = i & 3;
    const int g = 2 * s2 + hh;
    const half2v o2 = off[s2][t];
    const float h_im = (float)(oy - 1 + t / 3) + (float)o2[0];
    const float w_im = (float)(ox - 1 + t % 3) + (float)o2[1];
    if constexpr (decltype(interior)::value) q[sl] = dl_prepare_interior(wy0, wx0, g, h_im, w_im, msk[s2][t]);
    else q[sl] = dl_prepare(wy0, wx0, g, p.H, p.W, h_im, w_im, msk[s2][t]);
    v[sl][0] = *reinterpret_cast<const half8*>(win + q[sl].s00);
    v[sl][1] = *reinterpret_cast<const half8*>(win + q[sl].s01);
    v[sl][2] = *reinterpret_cast<const half8*>(win + q[sl].s10);
    v[sl][3] = *reinterpret_cast<const half8*>(win + q[sl].s11);
    if (__builtin_amdgcn_ballot_w64(!q[sl].iw) != 0) {            // rare: some lane's sample left the window
      if (!q[sl].iw) {
        if constexpr (decltype(interior)::value) q[sl] = dl_prepare(wy0, wx0, g, p.H, p.W, h_im, w_im, msk[s2][t]);   // with the border logic
        const half_t* xg = xn + g * 8;
        const int r0 = q[sl].y0 * p.W, r1 = q[sl].y1 * p.W;
        v[sl][0] = *reinterpret_cast<const half8*>(xg + (long)(r0 + q[sl].x0) * p.x_sp);
        v[sl][1] = *reinterpret_cast<const half8*>(xg + (long)(r0 + q[sl].x1) * p.x_sp);
        v[sl][2] = *reinterpret_cast<const half8*>(xg + (long)(r1 + q[sl].x0) * p.x_sp);
        v[sl][3] = *reinterpret_cast<const half8*>(xg + (long)(r1 + q[sl].x1) * p.x_sp);
      }
    }
  };
  auto consume = [&](int i) {
    const int sl = i & 1;
    half8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      b[j] = (half_t)__builtin_fmaf(q[sl].w1, (float)v[sl][0][j],
                                    __builtin_fmaf(q[sl].w2, (float)v[sl][1][j],
                                                   __builtin_fmaf(q[sl].w3, (float)v[sl][2][j], __builtin_fmaf(q[sl].w4, (float)v[sl][3][j], neg0))));
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i % (APF + 1)][0], b, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i % (APF + 1)][1], b, acc[1], 0, 0, 0);
  };
  auto run = [&](auto interior) {
#pragma unroll
    for (int i = 0; i < APF; ++i) issue_a(i);
    issue(interior, 0, 0);
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      if (i + APF < 36) issue_a(i + APF);
      if (i + 1 < 36) issue(interior, i + 1, (i + 1) & 1);
      consume(i);
    }
  };
  if constexpr (MODE != 1) {
    if (wy0 >= 0 && wx0 >= 0 && wy0 + DL_WH <= p.H && wx0 + DL_WW <= p.W) run(std::true_type{});
    else run(std::false_type{});
  }

  if (oy < p.H && ox < p.W) {
    half_t* yp = reinterpret_cast<half_t*>(p.y.p) + (long)n * p.y.sn + ((long)oy * p.W + ox) * p.y.sp;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int co = mt * 32 + 8 * gq + 4 * hh;
      if (co >= p.y.C) continue;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + co);
      half4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[mt][4 * gq + i] + b4[i];
        if (p.round16) v = (float)(half_t)v;
        v = act_apply(v, p.act, p.slope);
        o[i] = (half_t)v;
      }
      *reinterpret_cast<half4*>(yp + co) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------
// fp32 NCHW operator (`_ext.dcn_v2_forward`)
// block: 64 output pixels; loops over input channels, sampling kh*kw taps of one channel into
// LDS, then every thread accumulates its (pixel, cout-subset) outputs.
// ------------------------------------------------------------------------------------------
struct DcnF32Params {
  const float *input, *weight, *bias, *offset, *mask;
  float* output;
  int B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, G, Ho, Wo;
};

__device__ __forceinline__ float bilinear_f32(const float* im, int H, int W, float h, float w) {
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h - h_low, lw = w - w_low, hh = 1.f - lh, hw = 1.f - lw;
  float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
  if (h_low >= 0 && w_low >= 0) v1 = im[h_low * W + w_low];
  if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * W + w_high];
  if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * W + w_low];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * W + w_high];
  const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

constexpr int F32_COUT_PER_THREAD = 16;   // 4 waves x 16 = 64 output channels per pass

__global__ __launch_bounds__(256) void dcn_f32_forward_kernel(const DcnF32Params p) {
  __shared__ float col[49][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int npix = p.Ho * p.Wo;
  const int pix = blockIdx.x * 64 + lane;
  const bool valid = pix < npix;
  const int pcl = valid ? pix : npix - 1;
  const int ho = pcl / p.Wo, wo = pcl % p.Wo;
  const int K = p.kh * p.kw;
  const int cpg = p.C / p.G;
  const float* in_b = p.input + (long)b * p.C * p.H * p.W;
  const float* off_b = p.offset + (long)b * p.G * 2 * K * npix;
  const float* msk_b = p.mask + (long)b * p.G * K * npix;

  for (int cob = 0; cob < p.Cout; cob += 4 * F32_COUT_PER_THREAD) {
    float acc[F32_COUT_PER_THREAD];
#pragma unroll
    for (int i = 0; i < F32_COUT_PER_THREAD; ++i) acc[i] = 0.f;
    for (int c = 0; c < p.C; ++c) {
      const int g = c / cpg;
      __syncthreads();
      for (int t = wave; t < K; t += 4) {
        const int i = t / p.kw, j = t % p.kw;
        const float oh = off_b[((long)(g * 2 * K + 2 * t)) * npix + pcl];
        const float ow = off_b[((long)(g * 2 * K + 2 * t + 1)) * npix + pcl];
        const float m = msk_b[((long)(g * K + t)) * npix + pcl];
        const float h_im = (float)(ho * p.sh - p.ph + i * p.dh) + oh;
        const float w_im = (float)(wo * p.sw - p.pw + j * p.dw) + ow;
        float val = 0.f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W)
          val = bilinear_f32(in_b + (long)c * p.H * p.W, p.H, p.W, h_im, w_im);
        col[t][lane] = val * m;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < F32_COUT_PER_THREAD; ++i) {
        const int co = cob + wave * F32_COUT_PER_THREAD + i;
        if (co < p.Cout) {
          const float* wr = p.weight + ((long)co * p.C + c) * K;
          float a = acc[i];
          for (int t = 0; t < K; ++t) a = fmaf(wr[t], col[t][lane], a);
          acc[i] = a;
        }
      }
    }
    if (valid) {
#pragma unroll
      for (int i = 0; i < F32_COUT_PER_THREAD; ++i) {
        const int co = cob + wave * F32_COUT_PER_THREAD + i;
        if (co < p.Cout) p.output[((long)b * p.Cout + co) * npix + pix] = acc[i] + p.bias[co];
      }
    }
  }
}

}  // namespace

static int g_dcn_lds = getenv("TDVC_DCN_NO_LDS") == nullptr ? 1 : 0;
// tests and A/B benchmarks switch the LDS-window kernel off to run the same operator on dcn_fused_kernel
// (2, 3: the timing-only builds MODE 1, 2 of dcn_lds_kernel)
extern "C" void tdvc_debug_enable_dcn_lds(int enable) { g_dcn_lds = enable; }

extern "C" int tdvc_dcn_fused(const tdvc_dcn_desc* d, void* stream) {
  TDVC_CHECK(d, "tdvc_dcn_fused: null descriptor");
  TDVC_CHECK(fmap_ok16(d->x) && fmap_ok16(d->om) && fmap_ok16(d->y), "tdvc_dcn_fused: fmaps must be fp16, C/sp %% 8, aligned");
  const int G = d->groups;
  TDVC_CHECK(G == 8, "tdvc_dcn_fused: groups=%d unsupported (the fused kernel is built for the hot-path geometry: 8 groups x 8 channels; use tdvc_dcn_v2_forward_f32 otherwise)", G);
  TDVC_CHECK(d->x.C == 8 * G && d->y.C == 8 * G, "tdvc_dcn_fused: needs Cin = Cout = 8*groups (8 channels per group)");
  TDVC_CHECK(d->om.C >= 27 * G, "tdvc_dcn_fused: offset/mask fmap needs >= 27*groups channels");
  TDVC_CHECK(d->x.N == d->y.N && d->x.N == d->om.N && d->x.H == d->y.H && d->x.W == d->y.W && d->om.H == d->x.H && d->om.W == d->x.W,
             "tdvc_dcn_fused: geometry mismatch");
  TDVC_CHECK(d->w && aligned16(d->w) && d->bias && aligned16(d->bias), "tdvc_dcn_fused: weights/bias null or unaligned");
  TDVC_CHECK((long)d->x.H * d->x.W < 2147483647L, "tdvc_dcn_fused: image too large (pixel indices are 32-bit)");
  DcnParams p;
  p.x = reinterpret_cast<const half_t*>(d->x.p); p.x_sn = d->x.sn; p.x_sp = d->x.sp; p.H = d->x.H; p.W = d->x.W;
  p.om = reinterpret_cast<const half_t*>(d->om.p); p.om_sn = d->om.sn; p.om_sp = d->om.sp;
  p.y = to_dev(d->y);
  p.w = reinterpret_cast<const half_t*>(d->w); p.bias = d->bias;
  p.G = G; p.act = d->act; p.slope = d->slope; p.round16 = d->round_before_act;
  p.npix = (long)d->x.H * d->x.W;
  p.xp = reinterpret_cast<const half_t*>(d->x_planar);
  if (p.xp) {
    TDVC_CHECK(aligned16(d->x_planar), "tdvc_dcn_fused: x_planar scratch unaligned");
    const long items = p.npix * G;
    hipLaunchKernelGGL(dcn_planarise_kernel, dim3((unsigned)((items + 255) / 256), d->x.N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       p.x, p.x_sn, p.x_sp, p.npix, G, reinterpret_cast<half_t*>(d->x_planar));
  }
  if (g_dcn_lds && !p.xp && (long)d->x.H * d->x.W >= 8192) {
    static TdvcPerDeviceFlag attr_flags;
  bool& attr_done = attr_flags.flag();
    if (!attr_done) {
      for (const void* k : {reinterpret_cast<const void*>(&dcn_lds_kernel<0>), reinterpret_cast<const void*>(&dcn_lds_kernel<1>),
                            reinterpret_cast<const void*>(&dcn_lds_kernel<2>)}) {
        const hipError_t err = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, DL_LDS);
        TDVC_CHECK(err == hipSuccess, "tdvc_dcn_fused: cannot reserve %d bytes of LDS: %s", DL_LDS, hipGetErrorString(err));
      }
      attr_done = true;
    }
    dim3 grid_l((unsigned)(((d->x.W + DL_TX - 1) / DL_TX) * ((d->x.H + DL_TY - 1) / DL_TY)), d->x.N);
    auto kern = g_dcn_lds == 2 ? dcn_lds_kernel<1> : g_dcn_lds == 3 ? dcn_lds_kernel<2> : dcn_lds_kernel<0>;
    hipLaunchKernelGGL(kern, grid_l, dim3(256), DL_LDS, reinterpret_cast<hipStream_t>(stream), p);
    return tdvc_launch_status("tdvc_dcn_fused(lds)");
  }
  dim3 grid((unsigned)(((d->x.W + DCN_TPX - 1) / DCN_TPX) * ((d->x.H + DCN_TPY - 1) / DCN_TPY)), d->x.N);
  const size_t lds = (size_t)64 * (DCN_TAPS_PER_CHUNK * G * 16 + 16);
  hipLaunchKernelGGL(dcn_fused_kernel, grid, dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p);
  return tdvc_launch_status("tdvc_dcn_fused");
}

extern "C" int tdvc_dcn_v2_forward_f32(const float* input, const float* weight, const float* bias,
                                       const float* offset, const float* mask, float* output,
                                       int B, int C, int H, int W, int Cout,
                                       int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                       int deformable_group, void* stream) {
  TDVC_CHECK(input && weight && bias && offset && mask && output, "dcn_v2_forward: null tensor");
  TDVC_CHECK(B > 0 && C > 0 && H > 0 && W > 0 && Cout > 0, "dcn_v2_forward: empty tensor");
  TDVC_CHECK(kh >= 1 && kw >= 1 && kh * kw <= 49, "dcn_v2_forward: kernel %dx%d unsupported (<= 49 taps)", kh, kw);
  TDVC_CHECK(sh >= 1 && sw >= 1 && dh >= 1 && dw >= 1 && ph >= 0 && pw >= 0, "dcn_v2_forward: bad stride/dilation/pad");
  TDVC_CHECK(deformable_group >= 1 && C % deformable_group == 0, "dcn_v2_forward: channels %d not divisible by deformable_group %d", C, deformable_group);
  DcnF32Params p;
  p.input = input; p.weight = weight; p.bias = bias; p.offset = offset; p.mask = mask; p.output = output;
  p.B = B; p.C = C; p.H = H; p.W = W; p.Cout = Cout; p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw;
  p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw; p.G = deformable_group;
  p.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  p.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  TDVC_CHECK(p.Ho > 0 && p.Wo > 0, "dcn_v2_forward: empty output");
  dim3 grid((p.Ho * p.Wo + 63) / 64, B);
  hipLaunchKernelGGL(dcn_f32_forward_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  return tdvc_launch_status("tdvc_dcn_v2_forward_f32");
}
