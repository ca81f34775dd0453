// conv_f32 — the fp32 islands of the TDVC path: `main/model/pnet.py:33-49,57-73` switch autocast OFF around the two
// coders (g_a, h_a, h_s, context model, entropy parameters, g_s of compressai's Cheng2020Anchor), and with
// `enabled_amp=False` nothing on the path is reduced precision.  The default build runs the coders fp16-in /
// fp32-accumulate; this kernel is the exact mode: fp32 activations, fp32 weights, fp32 accumulation on
// v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak, MI355X_MICROARCH.md), so that round(y - mu) sees what the fp32 CPU
// reference sees and compress() emits the same symbols.
//
// Same implicit GEMM as the fp16 kernels: D[cout][pixel] = sum_{tap, cin} W[cout][tap][cin] * X[pixel + tap][cin],
// A = weights (rows = cout), B = activations (cols = pixels).  The weights are packed in the SAME fragment order as
// the fp16 layers ([cout tile 32][chunk][k-step][lane 64][8 values], `tdvc_pack_conv_weights_indexed_f32`), as floats:
// a lane's 8 values are 8 consecutive input channels of one tap.  The fp32 MFMA contracts K = 2 per instruction (lane
// (r, hh) supplies A[r][hh] and B[hh][r]), so one k-step of the fp16 layout becomes eight MFMAs, MFMA j taking value j
// of every lane's 8 — the K order inside a k-step is free as long as A and B agree.
//
// The matrix pipe is 16x slower than in fp16 (64 cycles per MFMA per SIMD), so operand traffic is not the problem
// here and nothing is staged: both operands come straight from L2 / L1 (the weight fragment is two coalesced 1 KB
// loads, the activation fragment 32 bytes per lane at a clamped address, zeroed by a select).  One workgroup =
// 32 output pixels (flattened over batch, rows, columns) x up to 4 x 32 output channels, one 32 x 32 tile per wave with
// the whole contraction; the waves of a workgroup read the same activations (L1 hits).  General epilogue (epilogue4):
// bias, GDN / inverse GDN (fp32 multiplicand), activation, fp32 / fp16 residuals, NHWC fp32 / fp16, PixelShuffle, NCHW.
#include "conv_common.h"

using convk::ConvParams;

namespace {

typedef float f32x8 __attribute__((ext_vector_type(8)));

struct F32Extra {
  const float* x;          // fp32 activations (ConvParams::x is typed half_t*)
  const float* w;          // fp32 packed weights
  int ck8;                 // channel chunk / 8 of the packing
  int cout_tiles;          // 32-row tiles of the padded cout
  int total_px;
};

// MT x NT register blocking (round 3): a wave computes MT cout tiles x NT pixel groups of 32, so that one pair of operand
// fragments feeds MT * NT MFMAs.  With one 32 x 32 tile per wave (round 2) every 8 MFMAs (512 matrix cycles) pulled 4 KB of
// operands per wave from L1 / L2 -- 32 B per cycle and CU with the four SIMDs busy, the practical L2 -> CU rate: the kernel
// sat on operand traffic at 0.54 of the fp32 matrix peak, not on the matrix pipe.  2 x 2 blocking halves the bytes per MFMA.
template <int MT, int NT>
__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvParams p, const F32Extra e) {
  __shared__ int tdy[TDVC_MAX_TAPS], tdx[TDVC_MAX_TAPS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, r = lane & 31;
  if (tid < p.ntaps) { tdy[tid] = p.tap_dy[tid]; tdx[tid] = p.tap_dx[tid]; }
  __syncthreads();
  // workgroup = 2 pixel blocks (NT * 32 pixels each) x 2 cout blocks (MT tiles each): waves 0,1 share the pixels, 0,2 the weights
  const int ct0 = (blockIdx.y * 2 + (wave & 1)) * MT;
  if (ct0 >= e.cout_tiles) return;                      // wave-uniform; no barrier follows
  const int pb = blockIdx.x * 2 + (wave >> 1);

  const int hw = p.Ho * p.Wo;
  bool pv[NT];
  int pn[NT], poy[NT], pox[NT], iy0[NT], ix0[NT];
  const float* xn[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int px = (pb * NT + nt) * 32 + r;
    pv[nt] = px < e.total_px;
    const int pxc = pv[nt] ? px : 0;
    pn[nt] = pxc / hw;
    const int rem = pxc - pn[nt] * hw;
    poy[nt] = rem / p.Wo;
    pox[nt] = rem - poy[nt] * p.Wo;
    iy0[nt] = poy[nt] * p.in_stride - p.pad;
    ix0[nt] = pox[nt] * p.in_stride - p.pad;
    xn[nt] = e.x + (long)pn[nt] * p.x_sn;
  }
  if (!__builtin_amdgcn_readfirstlane(__ballot(pv[0]) != 0ull)) return;      // a pixel block past the end (wave-uniform)

  const int T = p.nchunks * p.steps;
  const int ck8 = e.ck8, CK = ck8 * 8, H2 = ck8 >> 1;
  const float* wp = e.w + ((long)ct0 * T) * 512 + lane * 8;      // tile ct0 + mt: + mt * T * 512

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  // software pipeline over k-steps: the operand fragments of step g + 1 (MT + NT loads of 32 bytes per lane) are ISSUED before
  // the 8 * MT * NT MFMAs of step g and consumed after them.  fetch() only issues loads and derives the zeroing multiplier; every
  // use of a loaded register (the multiplier, the square) sits in mfmas(), and sched_barrier(0) pins the order load / matrix /
  // load / matrix -- left to itself hipcc sank the loads behind `ok ? x : 0` branches (round 2) or regrouped the two sets so
  // that every k-step waited vmcnt(0) for the loads it had just issued (first round-3 build): 0.5 of the fp32 matrix peak.
  f32x4 a[2][MT][2], b[2][NT][2];
  unsigned bm[2][NT];
  auto fetch = [&](int g, f32x4 (&af)[MT][2], f32x4 (&bf)[NT][2], unsigned (&mk)[NT]) {
    const int gc = min(g, T - 1);
    const int ch = gc / p.steps, s = gc - ch * p.steps;
    int tap, cofs;
    if (ck8 == 1) {
      tap = min(2 * s + hh, p.ntaps - 1);               // the padded half step carries zero weights
      cofs = ch * 8;
    } else {
      tap = s / H2;
      cofs = ch * CK + (s - tap * H2) * 16 + hh * 8;
    }
    const int dy = tdy[tap], dx = tdx[tap];
    const int cc = min(cofs, p.Cin - 8);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float* ap = wp + ((long)mt * T + gc) * 512;
      af[mt][0] = *reinterpret_cast<const f32x4*>(ap);
      af[mt][1] = *reinterpret_cast<const f32x4*>(ap + 4);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int iy = iy0[nt] + dy, ix = ix0[nt] + dx;
      const bool ok = (int)pv[nt] & (int)(g < T) & (int)(iy >= 0) & (int)(iy < p.H) & (int)(ix >= 0) & (int)(ix < p.W) & (int)(cofs < p.Cin);   // no short-circuit branches
      const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
      const float* bp = xn[nt] + ((long)iyc * p.W + ixc) * p.x_sp + cc;
      bf[nt][0] = *reinterpret_cast<const f32x4*>(bp);
      bf[nt][1] = *reinterpret_cast<const f32x4*>(bp + 4);
      // out-of-image taps are zeroed by an OPAQUE bit mask, not by a select (a select made hipcc sink the loads behind branches) and
      // not by a multiply (Inf * 0 at a clamped border address would be NaN): exact zeros for any input
      unsigned m = ok ? 0xFFFFFFFFu : 0u;
      asm volatile("" : "+v"(m));
      mk[nt] = m;
    }
  };
  auto mfmas = [&](f32x4 (&af)[MT][2], f32x4 (&bf)[NT][2], unsigned (&mk)[NT]) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bv[nt] = __uint_as_float(__float_as_uint(bf[nt][h2][j]) & mk[nt]);
          if (p.square) bv[nt] *= bv[nt];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][h2][j], bv[nt], acc[mt][nt], 0, 0, 0);
      }
  };
  fetch(0, a[0], b[0], bm[0]);
  for (int g = 0; g < T; g += 2) {
    fetch(g + 1, a[1], b[1], bm[1]);                    // g + 1 == T: a zeroed fragment (multiplier 0), harmless
    __builtin_amdgcn_sched_barrier(0);
    mfmas(a[0], b[0], bm[0]);
    __builtin_amdgcn_sched_barrier(0);
    fetch(g + 2, a[0], b[0], bm[0]);                    // unconditional (past the end: clamped addresses, multiplier 0): a load under
    __builtin_amdgcn_sched_barrier(0);                  // a branch makes hipcc drain vmcnt(0) at the join
    mfmas(a[1], b[1], bm[1]);                           // g + 1 == T (odd T): adds exact zeros
    __builtin_amdgcn_sched_barrier(0);
  }

#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (!pv[nt]) continue;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[mt][nt][4 * g + i];
        convk::epilogue4<true>(p, pn[nt], poy[nt], pox[nt], (ct0 + mt) * 32 + 8 * g + 4 * hh, v);
      }
  }
}

__global__ void pack_indexed_f32_kernel(const float* __restrict__ w, const int* __restrict__ row_off, const int* __restrict__ chan_off,
                                        const int* __restrict__ tap_off, const unsigned char* __restrict__ row_mask,
                                        const unsigned char* __restrict__ chan_mask, const unsigned char* __restrict__ tap_mask,
                                        int cout, int cin, int ntaps, int ck, int nchunks, int steps, long total, float* __restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per 8 packed values (same item order as the fp16 packer)
  if (e >= total) return;
  const int lane = (int)(e & 63);
  long q = e >> 6;
  const int s = (int)(q % steps); q /= steps;
  const int ch = (int)(q % nchunks);
  const int t = (int)(q / nchunks);
  const int ck8 = ck >> 3;
  const int r = lane & 31, h = lane >> 5;
  const int co = t * 32 + r, kc = 2 * s + h;
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (kc < ntaps * ck8 && co < cout) {
    const int tap = kc / ck8, c8 = kc - tap * ck8;
    const int ro = row_off[co], to = tap_off[tap];
    const unsigned tm = tap_mask[tap];
    if (ro >= 0 && (tm & row_mask[co]) == 0u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ci = ch * ck + c8 * 8 + j;
        if (ci < cin) {
          const int cof = chan_off[ci];
          if (cof >= 0 && (tm & chan_mask[ci]) == 0u) o[j] = w[(long)ro + cof + to];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) out[e * 8 + j] = o[j];
}

inline int cout_tiles32(int cout) { return cout <= 32 ? 1 : 2 * ((cout + 63) / 64); }

}  // namespace

extern "C" int tdvc_pack_conv_weights_indexed_f32(const float* w, const int32_t* row_off, const int32_t* chan_off, const int32_t* tap_off,
                                                  const uint8_t* row_mask, const uint8_t* chan_mask, const uint8_t* tap_mask,
                                                  int cout, int cin, int ntaps, int ck, float* dst, void* stream) {
  TDVC_CHECK(w && row_off && chan_off && tap_off && row_mask && chan_mask && tap_mask && dst && aligned16(dst),
             "tdvc_pack_conv_weights_indexed_f32: null / unaligned pointer");
  const int64_t bytes16 = tdvc_conv_packed_bytes(cout, cin, ntaps, ck);          // the fp16 blob: 16 bytes per item
  TDVC_CHECK(bytes16 > 0, "tdvc_pack_conv_weights_indexed_f32: bad geometry");
  const int ck8 = ck / 8, nchunks = (cin + ck - 1) / ck, steps = (ntaps * ck8 + 1) / 2;
  const long total = bytes16 / 16;
  hipLaunchKernelGGL(pack_indexed_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     w, row_off, chan_off, tap_off, row_mask, chan_mask, tap_mask, cout, cin, ntaps, ck, nchunks, steps, total, dst);
  return tdvc_launch_status("tdvc_pack_conv_weights_indexed_f32");
}

// tdvc_conv2d forwards here when the input fmap is fp32 (`d->w` then points at the fp32 packing).
extern "C" int tdvc_conv2d_f32(const tdvc_conv_desc* d, void* stream) {
  TDVC_CHECK(d, "tdvc_conv2d_f32: null descriptor");
  TDVC_CHECK(fmap_ok32(d->x) && (d->x.C % 8) == 0 && (d->x.sp % 4) == 0 && (d->x.sn % 4) == 0 && aligned16(d->x.p),
             "tdvc_conv2d_f32: input must be an fp32 fmap with C %% 8 == 0, 16-byte aligned pixels");
  TDVC_CHECK(d->w && aligned16(d->w), "tdvc_conv2d_f32: weights null/unaligned");
  TDVC_CHECK(d->stride == 1 || d->stride == 2, "tdvc_conv2d_f32: stride %d unsupported", d->stride);
  TDVC_CHECK(d->ntaps >= 1 && d->ntaps <= TDVC_MAX_TAPS && d->kh >= 1 && d->kh <= 7 && d->kw >= 1 && d->kw <= 7,
             "tdvc_conv2d_f32: bad kernel %dx%d ntaps=%d", d->kh, d->kw, d->ntaps);
  TDVC_CHECK(d->ck == 8 || d->ck == 16 || d->ck == 32 || d->ck == 64, "tdvc_conv2d_f32: bad ck %d", d->ck);
  TDVC_CHECK(d->cout >= 1 && !d->s2d && !d->round_before_act && !d->bcast_T, "tdvc_conv2d_f32: cout / s2d / round16 / bcast_T not supported in the fp32 form");
  for (int t = 0; t < d->ntaps; ++t)
    TDVC_CHECK(d->tap_dy[t] >= 0 && d->tap_dy[t] < d->kh && d->tap_dx[t] >= 0 && d->tap_dx[t] < d->kw,
               "tdvc_conv2d_f32: tap %d out of the %dx%d window", t, d->kh, d->kw);
  const int Ho = (d->x.H + 2 * d->pad - d->kh) / d->stride + 1;
  const int Wo = (d->x.W + 2 * d->pad - d->kw) / d->stride + 1;
  TDVC_CHECK(Ho > 0 && Wo > 0, "tdvc_conv2d_f32: empty output");
  TDVC_CHECK((long)Ho * Wo * 4 * d->x.N < 2147483647L && (long)d->x.H * d->x.W < 2147483647L, "tdvc_conv2d_f32: image too large (pixel indices are 32-bit)");
  const int shuf = d->out_mode == TDVC_OUT_SHUFFLE2;
  if (d->out_mode == TDVC_OUT_NCHW_F32) {
    TDVC_CHECK(d->y.p && d->y.N == d->x.N, "tdvc_conv2d_f32: NCHW output null / batch mismatch");
  } else {
    TDVC_CHECK(d->y.dtype == TDVC_F32 ? fmap_ok32(d->y) : fmap_ok16(d->y), "tdvc_conv2d_f32: bad output fmap");
    TDVC_CHECK(d->y.N == d->x.N && d->y.H == (shuf ? 2 * Ho : Ho) && d->y.W == (shuf ? 2 * Wo : Wo),
               "tdvc_conv2d_f32: output geometry %dx%d does not match conv result %dx%d%s", d->y.H, d->y.W, Ho, Wo, shuf ? " (x2 shuffle)" : "");
    if (shuf) TDVC_CHECK((d->cout % 16) == 0, "tdvc_conv2d_f32: SHUFFLE2 needs cout %% 16 == 0");
    if (d->y.dtype == TDVC_F16) TDVC_CHECK((d->y.C % 8) == 0, "tdvc_conv2d_f32: fp16 output C %% 8");
    else TDVC_CHECK((d->y.C % 4) == 0 && (d->y.sp % 4) == 0, "tdvc_conv2d_f32: fp32 output C, pixel stride %% 4");
  }
  if (d->gdn)
    TDVC_CHECK((d->aux.dtype == TDVC_F32 ? fmap_ok32(d->aux) : fmap_ok16(d->aux)) && d->aux.H == Ho && d->aux.W == Wo && d->aux.N == d->x.N &&
                   d->aux.C >= d->cout && !shuf && (d->cout % 4) == 0, "tdvc_conv2d_f32: GDN aux fmap mismatch");
  if (d->res.p) {
    TDVC_CHECK(d->res.dtype == TDVC_F32 ? fmap_ok32(d->res) : fmap_ok16(d->res), "tdvc_conv2d_f32: bad residual fmap");
    TDVC_CHECK(d->res.N == d->x.N && d->res.H == (shuf ? 2 * Ho : Ho) && d->res.W == (shuf ? 2 * Wo : Wo), "tdvc_conv2d_f32: residual geometry mismatch");
  }
  if (d->res2.p)
    TDVC_CHECK((d->res2.dtype == TDVC_F32 ? fmap_ok32(d->res2) : fmap_ok16(d->res2)) && d->res2.N == d->x.N &&
                   d->res2.H == (shuf ? 2 * Ho : Ho) && d->res2.W == (shuf ? 2 * Wo : Wo), "tdvc_conv2d_f32: bad second residual fmap");
  if (d->bias) TDVC_CHECK(aligned16(d->bias), "tdvc_conv2d_f32: bias unaligned");

  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.x = nullptr; p.x_sn = d->x.sn; p.x_sp = d->x.sp;
  p.H = d->x.H; p.W = d->x.W; p.Cin = d->x.C;
  p.w = nullptr; p.bias = d->bias;
  p.y = to_dev(d->y); p.Ho = Ho; p.Wo = Wo; p.cout = d->cout;
  p.aux = d->gdn ? to_dev(d->aux) : null_fmap();
  p.res = d->res.p ? to_dev(d->res) : null_fmap();
  p.res2 = d->res2.p ? to_dev(d->res2) : null_fmap();
  p.ntaps = d->ntaps; p.kh = d->kh; p.kw = d->kw; p.pad = d->pad;
  p.in_stride = d->stride;
  const int ck8 = d->ck / 8;
  p.nchunks = (d->x.C + d->ck - 1) / d->ck;
  p.steps = (d->ntaps * ck8 + 1) / 2;
  p.square = d->square_input; p.gdn = d->gdn; p.act = d->act; p.slope = d->slope;
  p.round16 = 0; p.out_mode = d->out_mode;
  memcpy(p.tap_dy, d->tap_dy, sizeof(p.tap_dy));
  memcpy(p.tap_dx, d->tap_dx, sizeof(p.tap_dx));
  F32Extra e;
  e.x = reinterpret_cast<const float*>(d->x.p);
  e.w = reinterpret_cast<const float*>(d->w);
  e.ck8 = ck8;
  e.cout_tiles = cout_tiles32(d->cout);
  e.total_px = d->x.N * Ho * Wo;
  // workgroup = 2 x (NT * 32) pixels by 2 x MT cout tiles; small maps keep NT = 1 so that the few pixels still spread over the CUs
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool big = e.total_px >= 16384;
  if (e.cout_tiles == 1) {
    if (big) hipLaunchKernelGGL((conv_f32_kernel<1, 2>), dim3((unsigned)((e.total_px + 127) / 128), 1), dim3(256), 0, st, p, e);
    else hipLaunchKernelGGL((conv_f32_kernel<1, 1>), dim3((unsigned)((e.total_px + 63) / 64), 1), dim3(256), 0, st, p, e);
  } else {
    const unsigned gy = (unsigned)((e.cout_tiles + 3) / 4);
    if (big) hipLaunchKernelGGL((conv_f32_kernel<2, 2>), dim3((unsigned)((e.total_px + 127) / 128), gy), dim3(256), 0, st, p, e);
    else hipLaunchKernelGGL((conv_f32_kernel<2, 1>), dim3((unsigned)((e.total_px + 63) / 64), gy), dim3(256), 0, st, p, e);
  }
  return tdvc_launch_status("tdvc_conv2d_f32");
}
