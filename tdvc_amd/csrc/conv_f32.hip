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

__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvParams p, const F32Extra e) {
  __shared__ int tdy[TDVC_MAX_TAPS], tdx[TDVC_MAX_TAPS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, r = lane & 31;
  if (tid < p.ntaps) { tdy[tid] = p.tap_dy[tid]; tdx[tid] = p.tap_dx[tid]; }
  __syncthreads();
  const int ct = blockIdx.y * 4 + wave;
  if (ct >= e.cout_tiles) return;                       // wave-uniform; no barrier follows

  const int px = blockIdx.x * 32 + r;
  const bool pv = px < e.total_px;
  const int pxc = pv ? px : 0;
  const int hw = p.Ho * p.Wo;
  const int n = pxc / hw, rem = pxc - n * hw;
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
  const int iy0 = oy * p.in_stride - p.pad, ix0 = ox * p.in_stride - p.pad;
  const float* xn = e.x + (long)n * p.x_sn;

  const int T = p.nchunks * p.steps;
  const int ck8 = e.ck8, CK = ck8 * 8, H2 = ck8 >> 1;
  const float* wp = e.w + ((long)ct * T) * 512 + lane * 8;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // two k-steps per trip: four 32-byte operand loads are in flight before the first MFMA consumes one (the odd tail
  // repeats the last k-step with a zeroed activation fragment)
  for (int g2 = 0; g2 < T; g2 += 2) {
    f32x8 a[2], b[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int g = min(g2 + u, T - 1);
      const int ch = g / p.steps, s = g - ch * p.steps;
      int tap, cofs;
      if (ck8 == 1) {
        tap = min(2 * s + hh, p.ntaps - 1);               // the padded half step carries zero weights
        cofs = ch * 8;
      } else {
        tap = s / H2;
        cofs = ch * CK + (s - tap * H2) * 16 + hh * 8;
      }
      const int iy = iy0 + tdy[tap], ix = ix0 + tdx[tap];
      ok[u] = pv && g2 + u < T && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && cofs < p.Cin;
      const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1), cc = min(cofs, p.Cin - 8);
      const float* bp = xn + ((long)iyc * p.W + ixc) * p.x_sp + cc;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
      const float* ap = wp + (long)g * 512;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[u][j] = a0[j]; a[u][4 + j] = a1[j]; b[u][j] = b0[j]; b[u][4 + j] = b1[j]; }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float bv = ok[u] ? b[u][j] : 0.f;
        if (p.square) bv = bv * bv;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][j], bv, acc, 0, 0, 0);
      }
    }
  }

  if (!pv) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = acc[4 * g + i];
    convk::epilogue4<true>(p, n, oy, ox, ct * 32 + 8 * g + 4 * hh, v);
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
  TDVC_CHECK(d->cout >= 1 && !d->s2d && !d->round_before_act, "tdvc_conv2d_f32: cout / s2d / round16 not supported in the fp32 form");
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
  const dim3 grid((unsigned)((e.total_px + 31) / 32), (unsigned)((e.cout_tiles + 3) / 4));
  hipLaunchKernelGGL(conv_f32_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, e);
  return tdvc_launch_status("tdvc_conv2d_f32");
}
