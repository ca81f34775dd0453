// Implicit-GEMM convolution on MFMA for gfx950 — the conv analysis/synthesis transforms,
// feature extractors, SPyNet 7x7 stacks, masked context conv, GDN norm pools (SURVEY §8 a2-a9,
// a13-a15).  fp16 NHWC in, fp32 accumulate, fused epilogue.
//
//   D[cout][pixel] = sum_{tap, cin} W[cout][tap][cin] * X[pixel + tap][cin]
//
// MFMA roles: A = weights (rows = cout), B = activations (cols = pixels), so that a lane's 16
// accumulator registers hold 4 groups of 4 CONSECUTIVE output channels of ONE pixel -> 8-byte
// NHWC stores (v_mfma_f32_32x32x16_f16: C/D col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)).
//
// Work decomposition: one 256-thread workgroup (4 waves) = 8x32 output pixels x (MT*32) output
// channels; wave w owns output rows 2w, 2w+1.  The input halo tile of one channel chunk (CK
// channels) is staged once in LDS and re-read by every tap (9x / 25x / 49x reuse); LDS pixel
// stride is CK*2+16 bytes so that the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte
// slots (conflict-free).  Stride-2 convs de-interleave columns by parity while staging so the
// same holds.  Weights are pre-packed on the host in MFMA fragment order: one wave-wide 1 KiB
// coalesced global_load_dwordx4 per fragment, served from L2 (a layer's weights are <= 0.6 MB).
#include "conv_common.h"

using convk::ConvParams;
int launch_conv_v2(const ConvParams& p, int ntiles, int cout_blocks, int N, hipStream_t st);
bool conv_v2_eligible(const tdvc_conv_desc* d, int Ho, int Wo);
int launch_conv_v3(const ConvParams& p, int cout_blocks, int N, hipStream_t st);
bool conv_v3_eligible(const tdvc_conv_desc* d, int Ho, int Wo);
int launch_conv_v5(const ConvParams& p, int cout_blocks, int N, hipStream_t st);
bool conv_v5_eligible(const tdvc_conv_desc* d, int Ho, int Wo);
int conv_v5_chan_sum_rows(int Ho, int Wo, int cout_blocks, int N);
int launch_conv_v7(const ConvParams& p, int cout_blocks, int N, hipStream_t st);
bool conv_v7_eligible(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo);
int launch_conv_v11(const ConvParams& p, int cout_blocks, int N, hipStream_t st);
bool conv_v11_eligible(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo);
int launch_conv_row(int geo, const ConvParams& p, int N, hipStream_t st);
int conv_row_geometry(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo);
int launch_conv_v10(const ConvParams& p, int cout_blocks, int N, hipStream_t st);
bool conv_v10_eligible(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo);
int launch_conv_v9(const ConvParams& p, int ck8, int cout_tiles32, int N, hipStream_t st);
bool conv_v9_eligible(const tdvc_conv_desc* d, int Ho, int Wo, bool v3_ok);
bool conv_c8_eligible(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo);
int launch_conv_c8(const ConvParams& p, int N, hipStream_t st);
bool conv_n16_eligible(const tdvc_conv_desc* d, int Ho, int Wo);
int launch_conv_n16(const ConvParams& p, int N, hipStream_t st);
bool gdn128_eligible(const tdvc_conv_desc* d, const ConvParams& p, int Ho, int Wo);
int launch_gdn128(const ConvParams& p, int N, hipStream_t st);

namespace {

constexpr int TH = 8, TW = 32;

// LEAN (MT == 2 only, chosen on the host from ConvParams::simple == 2): the instantiation carries ONLY the lean
// transposed epilogue -- with both epilogue forms and the per-register fallback inlined the MT = 2 kernels sat at 256
// VGPRs (two workgroups per CU); the first layers (3x3 8->64 at 1080p, four per frame) are latency-bound there.
template <int CK8, int MT, int STRIDE, bool LEAN = false>
__global__ __launch_bounds__(256, LEAN ? 3 : 1) void conv_mfma_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CK = CK8 * 8;
  constexpr int PS = (CK8 == 1) ? 16 : CK * 2 + 16;  // LDS bytes per staged pixel
  int* tapoff = reinterpret_cast<int*>(smem);          // [49], first 256 bytes
  unsigned char* tile = smem + 256;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, r = lane & 31;
  const int tile_id = p.reverse ? (int)gridDim.x - 1 - tdvc_xcd_tile(blockIdx.x) : tdvc_xcd_tile(blockIdx.x);
  const int tx = tile_id % p.tiles_x, ty = tile_id / p.tiles_x;
  const int cb = blockIdx.y, n = blockIdx.z;

  const int TIH = (TH - 1) * STRIDE + p.kh;
  const int TIW = (TW - 1) * STRIDE + p.kw;
  const int HALFW = (TIW + 1) >> 1;
  const int TIWp = (STRIDE == 2) ? 2 * HALFW : TIW;

  if (tid < p.ntaps) {
    const int dy = p.tap_dy[tid], dx = p.tap_dx[tid];
    tapoff[tid] = (STRIDE == 2) ? (dy * TIWp + (dx & 1) * HALFW + (dx >> 1)) * PS : (dy * TIWp + dx) * PS;
  }

  int base[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) base[nt] = (((wave * 2 + nt) * STRIDE) * TIWp + r) * PS;

  f32x16 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  const half_t* xn = p.x + (long)n * p.x_sn;
  const int iy0 = ty * TH * STRIDE - p.pad, ix0 = tx * TW * STRIDE - p.pad;
  const int total = TIH * TIW * CK8;

  for (int ch = 0; ch < p.nchunks; ++ch) {
    if (ch > 0) __syncthreads();
    // ---- stage the halo tile of channels [ch*CK, ch*CK+CK) ------------------------------
    for (int idx = tid; idx < total; idx += 256) {
      const int c8 = idx % CK8;
      const int pix = idx / CK8;
      const int c = pix % TIW, rr = pix / TIW;
      const int iy = iy0 + rr, ix = ix0 + c;
      const int cg = ch * CK + c8 * 8;
      half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && cg < p.Cin)
        v = *reinterpret_cast<const half8*>(xn + ((long)iy * p.W + ix) * p.x_sp + cg);
      if (p.square) v = v * v;
      const int cpos = (STRIDE == 2) ? ((c & 1) * HALFW + (c >> 1)) : c;
      *reinterpret_cast<half8*>(tile + (rr * TIWp + cpos) * PS + c8 * 16) = v;
    }
    __syncthreads();

    // ---- MFMA over (tap, channel) for this chunk ------------------------------------------
    const half_t* wp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      wp[mt] = p.w + ((((long)(cb * MT + mt) * p.nchunks + ch) * p.steps) * 64 + lane) * 8;

    if constexpr (CK8 == 1) {
      for (int s = 0; s < p.steps; ++s) {
        int tap = 2 * s + hh;
        tap = tap < p.ntaps ? tap : p.ntaps - 1;   // padded k-chunk: its weights are zero
        const int toff = tapoff[tap];
        half8 a[MT], b[2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { a[mt] = *reinterpret_cast<const half8*>(wp[mt]); wp[mt] += 512; }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b[nt] = *reinterpret_cast<const half8*>(tile + base[nt] + toff);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
      }
    } else {
      for (int tap = 0; tap < p.ntaps; ++tap) {
        const int toff = tapoff[tap] + hh * 16;
#pragma unroll
        for (int s2 = 0; s2 < CK8 / 2; ++s2) {
          half8 a[MT], b[2];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) { a[mt] = *reinterpret_cast<const half8*>(wp[mt]); wp[mt] += 512; }
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            b[nt] = *reinterpret_cast<const half8*>(tile + base[nt] + toff + s2 * 32);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------
  if constexpr (MT == 2 && LEAN) {
    __syncthreads();      // transposed full-line stores through LDS (the staging tile is dead now)
    convk::epilogue_simple_rows<2, false, 2>(p, acc, p.bias + cb * 64, tile + wave * (32 * 144), n, cb * 64,
                                             ty * TH + wave * 2, tx * TW, lane, false);
    return;
  } else if constexpr (MT == 2) {
    if (p.simple) {
      __syncthreads();
      convk::epilogue_simple_rows<2, false, 1>(p, acc, p.bias + cb * 64, tile + wave * (32 * 144), n, cb * 64,
                                               ty * TH + wave * 2, tx * TW, lane, false);
      return;
    }
  }
  const int ox = tx * TW + r;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int oy = ty * TH + wave * 2 + nt;
    if (oy >= p.Ho || ox >= p.Wo) continue;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = (cb * MT + mt) * 32 + 8 * g + 4 * hh;   // 4 consecutive channels co..co+3
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[mt][nt][4 * g + i];
        convk::epilogue4(p, n, oy, ox, co, v);
      }
    }
  }
}

template <int CK8, int MT, int STRIDE>
int launch(const ConvParams& p, dim3 grid, size_t lds, hipStream_t st) {
  if constexpr (MT == 2) {
    if (p.simple == 2) {
      hipLaunchKernelGGL((conv_mfma_kernel<CK8, MT, STRIDE, true>), grid, dim3(256), lds, st, p);
      return tdvc_launch_status("tdvc_conv2d");
    }
  }
  hipLaunchKernelGGL((conv_mfma_kernel<CK8, MT, STRIDE, false>), grid, dim3(256), lds, st, p);
  return tdvc_launch_status("tdvc_conv2d");
}

template <int CK8, int MT>
int launch_s(const ConvParams& p, int stride, dim3 grid, size_t lds, hipStream_t st) {
  return stride == 1 ? launch<CK8, MT, 1>(p, grid, lds, st) : launch<CK8, MT, 2>(p, grid, lds, st);
}

template <int CK8>
int launch_m(const ConvParams& p, int mt, int stride, dim3 grid, size_t lds, hipStream_t st) {
  return mt == 1 ? launch_s<CK8, 1>(p, stride, grid, lds, st) : launch_s<CK8, 2>(p, stride, grid, lds, st);
}

inline int lds_bytes(int ck, int kh, int kw, int stride) {
  const int ps = (ck == 8) ? 16 : ck * 2 + 16;
  const int tih = (TH - 1) * stride + kh, tiw = (TW - 1) * stride + kw;
  const int tiwp = stride == 2 ? 2 * ((tiw + 1) >> 1) : tiw;
  return 256 + tih * tiwp * ps;
}

inline int cout_tiles(int cout) { return cout <= 32 ? 1 : 2 * ((cout + 63) / 64); }

}  // namespace

extern "C" int tdvc_conv_plan(int cin, int kh, int kw, int stride) {
  if (cin <= 0 || (cin % 8) != 0 || kh < 1 || kw < 1 || kh > 7 || kw > 7 || (stride != 1 && stride != 2)) {
    tdvc_set_error("tdvc_conv_plan: unsupported geometry cin=%d k=%dx%d stride=%d", cin, kh, kw, stride);
    return TDVC_EINVAL;
  }
  const int cands[4] = {64, 32, 16, 8};
  for (int i = 0; i < 4; ++i) {
    const int ck = cands[i];
    if (ck > cin && ck != 8) {
      // do not stage more channels than exist unless cin is not a power-of-two multiple
      if (cin % ck != 0 && cin < ck) continue;
    }
    if (lds_bytes(ck, kh, kw, stride) <= 64 * 1024) return ck;
  }
  tdvc_set_error("tdvc_conv_plan: no LDS plan for k=%dx%d stride=%d", kh, kw, stride);
  return TDVC_EINVAL;
}

extern "C" int64_t tdvc_conv_packed_bytes(int cout, int cin, int ntaps, int ck) {
  if (cout <= 0 || cin <= 0 || ntaps <= 0 || ntaps > TDVC_MAX_TAPS || (ck != 8 && ck != 16 && ck != 32 && ck != 64)) return TDVC_EINVAL;
  const int ck8 = ck / 8;
  const int64_t nchunks = (cin + ck - 1) / ck;
  const int64_t steps = (ntaps * ck8 + 1) / 2;
  return (int64_t)cout_tiles(cout) * nchunks * steps * 64 * 8 * 2;
}

extern "C" int tdvc_pack_conv_weights(const float* w, int cout, int cin_real, int cin, int kh, int kw,
                                      int ntaps, const int8_t* tap_dy, const int8_t* tap_dx, int ck, uint16_t* dst) {
  TDVC_CHECK(w && dst && tap_dy && tap_dx, "tdvc_pack_conv_weights: null pointer");
  TDVC_CHECK(tdvc_conv_packed_bytes(cout, cin, ntaps, ck) > 0, "tdvc_pack_conv_weights: bad geometry");
  TDVC_CHECK(cin_real <= cin, "tdvc_pack_conv_weights: cin_real > cin");
  const int ck8 = ck / 8;
  const int nchunks = (cin + ck - 1) / ck;
  const int steps = (ntaps * ck8 + 1) / 2;
  const int tiles = cout_tiles(cout);
  half_t* out = reinterpret_cast<half_t*>(dst);
  for (int t = 0; t < tiles; ++t)
    for (int ch = 0; ch < nchunks; ++ch)
      for (int s = 0; s < steps; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int r = lane & 31, h = lane >> 5;
          const int co = t * 32 + r;
          const int kc = 2 * s + h;
          half_t* o = out + ((((int64_t)t * nchunks + ch) * steps + s) * 64 + lane) * 8;
          for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (kc < ntaps * ck8 && co < cout) {
              const int tap = kc / ck8, c8 = kc % ck8;
              const int ci = ch * ck + c8 * 8 + j;
              const int dy = tap_dy[tap], dx = tap_dx[tap];
              if (ci < cin_real && dy >= 0 && dy < kh && dx >= 0 && dx < kw)
                v = w[(((int64_t)co * cin_real + ci) * kh + dy) * kw + dx];
            }
            o[j] = (half_t)v;
          }
        }
  return TDVC_OK;
}

static thread_local char g_last_kernel[48] = "";

// Tile-walk direction.  The 256 MB Infinity Cache sits in front of HBM and every layer streams a map slightly larger
// than it (64 channels x 1088 x 1920 fp16 = 267 MB): when layer l+1 reads, in the same raster order, what layer l just
// wrote, the head of the map has already been pushed out by its tail and nothing hits.  Consecutive launches therefore
// walk the tile raster in OPPOSITE directions: the consumer starts on the producer's most recently written tiles (and on
// the tail of the residual the producer read), which are still resident.  Launch parity is per host thread (one stream
// of launches per rank); any order is correct, the alternation only decides what hits.
static thread_local unsigned g_walk_parity = 0;
static int g_walk_mode = -1;      // -1: read TDVC_CONV_WALK once (0 = always forward, 1 = alternate [default])
extern "C" void tdvc_debug_set_conv_walk(int mode) { g_walk_mode = mode; }
static int next_walk_reverse() {
  if (g_walk_mode < 0) { const char* e = getenv("TDVC_CONV_WALK"); g_walk_mode = e ? atoi(e) : 1; }
  return g_walk_mode == 1 ? (int)(g_walk_parity++ & 1u) : 0;
}

// query_rows: validate and dispatch as tdvc_conv2d would, but return the rows of tdvc_conv_desc::chan_sum instead of launching
static int conv2d_impl(const tdvc_conv_desc* d, void* stream, bool query_rows) {
  TDVC_CHECK(d, "tdvc_conv2d: null descriptor");
  if (d->x.dtype == TDVC_F32) {           // fp32 islands (pnet.py:33,57): fp32 activations + fp32 packing -> conv_f32.hip
    if (query_rows) return 0;
    TDVC_CHECK(!d->chan_sum, "tdvc_conv2d: chan_sum is not available on the fp32 path (tdvc_conv_chan_sum_rows() == 0)");
    snprintf(g_last_kernel, sizeof(g_last_kernel), "conv_f32");
    return tdvc_conv2d_f32(d, stream);
  }
  TDVC_CHECK(fmap_ok16(d->x), "tdvc_conv2d: input must be an fp16 fmap with C,sp %% 8 == 0 and 16-byte aligned");
  TDVC_CHECK(d->w && aligned16(d->w), "tdvc_conv2d: weights null/unaligned");
  TDVC_CHECK(d->stride == 1 || d->stride == 2, "tdvc_conv2d: stride %d unsupported", d->stride);
  TDVC_CHECK(d->ntaps >= 1 && d->ntaps <= TDVC_MAX_TAPS && d->kh >= 1 && d->kh <= 7 && d->kw >= 1 && d->kw <= 7,
             "tdvc_conv2d: bad kernel %dx%d ntaps=%d", d->kh, d->kw, d->ntaps);
  TDVC_CHECK(d->ck == 8 || d->ck == 16 || d->ck == 32 || d->ck == 64, "tdvc_conv2d: bad ck %d", d->ck);
  TDVC_CHECK(d->cout >= 1, "tdvc_conv2d: cout");
  for (int t = 0; t < d->ntaps; ++t)
    TDVC_CHECK(d->tap_dy[t] >= 0 && d->tap_dy[t] < d->kh && d->tap_dx[t] >= 0 && d->tap_dx[t] < d->kw,
               "tdvc_conv2d: tap %d out of the %dx%d window", t, d->kh, d->kw);
  int Ho = (d->x.H + 2 * d->pad - d->kh) / d->stride + 1;
  int Wo = (d->x.W + 2 * d->pad - d->kw) / d->stride + 1;
  if (d->s2d) {
    TDVC_CHECK(d->kh == 2 && d->kw == 2 && d->stride == 1 && d->pad == 1 && d->ntaps == 4 && d->ck == 32 && !d->square_input && !d->gdn,
               "tdvc_conv2d: s2d expects the virtual 2x2 / stride 1 / pad 1 conv packed with ck=32");
    TDVC_CHECK((d->x.H % 2) == 0 && (d->x.W % 2) == 0 && (d->x.C % 32) == 0 && d->cout >= 64,
               "tdvc_conv2d: s2d needs even H, W, C %% 32 == 0 and cout >= 64 (got %dx%dx%d, cout %d)", d->x.H, d->x.W, d->x.C, d->cout);
    Ho = d->x.H / 2;
    Wo = d->x.W / 2;
  }
  TDVC_CHECK(Ho > 0 && Wo > 0, "tdvc_conv2d: empty output");
  TDVC_CHECK((long)Ho * Wo * 4 < 2147483647L && (long)d->x.H * d->x.W < 2147483647L, "tdvc_conv2d: image too large (pixel indices are 32-bit)");
  const int lds = lds_bytes(d->ck, d->kh, d->kw, d->stride);

  const int shuf = d->out_mode == TDVC_OUT_SHUFFLE2;
  if (d->out_mode == TDVC_OUT_NCHW_F32) {
    TDVC_CHECK(d->y.p && d->y.N == d->x.N, "tdvc_conv2d: NCHW output null / batch mismatch");
  } else {
    TDVC_CHECK(d->y.dtype == TDVC_F32 ? fmap_ok32(d->y) : fmap_ok16(d->y), "tdvc_conv2d: bad output fmap");
    TDVC_CHECK(d->y.N == d->x.N && d->y.H == (shuf ? 2 * Ho : Ho) && d->y.W == (shuf ? 2 * Wo : Wo),
               "tdvc_conv2d: output geometry %dx%d does not match conv result %dx%d%s", d->y.H, d->y.W, Ho, Wo,
               shuf ? " (x2 shuffle)" : "");
    if (shuf) TDVC_CHECK((d->cout % 128) == 0, "tdvc_conv2d: SHUFFLE2 needs cout %% 128 == 0");
    if (d->y.dtype == TDVC_F16) TDVC_CHECK((d->y.C % 8) == 0, "tdvc_conv2d: fp16 output C %% 8");
  }
  if (d->gdn) {
    TDVC_CHECK(fmap_ok16(d->aux) && d->aux.H == Ho && d->aux.W == Wo && d->aux.N == d->x.N && d->aux.C >= d->cout &&
                   !shuf && (d->cout % 64) == 0,
               "tdvc_conv2d: GDN aux fmap mismatch");
  }
  if (d->res.p) {
    TDVC_CHECK(d->res.dtype == TDVC_F32 ? fmap_ok32(d->res) : fmap_ok16(d->res), "tdvc_conv2d: bad residual fmap");
    TDVC_CHECK(d->res.N == d->x.N && d->res.H == (shuf ? 2 * Ho : Ho) && d->res.W == (shuf ? 2 * Wo : Wo),
               "tdvc_conv2d: residual geometry mismatch");
  }
  if (d->res2.p) {
    TDVC_CHECK(fmap_ok16(d->res2) && d->res2.N == d->x.N && d->res2.H == (shuf ? 2 * Ho : Ho) && d->res2.W == (shuf ? 2 * Wo : Wo),
               "tdvc_conv2d: bad second residual fmap");
  }
  if (d->bias) TDVC_CHECK(aligned16(d->bias), "tdvc_conv2d: bias unaligned");

  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.x = reinterpret_cast<const half_t*>(d->x.p); p.x_sn = d->x.sn; p.x_sp = d->x.sp;
  p.H = d->x.H; p.W = d->x.W; p.Cin = d->x.C;
  p.w = reinterpret_cast<const half_t*>(d->w); p.bias = d->bias;
  p.y = to_dev(d->y); p.Ho = Ho; p.Wo = Wo; p.cout = d->cout;
  p.aux = d->gdn ? to_dev(d->aux) : null_fmap();
  p.res = d->res.p ? to_dev(d->res) : null_fmap();
  p.res2 = d->res2.p ? to_dev(d->res2) : null_fmap();
  p.ntaps = d->ntaps; p.kh = d->kh; p.kw = d->kw; p.pad = d->pad;
  p.in_stride = d->stride;
  const int ck8 = d->ck / 8;
  p.s2d = d->s2d; p.Corig = d->x.C;
  p.bcast_T = d->bcast_T; p.bcast_slope = d->bcast_slope;
  p.nchunks = d->s2d ? (4 * d->x.C) / d->ck : (d->x.C + d->ck - 1) / d->ck;
  p.steps = (d->ntaps * ck8 + 1) / 2;
  p.square = d->square_input; p.gdn = d->gdn; p.act = d->act; p.slope = d->slope;
  p.round16 = d->round_before_act; p.out_mode = d->out_mode;
  memcpy(p.tap_dy, d->tap_dy, sizeof(p.tap_dy));
  memcpy(p.tap_dx, d->tap_dx, sizeof(p.tap_dx));
  const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
  p.tiles_x = tiles_x;
  p.reverse = query_rows ? 0 : next_walk_reverse();
  p.csum = nullptr;
  const int tiles = cout_tiles(d->cout);
  const int mt = tiles == 1 ? 1 : 2;
  if (mt == 2 && convk::conv_is_simple(p)) {
    p.simple = convk::conv_is_lean(p) ? 2 : 1;       // 2: the lean packed-fp16 form of the transposed epilogue (conv_common.h)
    p.slope = convk::conv_simple_slope(p);
  }
  {
    // fused channel sums (tdvc_conv_desc::chan_sum): the lean epilogue of conv_mfma_v5 with one block of 64 output channels
    const bool csum_ok = !d->bcast_T && !d->s2d && tiles == 2 && p.simple == 2 && !conv_v9_eligible(d, Ho, Wo, conv_v3_eligible(d, Ho, Wo)) &&
                         !gdn128_eligible(d, p, Ho, Wo) && conv_v5_eligible(d, Ho, Wo);
    if (query_rows) return csum_ok ? conv_v5_chan_sum_rows(Ho, Wo, 1, d->x.N) : 0;
    TDVC_CHECK(!d->chan_sum || (csum_ok && (reinterpret_cast<uintptr_t>(d->chan_sum) & 15) == 0),
               "tdvc_conv2d: chan_sum on a conv whose kernel has no fused channel sum (tdvc_conv_chan_sum_rows() == 0) or unaligned");
    p.csum = d->chan_sum;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  auto chose = [](const char* name) { snprintf(g_last_kernel, sizeof(g_last_kernel), "%s", name); };
  if (d->bcast_T) {          // temporal 1x1 conv + broadcast add + LeakyReLU over the slices at y: one kernel takes it (conv_mfma_v5)
    TDVC_CHECK(d->bcast_T == 4 && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->cout == 64 && d->y.C == 64 && d->y.sp >= 4 * 64 &&
                   d->y.dtype == TDVC_F16 && d->out_mode == TDVC_OUT_NHWC && d->act == TDVC_ACT_NONE && !d->gdn && !d->res.p && !d->res2.p &&
                   !d->square_input && !d->round_before_act && d->bias && d->bcast_slope >= 0.f && d->bcast_slope <= 1.f && conv_v5_eligible(d, Ho, Wo) &&
                   convk::conv_is_lean(p),
               "tdvc_conv2d: bcast_T needs a plain 1x1 / stride 1 conv to 64 channels of >= 8192 pixels, y a 64-channel window of a buffer with >= 4 slices, bcast_T == 4");
    p.simple = 2;
    p.slope = 1.f;
    chose("conv_mfma_v5(bcast)");
    return launch_conv_v5(p, 1, d->x.N, st);
  }
  if (d->s2d) {
    if (const int geo = conv_row_geometry(d, p, Ho, Wo); geo >= 0) { chose("conv_row(s2d)"); return launch_conv_row(geo, p, d->x.N, st); }
    TDVC_CHECK(conv_v3_eligible(d, Ho, Wo), "tdvc_conv2d: s2d conv not eligible for the stage-pipelined kernel");
    chose("conv_mfma_v3(s2d)");
    return launch_conv_v3(p, tiles / 2, d->x.N, st);
  }
  if (conv_v9_eligible(d, Ho, Wo, conv_v3_eligible(d, Ho, Wo))) { chose("conv_mfma_v9"); return launch_conv_v9(p, ck8, tiles, d->x.N, st); }
  if (gdn128_eligible(d, p, Ho, Wo)) { chose("gdn128"); return launch_gdn128(p, d->x.N, st); }
  if (conv_v5_eligible(d, Ho, Wo)) { chose("conv_mfma_v5"); return launch_conv_v5(p, tiles / 2, d->x.N, st); }
  if (conv_c8_eligible(d, p, Ho, Wo)) { chose("conv_c8"); return launch_conv_c8(p, d->x.N, st); }
  if (conv_n16_eligible(d, Ho, Wo)) { chose("conv_n16"); return launch_conv_n16(p, d->x.N, st); }
  if (const int geo = conv_row_geometry(d, p, Ho, Wo); geo >= 0) { chose("conv_row"); return launch_conv_row(geo, p, d->x.N, st); }
  if (conv_v10_eligible(d, p, Ho, Wo)) { chose("conv_mfma_v10"); return launch_conv_v10(p, tiles / 2, d->x.N, st); }
  if (conv_v7_eligible(d, p, Ho, Wo)) { chose("conv_mfma_v7"); return launch_conv_v7(p, tiles / 2, d->x.N, st); }
  if (conv_v11_eligible(d, p, Ho, Wo)) { chose("conv_mfma_v11"); return launch_conv_v11(p, tiles / 2, d->x.N, st); }
  if (conv_v3_eligible(d, Ho, Wo)) { chose("conv_mfma_v3"); return launch_conv_v3(p, tiles / 2, d->x.N, st); }
  if (conv_v2_eligible(d, Ho, Wo)) { chose("conv_mfma_v2"); return launch_conv_v2(p, 0, tiles / 2, d->x.N, st); }
  TDVC_CHECK(lds <= 64 * 1024, "tdvc_conv2d: LDS plan %d bytes too large for the direct kernel (use tdvc_conv_plan)", lds);
  dim3 grid(tiles_x * tiles_y, tiles / mt, d->x.N);
  snprintf(g_last_kernel, sizeof(g_last_kernel), "conv_mfma<%d,%d,%d>", ck8, mt, d->stride);
  const size_t lds_v1 = (p.simple && lds < 256 + 4 * 32 * 144) ? 256 + 4 * 32 * 144 : lds;
  switch (ck8) {
    case 1: return launch_m<1>(p, mt, d->stride, grid, lds_v1, st);
    case 2: return launch_m<2>(p, mt, d->stride, grid, lds_v1, st);
    case 4: return launch_m<4>(p, mt, d->stride, grid, lds_v1, st);
    default: return launch_m<8>(p, mt, d->stride, grid, lds_v1, st);
  }
}

extern "C" int tdvc_conv2d(const tdvc_conv_desc* d, void* stream) { return conv2d_impl(d, stream, false); }
extern "C" int tdvc_conv_chan_sum_rows(const tdvc_conv_desc* d) { return conv2d_impl(d, nullptr, true); }

extern "C" const char* tdvc_last_conv_kernel(void) { return g_last_kernel; }
