// Autoregressive context-model support kernels (compress / decompress): gather the causal 5x5
// neighbourhoods of a batch of independent positions, quantise them and scatter them back.
// The dense math in between (context conv as a 12*M -> 2M 1x1 conv, entropy_parameters) runs on the
// MFMA conv kernel.  All integer / index outputs are exact; nothing here uses atomics.
#include "common.h"

namespace {

// the 12 causal taps of the type-A 5x5 mask in raster order: (dy, dx) relative to the centre
__constant__ int8_t kTapDy[12] = {-2, -2, -2, -2, -2, -1, -1, -1, -1, -1, 0, 0};
__constant__ int8_t kTapDx[12] = {-2, -1, 0, 1, 2, -2, -1, 0, 1, 2, -2, -1};

// T = half_t (default coders) or float (fp32 islands); VT = the 16-byte vector of T (8 halves / 4 floats)
template <typename T, typename VT>
__global__ void ar_gather_kernel(FMap yh, FMap pr, const int32_t* pos, int npos, FMap x1, FMap pc) {
  constexpr int VE = 16 / sizeof(T);              // elements per 16-byte chunk
  const int MV = yh.C / VE;                       // 16-byte chunks per position
  const int per = 12 * MV + pr.C / VE;            // chunks to move per position
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)npos * per) return;
  const int k = (int)(i / per), u = (int)(i % per);
  const int h = pos[2 * k], w = pos[2 * k + 1];
  if (u < 12 * MV) {
    const int t = u / MV, cv = u % MV;
    const int yy = h + kTapDy[t], xx = w + kTapDx[t];
    VT v;
#pragma unroll
    for (int j = 0; j < VE; ++j) v[j] = (T)0.f;
    if (yy >= 0 && xx >= 0 && xx < yh.W)
      v = *reinterpret_cast<const VT*>(reinterpret_cast<const T*>(yh.p) + ((long)yy * yh.W + xx) * yh.sp + cv * VE);
    *reinterpret_cast<VT*>(reinterpret_cast<T*>(x1.p) + (long)k * x1.sp + (t * MV + cv) * VE) = v;
  } else {
    const int cv = u - 12 * MV;
    const VT v = *reinterpret_cast<const VT*>(reinterpret_cast<const T*>(pr.p) + ((long)h * pr.W + w) * pr.sp + cv * VE);
    *reinterpret_cast<VT*>(reinterpret_cast<T*>(pc.p) + (long)k * pc.sp + cv * VE) = v;
  }
}

__device__ __forceinline__ int scale_index(float s, const float* table, int n) {
  s = fmaxf(s, 0.11f);
  int idx = n - 1;
  for (int j = 0; j < n - 1; ++j) idx -= (s <= table[j]) ? 1 : 0;
  return idx;
}

__global__ void ar_quantize_kernel(FMap y, FMap gp, const int32_t* pos, int npos, const float* table, int ntable,
                                   const int32_t* sym_in, FMap yh, int32_t* sym, int32_t* idx, long cbase) {
  // cbase < 0: sym_in / sym / idx are raster arrays [H][W][M]; cbase >= 0: they are compact arrays in the order of the
  // position list, this launch's position k at row cbase + k (the wavefront-ordered decoder)
  const int M = y.C;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)npos * M) return;
  const int k = (int)(i / M), c = (int)(i % M);
  const int h = pos[2 * k], w = pos[2 * k + 1];
  const float* g = reinterpret_cast<const float*>(gp.p) + (long)k * gp.sp;
  const float scale = g[c], mean = g[M + c];
  const long o = (cbase >= 0 ? cbase + k : (long)h * y.W + w) * M + c;
  int q;
  if (sym_in) q = sym_in[o];
  else q = (int)rintf(reinterpret_cast<const float*>(y.p)[((long)h * y.W + w) * y.sp + c] - mean);
  if (yh.f32) reinterpret_cast<float*>(yh.p)[((long)h * yh.W + w) * yh.sp + c] = (float)q + mean;
  else reinterpret_cast<half_t*>(yh.p)[((long)h * yh.W + w) * yh.sp + c] = (half_t)((float)q + mean);
  sym[o] = q;
  idx[o] = scale_index(scale, table, ntable);
}

__global__ void ar_indexes_kernel(FMap gp, const int32_t* pos, int npos, const float* table, int ntable, int M, int W, int32_t* idx, long cbase) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)npos * M) return;
  const int k = (int)(i / M), c = (int)(i % M);
  const int h = pos[2 * k], w = pos[2 * k + 1];
  const float* g = reinterpret_cast<const float*>(gp.p) + (long)k * gp.sp;
  idx[(cbase >= 0 ? cbase + k : (long)h * W + w) * M + c] = scale_index(g[c], table, ntable);
}

__global__ void round_symbols_kernel(FMap z, const float* median, int32_t* out) {
  const long npix = (long)z.H * z.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * z.C * z.N) return;
  const int c = (int)(i % z.C);
  const long t = i / z.C;
  const long pix = t % npix;
  const int n = (int)(t / npix);
  out[i] = (int)rintf(reinterpret_cast<const float*>(z.p)[(long)n * z.sn + pix * z.sp + c] - median[c]);
}

#define ST(s) reinterpret_cast<hipStream_t>(s)
inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

extern "C" int tdvc_ar_gather(const tdvc_fmap* y_hat, const tdvc_fmap* params, const int32_t* pos, int npos,
                              const tdvc_fmap* x1, const tdvc_fmap* pc, void* stream) {
  TDVC_CHECK(y_hat && params && pos && x1 && pc && npos >= 1, "tdvc_ar_gather: null / empty");
  const bool f32 = y_hat->dtype == TDVC_F32;
  auto okv = [&](const tdvc_fmap& f) {               // all four of one dtype; 16-byte chunks
    return f32 ? (fmap_ok32(f) && (f.C % 4) == 0 && (f.sp % 4) == 0 && aligned16(f.p)) : fmap_ok16(f);
  };
  TDVC_CHECK(okv(*y_hat) && okv(*params) && okv(*x1) && okv(*pc), "tdvc_ar_gather: four fp16 fmaps, or four fp32 fmaps (fp32 islands), expected");
  TDVC_CHECK(y_hat->N == 1 && params->N == 1 && params->H == y_hat->H && params->W == y_hat->W, "tdvc_ar_gather: one image at a time, params geometry must match y_hat");
  TDVC_CHECK(x1->C == 12 * y_hat->C && x1->W >= npos && x1->H == 1 && pc->C >= params->C && pc->W >= npos && pc->H == 1,
             "tdvc_ar_gather: x1 must be (1,1,>=npos,12*M), pc (1,1,>=npos,>=2M)");
  const int ve = f32 ? 4 : 8;
  const long total = (long)npos * (12 * y_hat->C / ve + params->C / ve);
  if (f32) hipLaunchKernelGGL((ar_gather_kernel<float, f32x4>), g1(total), dim3(256), 0, ST(stream), to_dev(*y_hat), to_dev(*params), pos, npos, to_dev(*x1), to_dev(*pc));
  else hipLaunchKernelGGL((ar_gather_kernel<half_t, half8>), g1(total), dim3(256), 0, ST(stream), to_dev(*y_hat), to_dev(*params), pos, npos, to_dev(*x1), to_dev(*pc));
  return tdvc_launch_status("tdvc_ar_gather");
}

extern "C" int tdvc_ar_quantize(const tdvc_fmap* y, const tdvc_fmap* gp, const int32_t* pos, int npos,
                                const float* scale_table, int ntable, const int32_t* symbols_in,
                                const tdvc_fmap* y_hat, int32_t* symbols, int32_t* indexes, void* stream) {
  TDVC_CHECK(gp && pos && scale_table && y_hat && symbols && indexes && npos >= 1 && ntable >= 2, "tdvc_ar_quantize: null / empty");
  TDVC_CHECK(symbols_in || (y && fmap_ok32(*y)), "tdvc_ar_quantize: need y (encoder) or symbols_in (decoder)");
  TDVC_CHECK(fmap_ok32(*gp) && (y_hat->dtype == TDVC_F32 ? fmap_ok32(*y_hat) : fmap_ok16(*y_hat)) && y_hat->N == 1 && gp->C >= 2 * y_hat->C && gp->W >= npos,
             "tdvc_ar_quantize: bad gp / y_hat");
  FMap yd = y ? to_dev(*y) : to_dev(*y_hat);
  yd.C = y_hat->C; yd.W = y_hat->W; yd.H = y_hat->H;
  hipLaunchKernelGGL(ar_quantize_kernel, g1((long)npos * y_hat->C), dim3(256), 0, ST(stream), yd, to_dev(*gp), pos, npos, scale_table, ntable,
                     symbols_in, to_dev(*y_hat), symbols, indexes, -1L);
  return tdvc_launch_status("tdvc_ar_quantize");
}

extern "C" int tdvc_ar_indexes(const tdvc_fmap* gp, const int32_t* pos, int npos, const float* scale_table, int ntable,
                               int M, int W, int32_t* indexes, void* stream) {
  TDVC_CHECK(gp && pos && scale_table && indexes && npos >= 1 && ntable >= 2 && M >= 1 && W >= 1 && fmap_ok32(*gp) && gp->C >= 2 * M,
             "tdvc_ar_indexes: bad arguments");
  hipLaunchKernelGGL(ar_indexes_kernel, g1((long)npos * M), dim3(256), 0, ST(stream), to_dev(*gp), pos, npos, scale_table, ntable, M, W, indexes, -1L);
  return tdvc_launch_status("tdvc_ar_indexes");
}

extern "C" int tdvc_round_symbols(const tdvc_fmap* z, const float* median, int32_t* out, void* stream) {
  TDVC_CHECK(z && median && out && fmap_ok32(*z), "tdvc_round_symbols: bad arguments");
  hipLaunchKernelGGL(round_symbols_kernel, g1((long)z->N * z->H * z->W * z->C), dim3(256), 0, ST(stream), to_dev(*z), median, out);
  return tdvc_launch_status("tdvc_round_symbols");
}

// The decoder's context loop in native code.  The y stream is in raster order (compressai's bitstream), and position
// (h, w) needs y_hat(h, w - 1): 8160 strictly serial steps per coder at 1080p.  Per step: gather -> context conv ->
// entropy_parameters (the caller's conv descriptors, fixed buffers) -> CDF indexes -> host range decoder -> quantise.
// Driving this from Python cost ~230 us per position (3.7 s per 1080p frame); here a step is seven enqueues, one
// 512-byte device->host copy + stream wait, the host decoder and one host->device copy.
extern "C" int tdvc_ar_decode_serial(const uint8_t* data, int64_t nbytes, const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                                     const int32_t* offsets, const tdvc_fmap* y_hat, const tdvc_fmap* params, const tdvc_fmap* x1,
                                     const tdvc_fmap* pc, const tdvc_conv_desc* convs, int nconvs, const tdvc_fmap* gp,
                                     const int32_t* pos_table, int npos_total, int M, int W, const float* scale_table, int ntable,
                                     int32_t* idx_dev, int32_t* sym_dev, void* stream) {
  TDVC_CHECK(data && cdfs && cdf_sizes && offsets && y_hat && params && x1 && pc && convs && gp && pos_table && scale_table && idx_dev && sym_dev,
             "tdvc_ar_decode_serial: null argument");
  TDVC_CHECK(nconvs >= 1 && nconvs <= 8 && npos_total >= 1 && M >= 1 && M <= 4096 && W >= 1, "tdvc_ar_decode_serial: bad sizes");
  void* dec = tdvc_rans_decoder_create(data, nbytes);
  if (!dec) return TDVC_EINVAL;
  int32_t* host = nullptr;                                // [2][M] indexes | symbols, pinned
  hipError_t err = hipHostMalloc(reinterpret_cast<void**>(&host), sizeof(int32_t) * 2 * (size_t)M, hipHostMallocDefault);
  if (err != hipSuccess) { tdvc_rans_decoder_destroy(dec); tdvc_set_error("tdvc_ar_decode_serial: hipHostMalloc failed: %s", hipGetErrorString(err)); return (int)err; }
  hipStream_t st = ST(stream);
  int rc = TDVC_OK;
  for (int k = 0; k < npos_total && rc == TDVC_OK; ++k) {
    const int32_t* pos = pos_table + 2 * (long)k;
    rc = tdvc_ar_gather(y_hat, params, pos, 1, x1, pc, stream);
    for (int c = 0; c < nconvs && rc == TDVC_OK; ++c) rc = tdvc_conv2d(&convs[c], stream);
    if (rc == TDVC_OK) rc = tdvc_ar_indexes(gp, pos, 1, scale_table, ntable, M, W, idx_dev, stream);
    if (rc != TDVC_OK) break;
    int32_t* idx_k = idx_dev + (long)k * M;               // raster order: position k = h * W + w owns [k*M, (k+1)*M)
    int32_t* sym_k = sym_dev + (long)k * M;
    err = hipMemcpyAsync(host, idx_k, sizeof(int32_t) * M, hipMemcpyDeviceToHost, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (err != hipSuccess) { tdvc_set_error("tdvc_ar_decode_serial: copy / sync failed: %s", hipGetErrorString(err)); rc = (int)err; break; }
    rc = tdvc_rans_decoder_decode(dec, host, M, cdfs, cdf_stride, cdf_sizes, offsets, host + M);
    if (rc != TDVC_OK) break;
    err = hipMemcpyAsync(sym_k, host + M, sizeof(int32_t) * M, hipMemcpyHostToDevice, st);
    if (err != hipSuccess) { tdvc_set_error("tdvc_ar_decode_serial: upload failed: %s", hipGetErrorString(err)); rc = (int)err; break; }
    // the next iteration's stream wait orders this upload before `host` is written again
    rc = tdvc_ar_quantize(nullptr, gp, pos, 1, scale_table, ntable, sym_dev, y_hat, sym_dev, idx_dev, stream);
  }
  (void)hipStreamSynchronize(st);
  (void)hipHostFree(host);
  tdvc_rans_decoder_destroy(dec);
  return rc;
}

// The context loop over anti-diagonals in native code, either direction.  Encoder: ~7 enqueues per step and no
// synchronisation (Python drove a step at ~0.5 ms: 0.33 s per 1080p frame for the two coders).  Decoder of a
// wavefront-ordered stream: per step the step's indexes to the host, its symbols out of the range decoder, and back.
extern "C" int tdvc_ar_wavefront(const uint8_t* data, int64_t nbytes, const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                                 const int32_t* offsets, const tdvc_fmap* y, const tdvc_fmap* y_hat, const tdvc_fmap* params,
                                 const tdvc_fmap* x1, const tdvc_fmap* pc, const tdvc_conv_desc* convs, int nconvs, const tdvc_fmap* gp,
                                 const int32_t* pos_dev, const int32_t* step_sizes, int nsteps, int M, int W,
                                 const float* scale_table, int ntable, int32_t* idx_dev, int32_t* sym_dev, void* stream) {
  TDVC_CHECK(y_hat && params && x1 && pc && convs && gp && pos_dev && step_sizes && scale_table && idx_dev && sym_dev, "tdvc_ar_wavefront: null argument");
  TDVC_CHECK((data != nullptr) != (y != nullptr), "tdvc_ar_wavefront: give y (encoder) or data (decoder), not both");
  TDVC_CHECK(!data || (cdfs && cdf_sizes && offsets && nbytes >= 4), "tdvc_ar_wavefront: the decoder needs the CDF tables");
  TDVC_CHECK(nconvs >= 1 && nconvs <= 8 && nsteps >= 1 && M >= 1 && M <= 4096 && W >= 1 && M == y_hat->C, "tdvc_ar_wavefront: bad sizes");
  long total = 0;
  int nmax = 0;
  for (int s = 0; s < nsteps; ++s) {
    TDVC_CHECK(step_sizes[s] >= 1 && step_sizes[s] <= x1->W && step_sizes[s] <= pc->W && step_sizes[s] <= gp->W, "tdvc_ar_wavefront: a step exceeds the staging buffers");
    total += step_sizes[s];
    nmax = step_sizes[s] > nmax ? step_sizes[s] : nmax;
  }
  TDVC_CHECK(total == (long)y_hat->H * y_hat->W, "tdvc_ar_wavefront: the steps must cover every position once");
  hipStream_t st = ST(stream);
  tdvc_conv_desc d[8];
  for (int c = 0; c < nconvs; ++c) d[c] = convs[c];
  void* dec = nullptr;
  int32_t* host = nullptr;                                // [2][nmax * M] indexes | symbols, pinned
  if (data) {
    dec = tdvc_rans_decoder_create(data, nbytes);
    if (!dec) return TDVC_EINVAL;
    hipError_t err = hipHostMalloc(reinterpret_cast<void**>(&host), sizeof(int32_t) * 2 * (size_t)nmax * M, hipHostMallocDefault);
    if (err != hipSuccess) { tdvc_rans_decoder_destroy(dec); tdvc_set_error("tdvc_ar_wavefront: hipHostMalloc failed: %s", hipGetErrorString(err)); return (int)err; }
  }
  int rc = TDVC_OK;
  long o = 0;
  for (int s = 0; s < nsteps && rc == TDVC_OK; ++s) {
    const int n = step_sizes[s];
    const int32_t* pos = pos_dev + 2 * o;
    rc = tdvc_ar_gather(y_hat, params, pos, n, x1, pc, stream);
    for (int c = 0; c < nconvs && rc == TDVC_OK; ++c) {
      d[c].x.W = n;                                      // the step's positions are the "pixels" of a (1, n) map
      d[c].y.W = n;
      rc = tdvc_conv2d(&d[c], stream);
    }
    if (rc != TDVC_OK) break;
    if (!data) {
      rc = tdvc_ar_quantize(y, gp, pos, n, scale_table, ntable, nullptr, y_hat, sym_dev, idx_dev, stream);
    } else {
      const long cnt = (long)n * M;
      hipLaunchKernelGGL(ar_indexes_kernel, g1(cnt), dim3(256), 0, st, to_dev(*gp), pos, n, scale_table, ntable, M, W, idx_dev, o);
      hipError_t err = hipMemcpyAsync(host, idx_dev + o * M, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, st);
      if (err == hipSuccess) err = hipStreamSynchronize(st);
      if (err != hipSuccess) { tdvc_set_error("tdvc_ar_wavefront: copy / sync failed: %s", hipGetErrorString(err)); rc = (int)err; break; }
      rc = tdvc_rans_decoder_decode(dec, host, cnt, cdfs, cdf_stride, cdf_sizes, offsets, host + (long)nmax * M);
      if (rc != TDVC_OK) break;
      err = hipMemcpyAsync(sym_dev + o * M, host + (long)nmax * M, sizeof(int32_t) * cnt, hipMemcpyHostToDevice, st);
      if (err != hipSuccess) { tdvc_set_error("tdvc_ar_wavefront: upload failed: %s", hipGetErrorString(err)); rc = (int)err; break; }
      // the next step's stream wait orders this upload before `host` is written again
      FMap yd = to_dev(*y_hat);
      hipLaunchKernelGGL(ar_quantize_kernel, g1(cnt), dim3(256), 0, st, yd, to_dev(*gp), pos, n, scale_table, ntable, sym_dev, yd, sym_dev, idx_dev, o);
      rc = tdvc_launch_status("tdvc_ar_wavefront");
    }
    o += n;
  }
  if (data) {
    (void)hipStreamSynchronize(st);
    (void)hipHostFree(host);
    tdvc_rans_decoder_destroy(dec);
  }
  return rc;
}
