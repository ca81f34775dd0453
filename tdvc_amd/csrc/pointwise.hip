// HBM-bound kernels of the TDVC hot path for gfx950: layout conversion, SE attention, bilinear
// resampling, the SPyNet warp (flow_warp), in-loop-filter patch matching and the entropy-model
// rate terms.  All are streaming kernels: channel-innermost fmaps, 16-byte accesses per lane,
// one pass over the data, reductions through wave shuffles / LDS with deterministic two-stage sums.
#include "pointwise_common.h"

namespace {

// ------------------------------------------------------------------ layout
__global__ void nchw_to_fmap_kernel(const float* src, int C, FMap y) {
  const long npix = (long)y.H * y.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * y.N) return;
  const int n = (int)(i / npix);
  const long pix = i % npix;
  for (int c0 = 0; c0 < y.C; c0 += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (c0 + j < C) ? src[((long)n * C + c0 + j) * npix + pix] : 0.f;
    store8(y, n, pix, c0, v);
  }
}

__global__ void fmap_to_nchw_kernel(FMap x, int C, float* dst) {
  const long npix = (long)x.H * x.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * x.N) return;
  const int n = (int)(i / npix);
  const long pix = i % npix;
  for (int c0 = 0; c0 < C; c0 += 8) {
    float v[8];
    load8(x, n, pix, c0, v);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c0 + j < C) dst[((long)n * C + c0 + j) * npix + pix] = v[j];
  }
}

// ------------------------------------------------------------------ elementwise
__global__ void scale_act_res_kernel(FMap a, const float* gate, int act, float slope, FMap r, float rs, FMap y, FMap y2) {
  const int chunks = (a.C + 7) / 8;
  const long npix = (long)a.H * a.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * chunks * a.N) return;
  const int c = (int)(i % chunks) * 8;
  const long t = i / chunks;
  const long pix = t % npix;
  const int n = (int)(t / npix);
  float v[8];
  load8(a, n, pix, c, v);
  if (gate) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c + j < a.C) v[j] *= gate[(long)n * a.C + c + j];
  }
  if (act) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = act_apply(v[j], act, slope);
  }
  if (r.p) {
    float q[8];
    load8(r, n, pix, c, q);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += rs * q[j];
  }
  store8(y, n, pix, c, v);
  if (y2.p) store8(y2, n, pix, c, v);
}

// fp16 fast path of the kernel above (every map fp16, C % 8 == 0: the full-resolution SE scalings, the residual
// subtraction, the coders' identity adds).  Same arithmetic, different work decomposition: a grid-stride loop in which a
// thread keeps U independent 16-byte loads (+ U residual loads) in flight, and the gate comes as two float4 loads instead
// of eight scalar ones -- with one item per thread and eight extra load instructions per item the kernel sat on the L1's
// instruction rate at 2.3 TB/s.
template <int U>
__global__ __launch_bounds__(256) void scale_act_res16_kernel(FMap a, const float* gate, int act, float slope, FMap r, float rs, FMap y, FMap y2,
                                                              unsigned total, unsigned chunks, unsigned npix) {
  const unsigned T = gridDim.x * 256u;
  for (unsigned base = blockIdx.x * 256u + threadIdx.x; base < total; base += U * T) {
    half8 av[U], rv[U];
    unsigned ck[U], nn[U];
    long pa[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned i = base + u * T;
      ok[u] = i < total;
      const unsigned ii = ok[u] ? i : 0u;
      const unsigned pp = ii / chunks;
      ck[u] = ii - pp * chunks;
      nn[u] = pp / npix;
      pa[u] = (long)(pp - nn[u] * npix);
      av[u] = *reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(a.p) + (long)nn[u] * a.sn + pa[u] * a.sp + ck[u] * 8);
      if (r.p) rv[u] = *reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(r.p) + (long)nn[u] * r.sn + pa[u] * r.sp + ck[u] * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (float)av[u][j];
      if (gate) {
        const f32x4* g4 = reinterpret_cast<const f32x4*>(gate + (long)nn[u] * a.C + ck[u] * 8);
        const f32x4 g0 = g4[0], g1 = g4[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] *= g0[j]; v[4 + j] *= g1[j]; }
      }
      if (act) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act_apply(v[j], act, slope);
      }
      if (r.p) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rs * (float)rv[u][j];
      }
      half8 h;
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = (half_t)v[j];
      if (ok[u]) {
        *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(y.p) + (long)nn[u] * y.sn + pa[u] * y.sp + ck[u] * 8) = h;
        if (y2.p) *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(y2.p) + (long)nn[u] * y2.sn + pa[u] * y2.sp + ck[u] * 8) = h;
      }
    }
  }
}

__global__ void add_flow_kernel(FMap off, FMap flow) {
  const int chunks = off.C / 8;
  const long npix = (long)off.H * off.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * chunks * off.N) return;
  const int c = (int)(i % chunks) * 8;
  const long t = i / chunks;
  const long pix = t % npix;
  const int n = (int)(t / npix);
  const float* fp = reinterpret_cast<const float*>(flow.p) + (long)n * flow.sn + pix * flow.sp;
  const float fx = fp[0], fy = fp[1];
  float v[8];
  load8(off, n, pix, c, v);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] += (j & 1) ? fy : fx;
  store8(off, n, pix, c, v);
}

__global__ void bcast_add_act_kernel(FMap x, FMap b, float slope) {
  const int chunks = x.C / 8;
  const long npix = (long)x.H * x.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * chunks * x.N) return;
  const int c = (int)(i % chunks) * 8;
  const long t = i / chunks;
  const long pix = t % npix;
  const int n = (int)(t / npix);
  float v[8], q[8];
  load8(x, n, pix, c, v);
  load8(b, n, pix, c % b.C, q);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = act_apply(v[j] + q[j], TDVC_ACT_LRELU, slope);
  store8(x, n, pix, c, v);
}

// the same with one thread per (pixel, 8 channels of b): b is loaded once, the T slices are T independent loads in flight
template <int T>
__global__ __launch_bounds__(256) void bcast_add_act_t_kernel(FMap x, FMap b, float slope, unsigned total, unsigned chunks, unsigned npix) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const unsigned pp = i / chunks, ck = i - pp * chunks;
  const unsigned n = pp / npix;
  const long pix = (long)(pp - n * npix);
  const half8 bv = *reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(b.p) + (long)n * b.sn + pix * b.sp + ck * 8);
  half_t* xp = reinterpret_cast<half_t*>(x.p) + (long)n * x.sn + pix * x.sp + ck * 8;
  half8 xv[T];
#pragma unroll
  for (int t = 0; t < T; ++t) xv[t] = *reinterpret_cast<const half8*>(xp + t * b.C);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (half_t)act_apply((float)xv[t][j] + (float)bv[j], TDVC_ACT_LRELU, slope);
    *reinterpret_cast<half8*>(xp + t * b.C) = h;
  }
}

// ------------------------------------------------------------------ SE attention
// partial[n][blk][c] = sum over the block's pixel range
__global__ __launch_bounds__(256) void channel_sum_kernel(FMap x, float* partial, int nblocks) {
  __shared__ float red[256][9];
  const int chunks = x.C / 8;            // <= 32
  const int lanes = 256 / chunks;        // pixel lanes
  const int tid = threadIdx.x;
  const int ck = tid % chunks, pl = tid / chunks;
  const int n = blockIdx.y;
  const long npix = (long)x.H * x.W;
  const long per = (npix + nblocks - 1) / nblocks;
  const long p0 = (long)blockIdx.x * per;
  const long p1 = p0 + per < npix ? p0 + per : npix;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (pl < lanes) {
    long pix = p0 + pl;
    for (; pix + 3 * lanes < p1; pix += 4 * lanes) {       // four independent 16-byte loads in flight per thread; summed in pixel order
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) load8(x, n, pix + u * lanes, ck * 8, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[u][j];
    }
    for (; pix < p1; pix += lanes) {
      float v[8];
      load8(x, n, pix, ck * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tid][j] = (pl < lanes) ? acc[j] : 0.f;
  __syncthreads();
  if (tid < x.C) {
    const int c = tid, cc = c / 8, j = c % 8;
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[l * chunks + cc][j];
    partial[((long)n * nblocks + blockIdx.x) * x.C + c] = s;
  }
}

__global__ __launch_bounds__(1024) void se_gate_kernel(const float* partial, int nblocks, float inv_count, int C, int Cmid,
                                                       const float* w1, const float* b1, const float* w2, const float* b2,
                                                       float* gate) {
  __shared__ float mean[256];
  __shared__ float mid[32];
  __shared__ float part[1024];
  const int n = blockIdx.x, tid = threadIdx.x;
  // fixed-order two-level sum of the partials: `ways` threads per channel, then a serial tail
  const int ways = 1024 / C < 1 ? 1 : (1024 / C > 1024 / 32 ? 32 : 1024 / C);
  for (int i = tid; i < C * ways; i += 1024) {
    const int c = i % C, wy = i / C;
    float s = 0.f;
    int b = wy;
    for (; b + 3 * ways < nblocks; b += 4 * ways) {        // four loads in flight, added in block order
      float q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = partial[((long)n * nblocks + b + u * ways) * C + c];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += q[u];
    }
    for (; b < nblocks; b += ways) s += partial[((long)n * nblocks + b) * C + c];
    part[wy * C + c] = s;
  }
  __syncthreads();
  if (tid < C) {
    float s = 0.f;
    for (int wy = 0; wy < ways; ++wy) s += part[wy * C + tid];
    mean[tid] = s * inv_count;
  }
  __syncthreads();
  if (tid < Cmid) {
    float s = b1[tid];
    for (int c = 0; c < C; ++c) s += w1[tid * C + c] * mean[c];
    mid[tid] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  if (tid < C) {
    float s = b2[tid];
    for (int j = 0; j < Cmid; ++j) s += w2[tid * Cmid + j] * mid[j];
    gate[(long)n * C + tid] = 1.f / (1.f + expf(-s));
  }
}

// ------------------------------------------------------------------ resampling
__global__ void upsample2x_kernel(FMap x, FMap y) {
  const int chunks = x.C / 8;
  const long npix = (long)y.H * y.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * chunks * y.N) return;
  const int c = (int)(i % chunks) * 8;
  const long t = i / chunks;
  const long pix = t % npix;
  const int n = (int)(t / npix);
  const int oy = (int)(pix / y.W), ox = (int)(pix % y.W);
  const float sy = fmaxf((oy + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf((ox + 0.5f) * 0.5f - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < x.H - 1), x1 = x0 + (x0 < x.W - 1);
  const float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  float a[8], b[8], cc[8], d[8], v[8];
  load8(x, n, (long)y0 * x.W + x0, c, a);
  load8(x, n, (long)y0 * x.W + x1, c, b);
  load8(x, n, (long)y1 * x.W + x0, c, cc);
  load8(x, n, (long)y1 * x.W + x1, c, d);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = ly0 * (lx0 * a[j] + lx1 * b[j]) + ly1 * (lx0 * cc[j] + lx1 * d[j]);
  store8(y, n, pix, c, v);
}

__global__ void avgpool2_kernel(FMap x, FMap y) {
  const long npix = (long)y.H * y.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * y.N) return;
  const int n = (int)(i / npix);
  const long pix = i % npix;
  const int oy = (int)(pix / y.W), ox = (int)(pix % y.W);
  const float* xp = reinterpret_cast<const float*>(x.p) + (long)n * x.sn;
  float* yp = reinterpret_cast<float*>(y.p) + (long)n * y.sn + pix * y.sp;
  const long p00 = ((long)(2 * oy) * x.W + 2 * ox) * x.sp, p10 = p00 + (long)x.W * x.sp;
  for (int c = 0; c < y.C; ++c)
    yp[c] = (xp[p00 + c] + xp[p00 + x.sp + c] + xp[p10 + c] + xp[p10 + x.sp + c]) * 0.25f;
}

// flow_up (x2, align_corners=True, *2) + border-clamped bilinear warp + 8-channel assembly
// Round 3: whole-pixel vector accesses when the maps allow it (VEC: ref / supp are 4-float pixels, the flows 2-float pixels, all
// naturally aligned -- the SPyNet pyramid's own buffers): 4 + 4 + 1 wide loads per pixel instead of 8 + 12 + 3 scalar ones (the
// kernel sat on the load-instruction rate: 64 us for a 120 MB pass at 1080p).  Same arithmetic, same order.
template <bool VEC>
__global__ void spynet_level_input_kernel(FMap ref, FMap supp, FMap flow_lo, FMap flow_up, FMap cat8) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int H = ref.H, W = ref.W;
  const long npix = (long)H * W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * ref.N) return;
  const int n = (int)(i / npix);
  const long pix = i % npix;
  const int y = (int)(pix / W), x = (int)(pix % W);
  float fx = 0.f, fy = 0.f;
  if (flow_lo.p) {
    const int h2 = flow_lo.H, w2 = flow_lo.W;
    const float ry = (H > 1) ? (float)(h2 - 1) / (float)(H - 1) : 0.f;
    const float rx = (W > 1) ? (float)(w2 - 1) / (float)(W - 1) : 0.f;
    const float sy = ry * y, sx = rx * x;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < h2 - 1), x1 = x0 + (x0 < w2 - 1);
    const float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* fl = reinterpret_cast<const float*>(flow_lo.p) + (long)n * flow_lo.sn;
    const float* pa = fl + ((long)y0 * w2 + x0) * flow_lo.sp;
    const float* pb = fl + ((long)y0 * w2 + x1) * flow_lo.sp;
    const float* pc = fl + ((long)y1 * w2 + x0) * flow_lo.sp;
    const float* pd = fl + ((long)y1 * w2 + x1) * flow_lo.sp;
    f32x2 a, b, c, d;
    if constexpr (VEC) {
      a = *reinterpret_cast<const f32x2*>(pa); b = *reinterpret_cast<const f32x2*>(pb);
      c = *reinterpret_cast<const f32x2*>(pc); d = *reinterpret_cast<const f32x2*>(pd);
    } else {
      a = f32x2{pa[0], pa[1]}; b = f32x2{pb[0], pb[1]}; c = f32x2{pc[0], pc[1]}; d = f32x2{pd[0], pd[1]};
    }
    fx = (ly0 * (lx0 * a[0] + lx1 * b[0]) + ly1 * (lx0 * c[0] + lx1 * d[0])) * 2.0f;
    fy = (ly0 * (lx0 * a[1] + lx1 * b[1]) + ly1 * (lx0 * c[1] + lx1 * d[1])) * 2.0f;
  }
  // grid_sample(bilinear, border, align_corners=True) through the normalise / unnormalise round trip
  const float gx = (float)x + fx, gy = (float)y + fy;
  const float wm = (float)(W - 1 > 1 ? W - 1 : 1), hm = (float)(H - 1 > 1 ? H - 1 : 1);
  const float nx = 2.0f * gx / wm - 1.0f, ny = 2.0f * gy / hm - 1.0f;
  float ix = ((nx + 1.f) / 2.f) * (float)(W - 1), iy = ((ny + 1.f) / 2.f) * (float)(H - 1);
  ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
  iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
  const float xw = floorf(ix), yn = floorf(iy);
  const int ix0 = (int)xw, iy0 = (int)yn, ix1 = ix0 + 1, iy1 = iy0 + 1;
  const float w_nw = ((float)ix1 - ix) * ((float)iy1 - iy), w_ne = (ix - (float)ix0) * ((float)iy1 - iy);
  const float w_sw = ((float)ix1 - ix) * (iy - (float)iy0), w_se = (ix - (float)ix0) * (iy - (float)iy0);
  const float* sp = reinterpret_cast<const float*>(supp.p) + (long)n * supp.sn;
  float wv[3] = {0.f, 0.f, 0.f};
  const bool x0ok = ix0 >= 0 && ix0 < W, x1ok = ix1 >= 0 && ix1 < W, y0ok = iy0 >= 0 && iy0 < H, y1ok = iy1 >= 0 && iy1 < H;
  const float* rp = reinterpret_cast<const float*>(ref.p) + (long)n * ref.sn + pix * ref.sp;
  float r3[3];
  if constexpr (VEC) {
    // the four corners as whole pixels at clamped addresses, weights zeroed where the reference's bounds tests fail
    const int cx0 = min(max(ix0, 0), W - 1), cx1 = min(max(ix1, 0), W - 1), cy0 = min(max(iy0, 0), H - 1), cy1 = min(max(iy1, 0), H - 1);
    const f32x4 nw = *reinterpret_cast<const f32x4*>(sp + ((long)cy0 * W + cx0) * supp.sp);
    const f32x4 ne = *reinterpret_cast<const f32x4*>(sp + ((long)cy0 * W + cx1) * supp.sp);
    const f32x4 sw = *reinterpret_cast<const f32x4*>(sp + ((long)cy1 * W + cx0) * supp.sp);
    const f32x4 se = *reinterpret_cast<const f32x4*>(sp + ((long)cy1 * W + cx1) * supp.sp);
    const f32x4 rr = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float s = 0.f;
      if (y0ok && x0ok) s += nw[c] * w_nw;
      if (y0ok && x1ok) s += ne[c] * w_ne;
      if (y1ok && x0ok) s += sw[c] * w_sw;
      if (y1ok && x1ok) s += se[c] * w_se;
      wv[c] = s;
      r3[c] = rr[c];
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float s = 0.f;
      if (y0ok && x0ok) s += sp[((long)iy0 * W + ix0) * supp.sp + c] * w_nw;
      if (y0ok && x1ok) s += sp[((long)iy0 * W + ix1) * supp.sp + c] * w_ne;
      if (y1ok && x0ok) s += sp[((long)iy1 * W + ix0) * supp.sp + c] * w_sw;
      if (y1ok && x1ok) s += sp[((long)iy1 * W + ix1) * supp.sp + c] * w_se;
      wv[c] = s;
      r3[c] = rp[c];
    }
  }
  float v[8] = {r3[0], r3[1], r3[2], wv[0], wv[1], wv[2], fx, fy};
  store8(cat8, n, pix, 0, v);
  float* fu = reinterpret_cast<float*>(flow_up.p) + (long)n * flow_up.sn + pix * flow_up.sp;
  if constexpr (VEC) {
    *reinterpret_cast<f32x2*>(fu) = f32x2{fx, fy};
  } else {
    fu[0] = fx;
    fu[1] = fy;
  }
}

__global__ void resize_bilinear_kernel(FMap x, FMap y, const float* chscale) {
  const long npix = (long)y.H * y.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * y.N) return;
  const int n = (int)(i / npix);
  const long pix = i % npix;
  const int oy = (int)(pix / y.W), ox = (int)(pix % y.W);
  const float rh = (float)x.H / (float)y.H, rw = (float)x.W / (float)y.W;
  const float sy = fmaxf(rh * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(rw * (ox + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy < x.H - 1 ? (int)sy : x.H - 1, x0 = (int)sx < x.W - 1 ? (int)sx : x.W - 1;
  const int y1 = y0 + (y0 < x.H - 1), x1 = x0 + (x0 < x.W - 1);
  const float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const float* xp = reinterpret_cast<const float*>(x.p) + (long)n * x.sn;
  float* yp = reinterpret_cast<float*>(y.p) + (long)n * y.sn + pix * y.sp;
  for (int c = 0; c < y.C; ++c) {
    const float a = xp[((long)y0 * x.W + x0) * x.sp + c], b = xp[((long)y0 * x.W + x1) * x.sp + c];
    const float cc = xp[((long)y1 * x.W + x0) * x.sp + c], d = xp[((long)y1 * x.W + x1) * x.sp + c];
    float v = ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * cc + lx1 * d);
    if (chscale) v *= chscale[c];
    yp[c] = v;
  }
}

// ------------------------------------------------------------------ in-loop filter matching
// Stage 1: one block per (pooled cell, strip of POOL_ROWS rows): channel sums of the strip -> work[cell][strip][C].
// Stage 2: one thread per (cell, channel): strips summed in order, / area.  (The FeatureFix pooling at 1080p is
// 136 x 136 pixels per cell and only 8 x 14 cells: one block per cell left 144 of 256 CUs idle and ran 0.8 TB/s.)
constexpr int POOL_ROWS = 8;
__global__ __launch_bounds__(256) void avgpool_k_strip_kernel(FMap x, int scale, float* work, int wp, int nsplit) {
  __shared__ float red[256][9];
  const int chunks = x.C / 8;
  const int lanes = 256 / chunks;
  const int tid = threadIdx.x;
  const int ck = tid % chunks, pl = tid / chunks;
  const int cell = blockIdx.x / nsplit, strip = blockIdx.x - cell * nsplit, n = blockIdx.y;
  const int py = cell / wp, px = cell % wp;
  const int r0 = strip * POOL_ROWS, rows = min(POOL_ROWS, scale - r0);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int area = rows * scale;
  if (pl < lanes) {
    int k = pl;                          // four independent 16-byte loads in flight per thread
    for (; k + 3 * lanes < area; k += 4 * lanes) {
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = k + u * lanes;
        load8(x, n, (long)(py * scale + r0 + kk / scale) * x.W + (px * scale + kk % scale), ck * 8, v[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[u][j];
    }
    for (; k < area; k += lanes) {
      float v[8];
      load8(x, n, (long)(py * scale + r0 + k / scale) * x.W + (px * scale + k % scale), ck * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tid][j] = (pl < lanes) ? acc[j] : 0.f;
  __syncthreads();
  if (tid < x.C) {
    const int cc = tid / 8, j = tid % 8;
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[l * chunks + cc][j];
    work[(((long)n * gridDim.x) + blockIdx.x) * x.C + tid] = s;
  }
}

__global__ void avgpool_k_final_kernel(const float* work, float* pooled, long total, int C_, int nsplit, float area) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long cell = i / C_;
  const int c = (int)(i - cell * C_);
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += work[(cell * nsplit + k) * C_ + c];
  pooled[i] = s / area;
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// patch (i,j) of a pooled map: rows 3i-3..3i-1, cols 3j-3..3j-1, zero outside (unfold k=3,s=3,p=3)
__device__ __forceinline__ float patch_elem(const float* pm, int hp, int wp, int C, int pi, int pj, int d) {
  const int c = d / 9, k = d % 9;
  const int yy = 3 * pi - 3 + k / 3, xx = 3 * pj - 3 + k % 3;
  if (yy < 0 || yy >= hp || xx < 0 || xx >= wp) return 0.f;
  return pm[((long)yy * wp + xx) * C + c];
}

// grid (L, N); block 256.  idx[n][l] = argmax_m <a_l/|a_l|, b_m/|b_m|>, first index on ties
__global__ __launch_bounds__(256) void patch_match_kernel(const float* pin, const float* pref, int hp, int wp, int C,
                                                           int nph, int npw, int32_t* idx) {
  extern __shared__ float sm[];
  float* a = sm;                  // [C*9] normalised input patch
  __shared__ float sh[4];
  const int D = C * 9, L = nph * npw;
  const int l = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const float* pa = pin + (long)n * hp * wp * C;
  const float* pb = pref + (long)n * hp * wp * C;
  float ss = 0.f;
  for (int d = tid; d < D; d += 256) {
    const float v = patch_elem(pa, hp, wp, C, l / npw, l % npw, d);
    a[d] = v;
    ss += v * v;
  }
  const float na = fmaxf(sqrtf(block_sum(ss, sh)), 1e-12f);
  for (int d = tid; d < D; d += 256) a[d] = a[d] / na;
  __syncthreads();
  float best = -INFINITY;
  int besti = 0;
  for (int m = 0; m < L; ++m) {
    float sb = 0.f;
    for (int d = tid; d < D; d += 256) {
      const float v = patch_elem(pb, hp, wp, C, m / npw, m % npw, d);
      sb += v * v;
    }
    const float nb = fmaxf(sqrtf(block_sum(sb, sh)), 1e-12f);
    float dot = 0.f;
    for (int d = tid; d < D; d += 256) dot += a[d] * (patch_elem(pb, hp, wp, C, m / npw, m % npw, d) / nb);
    const float s = block_sum(dot, sh);
    if (s > best) { best = s; besti = m; }
  }
  if (tid == 0) idx[(long)n * L + l] = besti;
}

// 8 lanes per pixel (8 channels each, C = 64): gather matched block, cosine weight, write cat
__global__ __launch_bounds__(256) void match_gather_kernel(FMap fin, FMap fref, const int32_t* idx, int ks, int nbh, int nbw, FMap cat) {
  const int lanes = fin.C / 8;            // 8
  const long npix = (long)fin.H * fin.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long t = i / lanes;
  const int c = (int)(i % lanes) * 8;
  const bool valid = t < npix * fin.N;
  const long tc = valid ? t : 0;
  const int n = (int)(tc / npix);
  const long pix = tc % npix;
  const int y = (int)(pix / fin.W), x = (int)(pix % fin.W);
  const int bi = y / ks + 1, bj = x / ks + 1;
  const int m = idx[(long)n * nbh * nbw + bi * nbw + bj];
  const int sy = (m / nbw - 1) * ks + y % ks, sx = (m % nbw - 1) * ks + x % ks;
  float a[8], b[8];
  load8(fin, n, pix, c, a);
  if (sy >= 0 && sy < fref.H && sx >= 0 && sx < fref.W) {
    load8(fref, n, (long)sy * fref.W + sx, c, b);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = 0.f;
  }
  float w12 = 0.f, w1 = 0.f, w2 = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { w12 += a[j] * b[j]; w1 += a[j] * a[j]; w2 += b[j] * b[j]; }
  for (int o = 1; o < lanes; o <<= 1) {
    w12 += __shfl_xor(w12, o);
    w1 += __shfl_xor(w1, o);
    w2 += __shfl_xor(w2, o);
  }
  const float cor = w12 / fmaxf(sqrtf(w1) * sqrtf(w2), 1e-8f);
  if (!valid) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] *= cor; b[j] *= cor; }
  store8(cat, n, pix, c, a);
  store8(cat, n, pix, fin.C + c, b);
}

// ------------------------------------------------------------------ entropy model
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void eb_forward_kernel(FMap z, const float* params, FMap noise, FMap z_hat, float* partial) {
  __shared__ float sh[4];
  const long npix = (long)z.H * z.W;
  const long total = npix * z.C * z.N;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float bits = 0.f;
  if (i < total) {
    const int c = (int)(i % z.C);
    const long t = i / z.C;
    const long pix = t % npix;
    const int n = (int)(t / npix);
    const float* P = params + (long)c * EB_NP;
    const float zv = reinterpret_cast<const float*>(z.p)[(long)n * z.sn + pix * z.sp + c];
    float v;
    if (noise.p) {
      v = zv + reinterpret_cast<const float*>(noise.p)[(long)n * noise.sn + pix * noise.sp + c];
    } else {
      const float med = P[58];
      v = rintf(zv - med) + med;
    }
    const float lower = eb_logits(P, v - 0.5f), upper = eb_logits(P, v + 0.5f);
    const float s = lower + upper;
    const float sign = s > 0.f ? -1.f : (s < 0.f ? 1.f : 0.f);
    float lik = fabsf(sigmoidf_(sign * upper) - sigmoidf_(sign * lower));
    lik = fmaxf(lik, 1e-9f);
    bits = -log2f(lik);
    if (z_hat.f32)
      reinterpret_cast<float*>(z_hat.p)[(long)n * z_hat.sn + pix * z_hat.sp + c] = v;
    else
      reinterpret_cast<half_t*>(z_hat.p)[(long)n * z_hat.sn + pix * z_hat.sp + c] = (half_t)v;
  }
  const float s = block_sum(bits, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void gc_forward_kernel(FMap y, FMap gp, FMap noise, float* partial) {
  __shared__ float sh[4];
  const long npix = (long)y.H * y.W;
  const long total = npix * y.C * y.N;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float bits = 0.f;
  if (i < total) {
    const int c = (int)(i % y.C);
    const long t = i / y.C;
    const long pix = t % npix;
    const int n = (int)(t / npix);
    const float yv = reinterpret_cast<const float*>(y.p)[(long)n * y.sn + pix * y.sp + c];
    const float* g = reinterpret_cast<const float*>(gp.p) + (long)n * gp.sn + pix * gp.sp;
    const float scale = fmaxf(g[c], 0.11f), mean = g[y.C + c];
    float v;
    if (noise.p)
      v = yv + reinterpret_cast<const float*>(noise.p)[(long)n * noise.sn + pix * noise.sp + c] - mean;
    else
      v = rintf(yv - mean);
    v = fabsf(v);
    const float k = 0.70710678118654752440f;
    const float upper = 0.5f * erfcf(-k * ((0.5f - v) / scale));
    const float lower = 0.5f * erfcf(-k * ((-0.5f - v) / scale));
    const float lik = fmaxf(upper - lower, 1e-9f);
    bits = -log2f(lik);
  }
  const float s = block_sum(bits, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void final_sum_kernel(const float* partial, int n, double* out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

__global__ void quantize_kernel(FMap y, FMap noise, FMap y_hat) {
  const int chunks = (y.C + 7) / 8;
  const long npix = (long)y.H * y.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * chunks * y.N) return;
  const int c = (int)(i % chunks) * 8;
  const long t = i / chunks;
  const long pix = t % npix;
  const int n = (int)(t / npix);
  float v[8];
  load8(y, n, pix, c, v);
  if (noise.p) {
    float q[8];
    load8(noise, n, pix, c, q);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += q[j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = rintf(v[j]);
  }
  store8(y_hat, n, pix, c, v);
}

}  // namespace

extern "C" int tdvc_nchw_to_fmap(const float* src, int C, const tdvc_fmap* y, void* stream) {
  TDVC_CHECK(src && y && fmap_any(*y) && C >= 1 && C <= y->C, "tdvc_nchw_to_fmap: bad arguments");
  const long total = (long)y->N * y->H * y->W;
  hipLaunchKernelGGL(nchw_to_fmap_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), src, C, to_dev(*y));
  return tdvc_launch_status("tdvc_nchw_to_fmap");
}

extern "C" int tdvc_fmap_to_nchw(const tdvc_fmap* x, int C, float* dst, void* stream) {
  TDVC_CHECK(dst && x && fmap_any(*x) && C >= 1 && C <= x->C, "tdvc_fmap_to_nchw: bad arguments");
  const long total = (long)x->N * x->H * x->W;
  hipLaunchKernelGGL(fmap_to_nchw_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*x), C, dst);
  return tdvc_launch_status("tdvc_fmap_to_nchw");
}

extern "C" int tdvc_scale_act_res(const tdvc_fmap* a, const float* gate, int act, float slope,
                                  const tdvc_fmap* r, float r_sign, const tdvc_fmap* y, const tdvc_fmap* y2, void* stream) {
  TDVC_CHECK(a && y && fmap_any(*a) && fmap_any(*y) && same_geom(*a, *y) && y->C == a->C, "tdvc_scale_act_res: bad a/y");
  if (r) TDVC_CHECK(fmap_any(*r) && same_geom(*a, *r) && r->C == a->C, "tdvc_scale_act_res: bad residual");
  if (y2) TDVC_CHECK(fmap_any(*y2) && same_geom(*a, *y2) && y2->C == a->C, "tdvc_scale_act_res: bad y2");
  const long total = (long)a->N * a->H * a->W * ((a->C + 7) / 8);
  const bool all16 = a->dtype == TDVC_F16 && y->dtype == TDVC_F16 && (!r || r->dtype == TDVC_F16) && (!y2 || y2->dtype == TDVC_F16) &&
                     (!gate || aligned16(gate)) && total < (1L << 31) - (1L << 23);
  if (all16) {
    constexpr int U = 4;
    long blocks = (total + 256 * U - 1) / (256 * U);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_act_res16_kernel<U>, dim3((unsigned)blocks), dim3(256), 0, ST(stream), to_dev(*a), gate, act, slope,
                       r ? to_dev(*r) : null_fmap(), r_sign, to_dev(*y), y2 ? to_dev(*y2) : null_fmap(), (unsigned)total, (unsigned)(a->C / 8),
                       (unsigned)((long)a->H * a->W));
    return tdvc_launch_status("tdvc_scale_act_res");
  }
  hipLaunchKernelGGL(scale_act_res_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*a), gate, act, slope,
                     r ? to_dev(*r) : null_fmap(), r_sign, to_dev(*y), y2 ? to_dev(*y2) : null_fmap());
  return tdvc_launch_status("tdvc_scale_act_res");
}

extern "C" int tdvc_add_flow(const tdvc_fmap* off, const tdvc_fmap* flow, void* stream) {
  TDVC_CHECK(off && flow && fmap_ok16(*off) && fmap_ok32(*flow) && flow->C >= 2 && same_geom(*off, *flow), "tdvc_add_flow: bad arguments");
  const long total = (long)off->N * off->H * off->W * (off->C / 8);
  hipLaunchKernelGGL(add_flow_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*off), to_dev(*flow));
  return tdvc_launch_status("tdvc_add_flow");
}

extern "C" int tdvc_bcast_add_act(const tdvc_fmap* x, const tdvc_fmap* b, int T, float slope, void* stream) {
  TDVC_CHECK(x && b && fmap_ok16(*x) && fmap_ok16(*b) && same_geom(*x, *b) && x->C == T * b->C, "tdvc_bcast_add_act: bad arguments");
  const long total = (long)x->N * x->H * x->W * (x->C / 8);
  if (T == 4 && total / 4 < (1L << 31)) {
    const long items = total / 4;
    hipLaunchKernelGGL(bcast_add_act_t_kernel<4>, grid1d(items), dim3(256), 0, ST(stream), to_dev(*x), to_dev(*b), slope, (unsigned)items,
                       (unsigned)(b->C / 8), (unsigned)((long)x->H * x->W));
    return tdvc_launch_status("tdvc_bcast_add_act");
  }
  hipLaunchKernelGGL(bcast_add_act_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*x), to_dev(*b), slope);
  return tdvc_launch_status("tdvc_bcast_add_act");
}

extern "C" int tdvc_channel_sum(const tdvc_fmap* x, float* partial, int nblocks, void* stream) {
  TDVC_CHECK(x && partial && fmap_any(*x) && (x->C % 8) == 0 && x->C <= 256 && nblocks >= 1 && nblocks <= 4096, "tdvc_channel_sum: bad arguments");
  hipLaunchKernelGGL(channel_sum_kernel, dim3(nblocks, x->N), dim3(256), 0, ST(stream), to_dev(*x), partial, nblocks);
  return tdvc_launch_status("tdvc_channel_sum");
}

extern "C" int tdvc_se_gate(const float* partial, int nblocks, float inv_count, int N, int C, int Cmid,
                            const float* w1, const float* b1, const float* w2, const float* b2, float* gate, void* stream) {
  TDVC_CHECK(partial && w1 && b1 && w2 && b2 && gate && C <= 256 && Cmid <= 32 && N >= 1, "tdvc_se_gate: bad arguments");
  hipLaunchKernelGGL(se_gate_kernel, dim3(N), dim3(1024), 0, ST(stream), partial, nblocks, inv_count, C, Cmid, w1, b1, w2, b2, gate);
  return tdvc_launch_status("tdvc_se_gate");
}

extern "C" int tdvc_upsample2x(const tdvc_fmap* x, const tdvc_fmap* y, void* stream) {
  TDVC_CHECK(x && y && fmap_ok16(*x) && fmap_ok16(*y) && y->H == 2 * x->H && y->W == 2 * x->W && y->C == x->C && y->N == x->N,
             "tdvc_upsample2x: bad arguments");
  const long total = (long)y->N * y->H * y->W * (y->C / 8);
  hipLaunchKernelGGL(upsample2x_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*x), to_dev(*y));
  return tdvc_launch_status("tdvc_upsample2x");
}

extern "C" int tdvc_avgpool2(const tdvc_fmap* x, const tdvc_fmap* y, void* stream) {
  TDVC_CHECK(x && y && fmap_ok32(*x) && fmap_ok32(*y) && y->H == x->H / 2 && y->W == x->W / 2 && y->C <= x->C && y->N == x->N,
             "tdvc_avgpool2: bad arguments");
  const long total = (long)y->N * y->H * y->W;
  hipLaunchKernelGGL(avgpool2_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*x), to_dev(*y));
  return tdvc_launch_status("tdvc_avgpool2");
}

extern "C" int tdvc_spynet_level_input(const tdvc_fmap* ref, const tdvc_fmap* supp, const tdvc_fmap* flow_lo,
                                       const tdvc_fmap* flow_up, const tdvc_fmap* cat8, void* stream) {
  TDVC_CHECK(ref && supp && flow_up && cat8, "tdvc_spynet_level_input: null");
  TDVC_CHECK(fmap_ok32(*ref) && fmap_ok32(*supp) && ref->C >= 3 && supp->C >= 3 && same_geom(*ref, *supp), "tdvc_spynet_level_input: bad ref/supp");
  TDVC_CHECK(fmap_ok32(*flow_up) && flow_up->C >= 2 && same_geom(*ref, *flow_up), "tdvc_spynet_level_input: bad flow_up");
  TDVC_CHECK(fmap_ok16(*cat8) && cat8->C == 8 && same_geom(*ref, *cat8), "tdvc_spynet_level_input: bad cat8");
  if (flow_lo) TDVC_CHECK(fmap_ok32(*flow_lo) && flow_lo->C >= 2 && flow_lo->N == ref->N && flow_lo->H * 2 == ref->H && flow_lo->W * 2 == ref->W,
                          "tdvc_spynet_level_input: bad flow_lo");
  const long total = (long)ref->N * ref->H * ref->W;
  auto pix16 = [](const tdvc_fmap& f) { return f.C >= 4 && (f.sp % 4) == 0 && (f.sn % 4) == 0 && aligned16(f.p); };             // whole 4-float pixels
  auto pix8 = [](const tdvc_fmap& f) { return (f.sp % 2) == 0 && (f.sn % 2) == 0 && (((uintptr_t)f.p) & 7) == 0; };             // whole 2-float pixels
  const bool vec = pix16(*ref) && pix16(*supp) && pix8(*flow_up) && (!flow_lo || pix8(*flow_lo));
  if (vec)
    hipLaunchKernelGGL(spynet_level_input_kernel<true>, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*ref), to_dev(*supp),
                       flow_lo ? to_dev(*flow_lo) : null_fmap(), to_dev(*flow_up), to_dev(*cat8));
  else
    hipLaunchKernelGGL(spynet_level_input_kernel<false>, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*ref), to_dev(*supp),
                       flow_lo ? to_dev(*flow_lo) : null_fmap(), to_dev(*flow_up), to_dev(*cat8));
  return tdvc_launch_status("tdvc_spynet_level_input");
}

extern "C" int tdvc_resize_bilinear(const tdvc_fmap* x, const tdvc_fmap* y, const float* chscale, void* stream) {
  TDVC_CHECK(x && y && fmap_ok32(*x) && fmap_ok32(*y) && y->C <= x->C && y->N == x->N, "tdvc_resize_bilinear: bad arguments");
  const long total = (long)y->N * y->H * y->W;
  hipLaunchKernelGGL(resize_bilinear_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*x), to_dev(*y), chscale);
  return tdvc_launch_status("tdvc_resize_bilinear");
}

extern "C" int64_t tdvc_avgpool_k_work_floats(int N, int hp, int wp, int C_, int scale) {
  return (int64_t)N * hp * wp * ((scale + POOL_ROWS - 1) / POOL_ROWS) * C_;
}

extern "C" int tdvc_avgpool_k(const tdvc_fmap* x, int scale, float* pooled, int hp, int wp, float* work, int64_t work_floats, void* stream) {
  TDVC_CHECK(x && pooled && work && fmap_ok16(*x) && x->C <= 256 && scale >= 1 && hp == x->H / scale && wp == x->W / scale && hp >= 1 && wp >= 1,
             "tdvc_avgpool_k: bad arguments");
  const int nsplit = (scale + POOL_ROWS - 1) / POOL_ROWS;
  TDVC_CHECK(work_floats >= tdvc_avgpool_k_work_floats(x->N, hp, wp, x->C, scale), "tdvc_avgpool_k: workspace too small (tdvc_avgpool_k_work_floats)");
  hipLaunchKernelGGL(avgpool_k_strip_kernel, dim3(hp * wp * nsplit, x->N), dim3(256), 0, ST(stream), to_dev(*x), scale, work, wp, nsplit);
  const long total = (long)x->N * hp * wp * x->C;
  hipLaunchKernelGGL(avgpool_k_final_kernel, grid1d(total), dim3(256), 0, ST(stream), work, pooled, total, x->C, nsplit, (float)scale * (float)scale);
  return tdvc_launch_status("tdvc_avgpool_k");
}

extern "C" int tdvc_patch_match(const float* pin, const float* pref, int N, int hp, int wp, int C, int32_t* idx, void* stream) {
  TDVC_CHECK(pin && pref && idx && N >= 1 && hp >= 1 && wp >= 1 && C >= 1 && C * 9 * 4 <= 60000, "tdvc_patch_match: bad arguments");
  const int nph = (hp + 3) / 3 + 1, npw = (wp + 3) / 3 + 1;
  hipLaunchKernelGGL(patch_match_kernel, dim3(nph * npw, N), dim3(256), (size_t)C * 9 * 4, ST(stream), pin, pref, hp, wp, C, nph, npw, idx);
  return tdvc_launch_status("tdvc_patch_match");
}

extern "C" int tdvc_match_gather(const tdvc_fmap* fin, const tdvc_fmap* fref, const int32_t* idx, int scale, int hp, int wp,
                                 const tdvc_fmap* cat, void* stream) {
  TDVC_CHECK(fin && fref && idx && cat && fmap_ok16(*fin) && fmap_ok16(*fref) && fmap_ok16(*cat), "tdvc_match_gather: bad fmaps");
  TDVC_CHECK(fin->C == 64 && fref->C == 64 && cat->C == 128 && same_geom(*fin, *fref) && same_geom(*fin, *cat), "tdvc_match_gather: needs C=64 / cat C=128");
  const int ks = 3 * scale;
  const int nbh = (fin->H + ks) / ks + 1, nbw = (fin->W + ks) / ks + 1;   // fold(kernel=ks, pad=ks, stride=ks)
  TDVC_CHECK(nbh == (hp + 3) / 3 + 1 && nbw == (wp + 3) / 3 + 1,
             "tdvc_match_gather: fold grid %dx%d != patch grid %dx%d (F.fold would raise, pnet.py:252)", nbh, nbw, (hp + 3) / 3 + 1, (wp + 3) / 3 + 1);
  const long total = (long)fin->N * fin->H * fin->W * 8;
  hipLaunchKernelGGL(match_gather_kernel, grid1d(total), dim3(256), 0, ST(stream), to_dev(*fin), to_dev(*fref), idx, ks, nbh, nbw, to_dev(*cat));
  return tdvc_launch_status("tdvc_match_gather");
}

extern "C" int tdvc_eb_forward(const tdvc_fmap* z, const float* params, const tdvc_fmap* noise,
                               const tdvc_fmap* z_hat, double* bits_out, float* partial, int partial_cap, void* stream) {
  TDVC_CHECK(z && params && z_hat && bits_out && partial && fmap_ok32(*z) && fmap_any(*z_hat) && same_geom(*z, *z_hat) && z_hat->C >= z->C,
             "tdvc_eb_forward: bad arguments");
  if (noise) TDVC_CHECK(fmap_ok32(*noise) && same_geom(*z, *noise) && noise->C >= z->C, "tdvc_eb_forward: bad noise");
  const long total = (long)z->N * z->H * z->W * z->C;
  const int nb = (int)((total + 255) / 256);
  TDVC_CHECK(nb <= partial_cap, "tdvc_eb_forward: partial buffer too small (%d < %d)", partial_cap, nb);
  hipLaunchKernelGGL(eb_forward_kernel, dim3(nb), dim3(256), 0, ST(stream), to_dev(*z), params, noise ? to_dev(*noise) : null_fmap(), to_dev(*z_hat), partial);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, ST(stream), partial, nb, bits_out);
  return tdvc_launch_status("tdvc_eb_forward");
}

extern "C" int tdvc_gc_forward(const tdvc_fmap* y, const tdvc_fmap* gp, const tdvc_fmap* noise,
                               double* bits_out, float* partial, int partial_cap, void* stream) {
  TDVC_CHECK(y && gp && bits_out && partial && fmap_ok32(*y) && fmap_ok32(*gp) && same_geom(*y, *gp) && gp->C >= 2 * y->C, "tdvc_gc_forward: bad arguments");
  if (noise) TDVC_CHECK(fmap_ok32(*noise) && same_geom(*y, *noise) && noise->C >= y->C, "tdvc_gc_forward: bad noise");
  const long total = (long)y->N * y->H * y->W * y->C;
  const int nb = (int)((total + 255) / 256);
  TDVC_CHECK(nb <= partial_cap, "tdvc_gc_forward: partial buffer too small (%d < %d)", partial_cap, nb);
  hipLaunchKernelGGL(gc_forward_kernel, dim3(nb), dim3(256), 0, ST(stream), to_dev(*y), to_dev(*gp), noise ? to_dev(*noise) : null_fmap(), partial);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, ST(stream), partial, nb, bits_out);
  return tdvc_launch_status("tdvc_gc_forward");
}

extern "C" int tdvc_quantize(const tdvc_fmap* y, const tdvc_fmap* noise, const tdvc_fmap* y_hat, void* stream) {
  TDVC_CHECK(y && y_hat && fmap_any(*y) && fmap_any(*y_hat) && same_geom(*y, *y_hat) && y_hat->C >= y->C, "tdvc_quantize: bad arguments");
  if (noise) TDVC_CHECK(fmap_any(*noise) && same_geom(*y, *noise) && noise->C >= y->C, "tdvc_quantize: bad noise");
  const long total = (long)y->N * y->H * y->W * ((y->C + 7) / 8);
  hipLaunchKernelGGL(quantize_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*y), noise ? to_dev(*noise) : null_fmap(), to_dev(*y_hat));
  return tdvc_launch_status("tdvc_quantize");
}
