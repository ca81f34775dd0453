// `_ext.dcn_v2_backward` for gfx950 — fp32 NCHW, same per-sample structure as the reference
// (src/cuda/dcn_v2_cuda.cu:97-216): columns = W^T dY ; col2im_coord -> dOffset, dMask ; col2im ->
// dInput ; im2col again ; dW += dY columns^T ; db += dY 1.   Kernels follow
// src/cuda/dcn_v2_im2col_cuda.cu:56-123 (weights), :197-254 (col2im), :256-327 (col2im_coord).
// dInput is a float-atomic scatter like the reference's (summation order is not reproducible run to
// run; everything else is deterministic).  `columns` is caller-provided scratch, reused per sample.
#include "common.h"

namespace {

struct Geo {
  int C, H, W, Ho, Wo, kh, kw, ph, pw, sh, sw, dh, dw, G, Cout;
};

__device__ __forceinline__ float bilin(const float* im, int H, int W, float h, float w) {
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h - h_low, lw = w - w_low, hh = 1.f - lh, hw = 1.f - lw;
  float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
  if (h_low >= 0 && w_low >= 0) v1 = im[h_low * W + w_low];
  if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * W + w_high];
  if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * W + w_low];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * W + w_high];
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

__device__ __forceinline__ float coord_weight(float ah, float aw, int H, int W, const float* im, int dir) {
  if (ah <= -1.f || ah >= (float)H || aw <= -1.f || aw >= (float)W) return 0.f;
  const int hl = (int)floorf(ah), wl = (int)floorf(aw), hh = hl + 1, wh = wl + 1;
  float wt = 0.f;
  if (dir == 0) {
    if (hl >= 0 && wl >= 0) wt += -1.f * (wl + 1 - aw) * im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) wt += -1.f * (aw - wl) * im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) wt += (wl + 1 - aw) * im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) wt += (aw - wl) * im[hh * W + wh];
  } else {
    if (hl >= 0 && wl >= 0) wt += -1.f * (hl + 1 - ah) * im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) wt += (hl + 1 - ah) * im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) wt += -1.f * (ah - hl) * im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) wt += (ah - hl) * im[hh * W + wh];
  }
  return wt;
}

// columns[k][p] = sum_co W[co][k] * dY[co][p]        (k = c*K + t)
__global__ void dcol_kernel(const float* __restrict__ w, const float* __restrict__ gy, float* __restrict__ col, int CK, int P, int Cout) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)CK * P) return;
  const int k = (int)(i / P), p = (int)(i % P);
  float s = 0.f;
  for (int co = 0; co < Cout; ++co) s += w[(long)co * CK + k] * gy[(long)co * P + p];
  col[i] = s;
}

// one thread per (deformable group, tap, pixel): both offset directions + the mask gradient
__global__ void coord_kernel(const float* __restrict__ dcol, const float* __restrict__ im, const float* __restrict__ off,
                             const float* __restrict__ msk, Geo g, float* __restrict__ goff, float* __restrict__ gmsk) {
  const int K = g.kh * g.kw, P = g.Ho * g.Wo, cpg = g.C / g.G;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)g.G * K * P) return;
  const int p = (int)(i % P), t = (int)((i / P) % K), dg = (int)(i / ((long)P * K));
  const int ho = p / g.Wo, wo = p % g.Wo, ti = t / g.kw, tj = t % g.kw;
  const float oh = off[((long)dg * 2 * K + 2 * t) * P + p], ow = off[((long)dg * 2 * K + 2 * t + 1) * P + p];
  const float m = msk[((long)dg * K + t) * P + p];
  float ih = (float)(ho * g.sh - g.ph + ti * g.dh) + oh, iw = (float)(wo * g.sw - g.pw + tj * g.dw) + ow;
  const bool outside = ih <= -1.f || iw <= -1.f || ih >= (float)g.H || iw >= (float)g.W;
  if (outside) ih = iw = -2.f;
  float vh = 0.f, vw = 0.f, mval = 0.f;
  for (int cl = 0; cl < cpg; ++cl) {
    const int c = dg * cpg + cl;
    const float* imc = im + (long)c * g.H * g.W;
    const float d = dcol[((long)c * K + t) * P + p];
    if (!outside) mval += d * bilin(imc, g.H, g.W, ih, iw);
    vh += coord_weight(ih, iw, g.H, g.W, imc, 0) * d * m;
    vw += coord_weight(ih, iw, g.H, g.W, imc, 1) * d * m;
  }
  goff[((long)dg * 2 * K + 2 * t) * P + p] = vh;
  goff[((long)dg * 2 * K + 2 * t + 1) * P + p] = vw;
  gmsk[((long)dg * K + t) * P + p] = mval;
}

// one thread per (channel, tap, pixel): scatter to the (at most) 4 bilinear corners
__global__ void col2im_kernel(const float* __restrict__ dcol, const float* __restrict__ off, const float* __restrict__ msk, Geo g,
                              float* __restrict__ gim) {
  const int K = g.kh * g.kw, P = g.Ho * g.Wo, cpg = g.C / g.G;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)g.C * K * P) return;
  const int p = (int)(i % P), t = (int)((i / P) % K), c = (int)(i / ((long)P * K));
  const int dg = c / cpg, ho = p / g.Wo, wo = p % g.Wo, ti = t / g.kw, tj = t % g.kw;
  const float oh = off[((long)dg * 2 * K + 2 * t) * P + p], ow = off[((long)dg * 2 * K + 2 * t + 1) * P + p];
  const float m = msk[((long)dg * K + t) * P + p];
  const float ih = (float)(ho * g.sh - g.ph + ti * g.dh) + oh, iw = (float)(wo * g.sw - g.pw + tj * g.dw) + ow;
  if (ih <= -1.f || ih >= (float)g.H || iw <= -1.f || iw >= (float)g.W) return;
  const float top = dcol[i] * m;
  const int hl = (int)floorf(ih), wl = (int)floorf(iw), hh = hl + 1, wh = wl + 1;
  float* gc = gim + (long)c * g.H * g.W;
  if (hl >= 0 && wl >= 0) atomicAdd(gc + hl * g.W + wl, (hl + 1 - ih) * (wl + 1 - iw) * top);
  if (hl >= 0 && wh <= g.W - 1) atomicAdd(gc + hl * g.W + wh, (hl + 1 - ih) * (iw + 1 - wh) * top);
  if (hh <= g.H - 1 && wl >= 0) atomicAdd(gc + hh * g.W + wl, (ih + 1 - hh) * (wl + 1 - iw) * top);
  if (hh <= g.H - 1 && wh <= g.W - 1) atomicAdd(gc + hh * g.W + wh, (ih + 1 - hh) * (iw + 1 - wh) * top);
}

__global__ void im2col_kernel(const float* __restrict__ im, const float* __restrict__ off, const float* __restrict__ msk, Geo g,
                              float* __restrict__ col) {
  const int K = g.kh * g.kw, P = g.Ho * g.Wo, cpg = g.C / g.G;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)g.C * K * P) return;
  const int p = (int)(i % P), t = (int)((i / P) % K), c = (int)(i / ((long)P * K));
  const int dg = c / cpg, ho = p / g.Wo, wo = p % g.Wo, ti = t / g.kw, tj = t % g.kw;
  const float oh = off[((long)dg * 2 * K + 2 * t) * P + p], ow = off[((long)dg * 2 * K + 2 * t + 1) * P + p];
  const float m = msk[((long)dg * K + t) * P + p];
  const float ih = (float)(ho * g.sh - g.ph + ti * g.dh) + oh, iw = (float)(wo * g.sw - g.pw + tj * g.dw) + ow;
  float v = 0.f;
  if (ih > -1.f && iw > -1.f && ih < (float)g.H && iw < (float)g.W) v = bilin(im + (long)c * g.H * g.W, g.H, g.W, ih, iw);
  col[i] = v * m;
}

// dW[co][k] += sum_p dY[co][p] * col[k][p] ; one workgroup per k, LDS tree over p, all co in the block
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ col, float* __restrict__ gw,
                                                     int CK, int P, int Cout) {
  __shared__ float red[256];
  const int k = blockIdx.x, tid = threadIdx.x;
  const float* ck = col + (long)k * P;
  for (int co = 0; co < Cout; ++co) {
    const float* gr = gy + (long)co * P;
    float s = 0.f;
    for (int p = tid; p < P; p += 256) s += gr[p] * ck[p];
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) gw[(long)co * CK + k] += red[0];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void bgrad_kernel(const float* __restrict__ gy, float* __restrict__ gb, int P) {
  __shared__ float red[256];
  const int co = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int p = tid; p < P; p += 256) s += gy[(long)co * P + p];
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) gb[co] += red[0];
}

// ---------------------------------------------------------------------------------------------------
// fp16 channel-innermost backward of the fused DCN (training path).  Column channel order k = g * 72 + t * 8 + j
// (group-major: the 9 taps x 8 channels one (pixel, group) thread reads are 144 contiguous bytes).
//   dcn_columns_kernel : col[p][k] = mask(g,t,p) * bilinear(x[.., g*8+j], p + tap t + offset(g,t,p))      (for dW)
//   dcn_col2im_kernel  : from dcol[p][k] = sum_co W[co][k] dY[p][co]:  d offset, d mask (raw, through the sigmoid) into
//                        dom, and the bilinear scatter of dcol * mask into dx32 (float atomics: summation order is not
//                        reproducible, like the reference's col2im, dcn_v2_im2col_cuda.cu:197-254)
// One thread per (pixel, group, tap), 8 channels = one 16-byte access per bilinear corner.
struct DcnBw {
  FMap x, om, col;           // col doubles as dcol
  FMap dom;
  float* dx32;
  float* wbuf;               // [N*G][tiles][32*32*8] scatter windows
  int G;
  // deterministic mode (tdvc_dcn_col2im_det): samples displaced out of the window are RECORDED instead of added with float
  // atomics: key = (target (n, y, x, g) << 27) | (source pixel * 36 + tap * 4 + corner), 8 values; the caller sorts the keys and
  // tdvc_dcn_far_apply adds each target's records in key order
  long long* far_key;
  float* far_val;
  int* far_count;
  int far_cap;
};

__device__ __forceinline__ void dcn_geom(const DcnBw& p, long i, int& n, long& pix, int& g, int& t, float& h_im, float& w_im, float& mask) {
  const long npix = (long)p.x.H * p.x.W;
  t = (int)(i % 9);
  long q = i / 9;
  g = (int)(q % p.G);
  q /= p.G;
  pix = q % npix;
  n = (int)(q / npix);
  const int y = (int)(pix / p.x.W), xx = (int)(pix % p.x.W);
  const half_t* omp = reinterpret_cast<const half_t*>(p.om.p) + (long)n * p.om.sn + pix * p.om.sp;
  const float oh = (float)omp[g * 18 + 2 * t], ow = (float)omp[g * 18 + 2 * t + 1];
  mask = 1.f / (1.f + __expf(-(float)omp[18 * p.G + g * 9 + t]));
  h_im = (float)(y - 1 + t / 3) + oh;
  w_im = (float)(xx - 1 + t % 3) + ow;
}

__global__ void dcn_columns_kernel(const DcnBw p, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int n, g, t; long pix; float h_im, w_im, mask;
  dcn_geom(p, i, n, pix, g, t, h_im, w_im, mask);
  const int H = p.x.H, W = p.x.W;
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (half_t)0.f;
  if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
    const int hl = (int)floorf(h_im), wl = (int)floorf(w_im), hh = hl + 1, wh = wl + 1;
    const float lh = h_im - hl, lw = w_im - wl;
    const half_t* xg = reinterpret_cast<const half_t*>(p.x.p) + (long)n * p.x.sn + g * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto corner = [&](int yy, int xc, float wt) {
      if (yy < 0 || yy > H - 1 || xc < 0 || xc > W - 1) return;
      const half8 v = *reinterpret_cast<const half8*>(xg + ((long)yy * W + xc) * p.x.sp);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += wt * (float)v[j];
    };
    corner(hl, wl, (1.f - lh) * (1.f - lw));
    corner(hl, wh, (1.f - lh) * lw);
    corner(hh, wl, lh * (1.f - lw));
    corner(hh, wh, lh * lw);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(acc[j] * mask);
  }
  *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(p.col.p) + (long)n * p.col.sn + pix * p.col.sp + g * 72 + t * 8) = o;
}

// One single-wave workgroup = an 8x8 pixel tile of one (image, deformable group).  The bilinear scatter goes to an LDS
// window of 24x24 pixels x 8 fp32 channels around the tile; samples displaced by more than 8 pixels fall back to global
// float atomics.  LDS float atomics run at ~0.5 lane-adds per clock per CU on this part (604 M adds = 2.4 of the first
// version's 3.1 ms), so they are the exception here: per (tap, corner) every lane stamps a tag at its target pixel and
// reads it back; the lane whose stamp survived owns the pixel for this instruction and updates its 8 channels with
// plain 128-bit LDS reads / writes, the (rare: the offset field is locally smooth) lanes that lost the stamp follow
// with ds_add_f32.  One wave per workgroup, so LDS operations retire in program order and nobody else touches the
// window.  The window is stored whole to the workgroup's slot of `wbuf` and dcn_window_gather_kernel adds the nine
// windows that cover each pixel into dx32 in a fixed order.
constexpr int C2I_T = 8, C2I_R = 8, C2I_W = C2I_T + 2 * C2I_R;
__global__ __launch_bounds__(64) void dcn_col2im_kernel(const DcnBw p) {
  __shared__ __attribute__((aligned(16))) float win[C2I_W * C2I_W * 8];
  __shared__ half8 xw[C2I_W * C2I_W];                                // this group's 8 channels of x over the window (0 outside the image)
  __shared__ int tag[C2I_W * C2I_W];
  const int H = p.x.H, W = p.x.W;
  const int tiles_x = (W + C2I_T - 1) / C2I_T;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int g = blockIdx.y % p.G, n = blockIdx.y / p.G;
  const int lane = threadIdx.x;
  const int wy0 = ty * C2I_T - C2I_R, wx0 = tx * C2I_T - C2I_R;
  const half_t* xg = reinterpret_cast<const half_t*>(p.x.p) + (long)n * p.x.sn + g * 8;
  float* dxn = p.dx32 + ((long)n * H * W) * (8 * p.G) + g * 8;
  for (int i = lane; i < C2I_W * C2I_W; i += 64) {
    const int yy = wy0 + i / C2I_W, xc = wx0 + i % C2I_W;
    half8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (half_t)0.f;
    if (yy >= 0 && yy < H && xc >= 0 && xc < W) v = *reinterpret_cast<const half8*>(xg + ((long)yy * W + xc) * p.x.sp);
    xw[i] = v;
    tag[i] = -1;
  }
  for (int i = lane; i < C2I_W * C2I_W * 2; i += 64) reinterpret_cast<float4*>(win)[i] = float4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  volatile __attribute__((address_space(3))) int* vtag = (volatile __attribute__((address_space(3))) int*)tag;      // ds_ ops, never forwarded
  const int y = ty * C2I_T + (lane >> 3), xx = tx * C2I_T + (lane & 7);
  const bool valid = y < H && xx < W;
  const long pix = valid ? (long)y * W + xx : 0;
  const half_t* omp = reinterpret_cast<const half_t*>(p.om.p) + (long)n * p.om.sn + pix * p.om.sp;
  half_t* dop = reinterpret_cast<half_t*>(p.dom.p) + (long)n * p.dom.sn + pix * p.dom.sp;
  const half_t* dcp = reinterpret_cast<const half_t*>(p.col.p) + (long)n * p.col.sn + pix * p.col.sp + g * 72;
  half2v off[9];
  float mk[9];
  half8 dc8[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    off[t] = *reinterpret_cast<const half2v*>(omp + g * 18 + 2 * t);
    mk[t] = __builtin_amdgcn_rcpf(1.f + __expf(-(float)omp[18 * p.G + g * 9 + t]));
    dc8[t] = *reinterpret_cast<const half8*>(dcp + t * 8);
  }
  float dof[18], dmk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float mask = mk[t];
    const float h_im = (float)(y - 1 + t / 3) + (float)off[t][0], w_im = (float)(xx - 1 + t % 3) + (float)off[t][1];
    const bool inside = valid && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
    float dc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dc[j] = (float)dc8[t][j];
    const int hl = (int)floorf(h_im), wl = (int)floorf(w_im), hh = hl + 1, wh = wl + 1;
    const float lh = h_im - hl, lw = w_im - wl;
    float dval = 0.f, dh = 0.f, dw = 0.f;                     // sum_c dcol_c * {val_c, dval_c/dh, dval_c/dw}
    auto corner = [&](int ci, int yy, int xc, float wt, float wth, float wtw) {
      const bool active = inside && yy >= 0 && yy <= H - 1 && xc >= 0 && xc <= W - 1;
      const int ly = yy - wy0, lx = xc - wx0;
      const bool inwin = active && ly >= 0 && ly < C2I_W && lx >= 0 && lx < C2I_W;
      const int tgt = inwin ? ly * C2I_W + lx : 0;
      const long off_px = active ? ((long)yy * W + xc) : 0;
      half8 v = xw[tgt];
      if (active && !inwin) v = *reinterpret_cast<const half8*>(xg + off_px * p.x.sp);
      float a[8], sx = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sx += dc[j] * (float)v[j];
        a[j] = dc[j] * mask * wt;
      }
      if (active) { dval += wt * sx; dh += wth * sx; dw += wtw * sx; }
      const int uid = ((t * 4 + ci) << 6) | lane;
      if (inwin) vtag[tgt] = uid;
      const bool owner = inwin && vtag[tgt] == uid;
      if (owner) {
        float4* w4 = reinterpret_cast<float4*>(win + tgt * 8);
        float4 lo = w4[0], hi = w4[1];
        lo.x += a[0]; lo.y += a[1]; lo.z += a[2]; lo.w += a[3];
        hi.x += a[4]; hi.y += a[5]; hi.z += a[6]; hi.w += a[7];
        w4[0] = lo; w4[1] = hi;
      }
      asm volatile("" ::: "memory");                         // the owners' stores are issued before the others' atomics
      if (inwin && !owner) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(win + tgt * 8 + j, a[j]);
      }
      asm volatile("" ::: "memory");
      if (active && !inwin) {
        if (p.far_key) {
          const int slot = atomicAdd(p.far_count, 1);          // slot order is arbitrary: the records are sorted by key afterwards
          if (slot < p.far_cap) {
            const long long tgt_id = (((long long)n * H * W + off_px) * p.G + g);
            p.far_key[slot] = (tgt_id << 27) | (long long)(pix * 36 + t * 4 + ci);
            float4* fv = reinterpret_cast<float4*>(p.far_val + (long)slot * 8);
            fv[0] = float4{a[0], a[1], a[2], a[3]};
            fv[1] = float4{a[4], a[5], a[6], a[7]};
          }
        } else {
          float* dg = dxn + off_px * (8 * p.G);
#pragma unroll
          for (int j = 0; j < 8; ++j) unsafeAtomicAdd(dg + j, a[j]);
        }
      }
    };
    corner(0, hl, wl, (1.f - lh) * (1.f - lw), -(1.f - lw), -(1.f - lh));
    corner(1, hl, wh, (1.f - lh) * lw, -lw, (1.f - lh));
    corner(2, hh, wl, lh * (1.f - lw), (1.f - lw), -lh);
    corner(3, hh, wh, lh * lw, lw, lh);
    dof[2 * t] = dh * mask; dof[2 * t + 1] = dw * mask; dmk[t] = dval * mask * (1.f - mask);
  }
  if (valid) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      half2v* d2 = reinterpret_cast<half2v*>(dop + g * 18 + 2 * t);
      const half2v o = *d2;
      *d2 = half2v{(half_t)((float)o[0] + dof[2 * t]), (half_t)((float)o[1] + dof[2 * t + 1])};
      half_t* dm = dop + 18 * p.G + g * 9 + t;
      *dm = (half_t)((float)*dm + dmk[t]);
    }
  }
  __syncthreads();
  float4* wb = reinterpret_cast<float4*>(p.wbuf + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (C2I_W * C2I_W * 8));
  for (int i = lane; i < C2I_W * C2I_W * 2; i += 64) wb[i] = reinterpret_cast<const float4*>(win)[i];
}

// dx32[n][y][x][g*8 ..] += the (up to) nine windows that cover pixel (y, x): the tiles (ty - 1 .. ty + 1, tx - 1 .. tx + 1).
// One thread per (n, g, y, x); fixed order, no atomics.
__global__ void dcn_window_gather_kernel(const DcnBw p, int tiles_x, int tiles_y) {
  const int H = p.x.H, W = p.x.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)p.x.N * p.G * H * W) return;
  const int g = (int)(i % p.G);
  long q = i / p.G;
  const int xx = (int)(q % W);
  q /= W;
  const int y = (int)(q % H), n = (int)(q / H);
  const int ty = y / C2I_T, tx = xx / C2I_T;
  float4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int wy = ty + dy, wx = tx + dx;
      if (wy < 0 || wy >= tiles_y || wx < 0 || wx >= tiles_x) continue;
      const int ly = y - (wy * C2I_T - C2I_R), lx = xx - (wx * C2I_T - C2I_R);
      const float4* w = reinterpret_cast<const float4*>(p.wbuf + ((long)(n * p.G + g) * (tiles_x * tiles_y) + wy * tiles_x + wx) * (C2I_W * C2I_W * 8) +
                                                        (ly * C2I_W + lx) * 8);
      const float4 u = w[0], v = w[1];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
  float4* d = reinterpret_cast<float4*>(p.dx32 + (((long)n * H + y) * W + xx) * (8 * p.G) + g * 8);
  float4 u = d[0], v = d[1];
  u.x += a.x; u.y += a.y; u.z += a.z; u.w += a.w;
  v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  d[0] = u; d[1] = v;
}

// deterministic mode: record `order[i]` is the i-th in key order; the thread at the first record of a target adds that
// target's records in key order (unique keys: a fixed order), then into dx32 (no other thread owns this target)
__global__ void dcn_far_apply_kernel(const long long* keys_sorted, const long long* order, const float* vals, int count, float* dx32) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const long long tgt = keys_sorted[i] >> 27;
  if (i > 0 && (keys_sorted[i - 1] >> 27) == tgt) return;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = i; j < count && (keys_sorted[j] >> 27) == tgt; ++j) {
    const float* v = vals + order[j] * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c] += v[c];
  }
  float* d = dx32 + tgt * 8;
#pragma unroll
  for (int c = 0; c < 8; ++c) d[c] += s[c];
}

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

extern "C" int tdvc_dcn_v2_backward_f32(const float* input, const float* weight, const float* bias,
                                        const float* offset, const float* mask, const float* grad_output,
                                        float* grad_input, float* grad_offset, float* grad_mask,
                                        float* grad_weight, float* grad_bias, float* columns,
                                        int B, int C, int H, int W, int Cout,
                                        int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                        int deformable_group, void* stream) {
  TDVC_CHECK(input && weight && bias && offset && mask && grad_output && grad_input && grad_offset && grad_mask && grad_weight &&
                 grad_bias && columns, "dcn_v2_backward: null tensor");
  TDVC_CHECK(B > 0 && C > 0 && H > 0 && W > 0 && Cout > 0 && kh >= 1 && kw >= 1 && kh * kw <= 49, "dcn_v2_backward: bad geometry");
  TDVC_CHECK(sh >= 1 && sw >= 1 && dh >= 1 && dw >= 1 && ph >= 0 && pw >= 0, "dcn_v2_backward: bad stride/dilation/pad");
  TDVC_CHECK(deformable_group >= 1 && C % deformable_group == 0, "dcn_v2_backward: channels %d not divisible by deformable_group %d", C, deformable_group);
  Geo g;
  g.C = C; g.H = H; g.W = W; g.kh = kh; g.kw = kw; g.ph = ph; g.pw = pw; g.sh = sh; g.sw = sw; g.dh = dh; g.dw = dw;
  g.G = deformable_group; g.Cout = Cout;
  g.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  g.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  TDVC_CHECK(g.Ho > 0 && g.Wo > 0, "dcn_v2_backward: empty output");
  const int K = kh * kw, P = g.Ho * g.Wo, CK = C * K;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipError_t e;
  if ((e = hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)B * C * H * W, st)) != hipSuccess ||
      (e = hipMemsetAsync(grad_weight, 0, sizeof(float) * (size_t)Cout * CK, st)) != hipSuccess ||
      (e = hipMemsetAsync(grad_bias, 0, sizeof(float) * (size_t)Cout, st)) != hipSuccess) {
    tdvc_set_error("dcn_v2_backward: memset failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  for (int n = 0; n < B; ++n) {
    const float* xn = input + (long)n * C * H * W;
    const float* on = offset + (long)n * g.G * 2 * K * P;
    const float* mn = mask + (long)n * g.G * K * P;
    const float* gn = grad_output + (long)n * Cout * P;
    hipLaunchKernelGGL(dcol_kernel, g1((long)CK * P), dim3(256), 0, st, weight, gn, columns, CK, P, Cout);
    hipLaunchKernelGGL(coord_kernel, g1((long)g.G * K * P), dim3(256), 0, st, columns, xn, on, mn, g,
                       grad_offset + (long)n * g.G * 2 * K * P, grad_mask + (long)n * g.G * K * P);
    hipLaunchKernelGGL(col2im_kernel, g1((long)CK * P), dim3(256), 0, st, columns, on, mn, g, grad_input + (long)n * C * H * W);
    hipLaunchKernelGGL(im2col_kernel, g1((long)CK * P), dim3(256), 0, st, xn, on, mn, g, columns);
    hipLaunchKernelGGL(wgrad_kernel, dim3(CK), dim3(256), 0, st, gn, columns, grad_weight, CK, P, Cout);
    hipLaunchKernelGGL(bgrad_kernel, dim3(Cout), dim3(256), 0, st, gn, grad_bias, P);
  }
  return tdvc_launch_status("tdvc_dcn_v2_backward_f32");
}

static inline bool dcn_bw_ok(const tdvc_fmap& x, const tdvc_fmap& om, const tdvc_fmap& col, int G) {
  return fmap_ok16(x) && fmap_ok16(om) && fmap_ok16(col) && G >= 1 && x.C == 8 * G && om.C >= 27 * G && col.C == 72 * G && x.N == om.N && x.N == col.N &&
         x.H == om.H && x.W == om.W && x.H == col.H && x.W == col.W && (long)x.H * x.W < 2147483647L;
}

extern "C" int tdvc_dcn_columns(const tdvc_fmap* x, const tdvc_fmap* om, int groups, const tdvc_fmap* col, void* stream) {
  TDVC_CHECK(x && om && col && dcn_bw_ok(*x, *om, *col, groups), "tdvc_dcn_columns: bad arguments (x C = 8*groups, om C >= 27*groups, col C = 72*groups)");
  DcnBw p;
  p.wbuf = nullptr;
  p.x = to_dev(*x); p.om = to_dev(*om); p.col = to_dev(*col); p.dom = null_fmap(); p.dx32 = nullptr; p.G = groups;
  p.far_key = nullptr; p.far_val = nullptr; p.far_count = nullptr; p.far_cap = 0;
  const long total = (long)x->N * x->H * x->W * groups * 9;
  hipLaunchKernelGGL(dcn_columns_kernel, g1(total), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, total);
  return tdvc_launch_status("tdvc_dcn_columns");
}

extern "C" int64_t tdvc_dcn_col2im_work_floats(int N, int H, int W, int groups) {
  if (N <= 0 || H <= 0 || W <= 0 || groups <= 0) return TDVC_EINVAL;
  return (int64_t)N * groups * ((W + C2I_T - 1) / C2I_T) * ((H + C2I_T - 1) / C2I_T) * (C2I_W * C2I_W * 8);
}

static int dcn_col2im_impl(const tdvc_fmap* x, const tdvc_fmap* om, const tdvc_fmap* dcol, int groups, float* dx32, const tdvc_fmap* dom,
                           float* work, int64_t work_floats, long long* far_key, float* far_val, int* far_count, int far_cap, void* stream);

extern "C" int tdvc_dcn_col2im(const tdvc_fmap* x, const tdvc_fmap* om, const tdvc_fmap* dcol, int groups, float* dx32, const tdvc_fmap* dom,
                               float* work, int64_t work_floats, void* stream) {
  return dcn_col2im_impl(x, om, dcol, groups, dx32, dom, work, work_floats, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int tdvc_dcn_col2im_det(const tdvc_fmap* x, const tdvc_fmap* om, const tdvc_fmap* dcol, int groups, float* dx32, const tdvc_fmap* dom,
                                   float* work, int64_t work_floats, int64_t* far_keys, float* far_vals, int32_t* far_count, int32_t far_cap,
                                   void* stream) {
  TDVC_CHECK(far_keys && far_vals && far_count && far_cap >= 1 && (reinterpret_cast<uintptr_t>(far_vals) & 15) == 0, "tdvc_dcn_col2im_det: record buffers missing or unaligned");
  TDVC_CHECK(x && (long)x->H * x->W * 36 < (1L << 27) && (long)x->N * x->H * x->W * groups < (1L << 36), "tdvc_dcn_col2im_det: map too large for the 27 + 36 bit record key");
  return dcn_col2im_impl(x, om, dcol, groups, dx32, dom, work, work_floats, reinterpret_cast<long long*>(far_keys), far_vals, far_count, far_cap, stream);
}

extern "C" int tdvc_dcn_far_apply(const int64_t* keys_sorted, const int64_t* order, const float* far_vals, int32_t count, float* dx32, void* stream) {
  TDVC_CHECK(keys_sorted && order && far_vals && dx32 && count >= 0, "tdvc_dcn_far_apply: bad arguments");
  if (count == 0) return TDVC_OK;
  hipLaunchKernelGGL(dcn_far_apply_kernel, g1(count), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const long long*>(keys_sorted),
                     reinterpret_cast<const long long*>(order), far_vals, count, dx32);
  return tdvc_launch_status("tdvc_dcn_far_apply");
}

static int dcn_col2im_impl(const tdvc_fmap* x, const tdvc_fmap* om, const tdvc_fmap* dcol, int groups, float* dx32, const tdvc_fmap* dom,
                           float* work, int64_t work_floats, long long* far_key, float* far_val, int* far_count, int far_cap, void* stream) {
  TDVC_CHECK(x && om && dcol && dx32 && dom && work && dcn_bw_ok(*x, *om, *dcol, groups) && fmap_ok16(*dom) && dom->C >= 27 * groups && dom->N == x->N &&
                 dom->H == x->H && dom->W == x->W, "tdvc_dcn_col2im: bad arguments");
  TDVC_CHECK(work_floats >= tdvc_dcn_col2im_work_floats(x->N, x->H, x->W, groups) && (reinterpret_cast<uintptr_t>(work) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(dx32) & 15) == 0, "tdvc_dcn_col2im: workspace too small or unaligned");
  DcnBw p;
  p.x = to_dev(*x); p.om = to_dev(*om); p.col = to_dev(*dcol); p.dom = to_dev(*dom); p.dx32 = dx32; p.wbuf = work; p.G = groups;
  p.far_key = far_key; p.far_val = far_val; p.far_count = far_count; p.far_cap = far_cap;
  const int tiles_x = (x->W + C2I_T - 1) / C2I_T, tiles_y = (x->H + C2I_T - 1) / C2I_T;
  const dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)(x->N * groups));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(dcn_col2im_kernel, grid, dim3(64), 0, st, p);
  hipLaunchKernelGGL(dcn_window_gather_kernel, g1((long)x->N * groups * x->H * x->W), dim3(256), 0, st, p, tiles_x, tiles_y);
  return tdvc_launch_status("tdvc_dcn_col2im");
}
