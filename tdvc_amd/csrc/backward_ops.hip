// Streaming kernels of the backward pass (training path): the adjoints of the fused conv epilogue pieces and of
// the HBM-bound forward operators.  Same conventions as pointwise.hip: channel-innermost fmaps, 16-byte accesses,
// one pass, order-fixed two-stage reductions.
#include "pointwise_common.h"

namespace {

// out = g * act'(z) with the sign of the pre-activation z recovered from the stored output: z > 0 <=> y - res > 0
// (ReLU / LeakyReLU with slope >= 0).  `out` may alias `g`.
__global__ void act_backward_kernel(FMap g, FMap y, FMap res, float slope, FMap out) {
  const long npix = (long)g.H * g.W;
  const int chunks = g.C / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * g.N * chunks) return;
  const int c = (int)(i % chunks) * 8;
  const long q = i / chunks;
  const int n = (int)(q / npix);
  const long pix = q % npix;
  float gv[8], yv[8];
  load8(g, n, pix, c, gv);
  load8(y, n, pix, c, yv);
  if (res.p) {
    float rv[8];
    load8(res, n, pix, c, rv);
#pragma unroll
    for (int j = 0; j < 8; ++j) yv[j] -= rv[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) gv[j] = yv[j] > 0.f ? gv[j] : gv[j] * slope;
  store8(out, n, pix, c, gv);
}

// out[n][Y][X][(i*2+j)*C + c] = y[n][2Y+i][2X+j][c]: the sub-pixel conv's gradient in packed-row order
__global__ void pixel_unshuffle_kernel(FMap y, FMap out) {
  const long npix = (long)out.H * out.W;
  const int chunks = y.C / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * out.N * chunks * 4) return;
  const int c = (int)(i % chunks) * 8;
  long q = i / chunks;
  const int sub = (int)(q & 3);
  q >>= 2;
  const int n = (int)(q / npix);
  const long pix = q % npix;
  const int Y = (int)(pix / out.W), X = (int)(pix % out.W);
  float v[8];
  load8(y, n, (long)(2 * Y + (sub >> 1)) * y.W + 2 * X + (sub & 1), c, v);
  store8(out, n, pix, sub * y.C + c, v);
}

// partial[(n*nblocks + blk)][c] = sum over the block's pixel range, any channel count (256 channels per blockIdx.z)
__global__ __launch_bounds__(256) void channel_sum_wide_kernel(FMap x, float* partial, int nblocks) {
  __shared__ float red[256][9];
  const int c0 = blockIdx.z * 256;
  const int cw = min(256, x.C - c0);
  const int chunks = cw / 8;
  const int lanes = 256 / chunks;
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
    for (long pix = p0 + pl; pix < p1; pix += lanes) {
      float v[8];
      load8(x, n, pix, c0 + ck * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tid][j] = (pl < lanes) ? acc[j] : 0.f;
  __syncthreads();
  if (tid < cw) {
    const int cc = tid / 8, j = tid % 8;
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[l * chunks + cc][j];
    partial[((long)n * nblocks + blockIdx.x) * x.C + c0 + tid] = s;
  }
}

// out[dst[c] or c] += scale * sum_rows partial[row][c]: 64 channels x 4 row lanes per workgroup, fixed order
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* partial, int rows, int C_, int nvalid, const int* dst, float scale, float* out) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < nvalid)
    for (int r = rl; r < rows; r += 4) s += partial[(long)r * C_ + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < nvalid) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    const int d = dst ? dst[c] : c;
    if (d >= 0) out[d] += t * scale;
  }
}

// dst[..., c] = c < src.C ? src[..., c] : 0 (any dtypes; dst.C % 8 == 0 when fp16)
__global__ void copy_cast_kernel(FMap src, FMap dst) {
  const long npix = (long)dst.H * dst.W;
  const int chunks = (dst.C + 7) / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * dst.N * chunks) return;
  const int c = (int)(i % chunks) * 8;
  const long q = i / chunks;
  const int n = (int)(q / npix);
  const long pix = q % npix;
  float v[8];
  if (src.f32) {
    const float* sp = reinterpret_cast<const float*>(src.p) + (long)n * src.sn + pix * src.sp;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (c + j < src.C) ? sp[c + j] : 0.f;
  } else if (c < src.C) {
    load8(src, n, pix, c, v);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  }
  store8(dst, n, pix, c, v);
}

// g * [0 < y < 1] (the reconstruction clamp), in place
__global__ void clamp01_backward_kernel(FMap g, FMap y) {
  const long npix = (long)g.H * g.W;
  const int chunks = g.C / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * g.N * chunks) return;
  const int c = (int)(i % chunks) * 8;
  const long q = i / chunks;
  const int n = (int)(q / npix);
  const long pix = q % npix;
  float gv[8], yv[8];
  load8(g, n, pix, c, gv);
  load8(y, n, pix, c, yv);
#pragma unroll
  for (int j = 0; j < 8; ++j) gv[j] = (yv[j] > 0.f && yv[j] < 1.f) ? gv[j] : 0.f;
  store8(g, n, pix, c, gv);
}

// y = a * gate[n][c]:  da += g * gate;  partial[(n*nblocks + blk)][c] = sum_pix g * a
__global__ __launch_bounds__(256) void gate_backward_kernel(FMap g, FMap a, const float* gate, FMap da, float* partial, int nblocks) {
  __shared__ float red[256][9];
  const int chunks = g.C / 8;            // <= 32
  const int lanes = 256 / chunks;
  const int tid = threadIdx.x;
  const int ck = tid % chunks, pl = tid / chunks;
  const int n = blockIdx.y;
  const long npix = (long)g.H * g.W;
  const long per = (npix + nblocks - 1) / nblocks;
  const long p0 = (long)blockIdx.x * per;
  const long p1 = p0 + per < npix ? p0 + per : npix;
  float acc[8], gt[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[j] = 0.f; gt[j] = (pl < lanes) ? gate[(long)n * g.C + ck * 8 + j] : 0.f; }
  if (pl < lanes) {
    for (long pix = p0 + pl; pix < p1; pix += lanes) {
      float gv[8], av[8];
      load8(g, n, pix, ck * 8, gv);
      load8(a, n, pix, ck * 8, av);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += gv[j] * av[j];
      if (da.p) {
        float dv[8];
        load8(da, n, pix, ck * 8, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[j] += gv[j] * gt[j];
        store8(da, n, pix, ck * 8, dv);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tid][j] = (pl < lanes) ? acc[j] : 0.f;
  __syncthreads();
  if (tid < g.C) {
    const int cc = tid / 8, j = tid % 8;
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[l * chunks + cc][j];
    partial[((long)n * nblocks + blockIdx.x) * g.C + tid] = s;
  }
}

// out[n][c] += sum_blk partial[(n*nblocks + blk)][c]
__global__ void reduce_groups_kernel(const float* partial, int N, int nblocks, int C_, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C_) return;
  const int n = i / C_, c = i - n * C_;
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += partial[((long)n * nblocks + b) * C_ + c];
  out[i] += s;
}

// Backward of se_gate_kernel (gate = sigmoid(W2 relu(W1 mean + b1) + b2)): one workgroup; W1 / W2 are staged in LDS, the
// batch items are walked in order and the parameter gradients leave the registers once.
// dmean[n][c] (written), dW1 / db1 / dW2 / db2 += scale * ...
__global__ __launch_bounds__(1024) void se_gate_backward_kernel(const float* partial, int nblocks, float inv_count, int N, int C_, int Cmid,
                                                                const float* w1, const float* b1, const float* w2, const float* b2,
                                                                const float* gate, const float* dgate, float scale, float* dmean,
                                                                float* dw1, float* db1, float* dw2, float* db2) {
  extern __shared__ float sm[];
  float* w1s = sm;                       // [Cmid][C_]
  float* w2s = w1s + Cmid * C_;          // [C_][Cmid + 1]
  float* mean = w2s + C_ * (Cmid + 1);   // [N][C_]
  float* ds = mean + N * C_;             // [N][C_]
  float* mid = ds + N * C_;              // [32]
  float* dmid = mid + 32;                // [32]
  const int tid = threadIdx.x;
  for (int i = tid; i < Cmid * C_; i += 1024) {
    w1s[i] = w1[i];
    w2s[(i / Cmid) * (Cmid + 1) + (i % Cmid)] = w2[i];
  }
  for (int i = tid; i < N * C_; i += 1024) {
    const int n = i / C_, c = i - n * C_;
    float s = 0.f;
#pragma unroll 8
    for (int b = 0; b < nblocks; ++b) s += partial[((long)n * nblocks + b) * C_ + c];
    mean[i] = s * inv_count;
    const float gt = gate[i];
    ds[i] = dgate[i] * gt * (1.f - gt);                                  // d pre-sigmoid
  }
  float a1[32], a2[32], ab1 = 0.f, ab2 = 0.f;                            // dW1[j][tid], dW2[tid][j], db1[tid], db2[tid]
#pragma unroll
  for (int j = 0; j < 32; ++j) { a1[j] = 0.f; a2[j] = 0.f; }
  __syncthreads();
  for (int n = 0; n < N; ++n) {
    const float* mn = mean + n * C_;
    const float* dn = ds + n * C_;
    if (tid < Cmid) {
      float s = b1[tid], d = 0.f;
      for (int c = 0; c < C_; ++c) { s += w1s[tid * C_ + c] * mn[c]; d += w2s[c * (Cmid + 1) + tid] * dn[c]; }
      mid[tid] = s > 0.f ? s : 0.f;
      dmid[tid] = s > 0.f ? d : 0.f;
      ab1 += s > 0.f ? d : 0.f;
    }
    __syncthreads();
    if (tid < C_) {
      ab2 += dn[tid];
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j)
        if (j < Cmid) {
          a2[j] += dn[tid] * mid[j];
          a1[j] += dmid[j] * mn[tid];
          d += w1s[j * C_ + tid] * dmid[j];
        }
      dmean[(long)n * C_ + tid] = d;
    }
    __syncthreads();
  }
  if (tid < Cmid) db1[tid] += scale * ab1;
  if (tid < C_) {
    db2[tid] += scale * ab2;
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (j < Cmid) {
        dw2[tid * Cmid + j] += scale * a2[j];
        dw1[j * C_ + tid] += scale * a1[j];
      }
  }
}

// dx[n][pix][c] += v[n][c] * scale
__global__ void bcast_channel_add_kernel(FMap dx, const float* v, float scale) {
  const long npix = (long)dx.H * dx.W;
  const int chunks = dx.C / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * dx.N * chunks) return;
  const int c = (int)(i % chunks) * 8;
  const long q = i / chunks;
  const int n = (int)(q / npix);
  const long pix = q % npix;
  float d[8];
  load8(dx, n, pix, c, d);
#pragma unroll
  for (int j = 0; j < 8; ++j) d[j] += v[(long)n * dx.C + c + j] * scale;
  store8(dx, n, pix, c, d);
}

// off += repeat(flow, C/2): dflow[.., 0/1] += sum over even / odd channels of doff
__global__ void add_flow_backward_kernel(FMap doff, FMap dflow) {
  const long npix = (long)doff.H * doff.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * doff.N) return;
  const int n = (int)(i / npix);
  const long pix = i % npix;
  float sx = 0.f, sy = 0.f;
  for (int c = 0; c < doff.C; c += 8) {
    float v[8];
    load8(doff, n, pix, c, v);
    sx += v[0] + v[2] + v[4] + v[6];
    sy += v[1] + v[3] + v[5] + v[7];
  }
  float* fp = reinterpret_cast<float*>(dflow.p) + (long)n * dflow.sn + pix * dflow.sp;
  fp[0] += sx;
  fp[1] += sy;
}

// x = lrelu(x_pre + b[c % b.C]) in place:  dx <- dx * lrelu'(x) (in place),  db[c'] += sum over the T slices
__global__ void bcast_add_act_backward_kernel(FMap dx, FMap x, FMap db, float slope) {
  const long npix = (long)dx.H * dx.W;
  const int chunks = db.C / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * dx.N * chunks) return;
  const int c = (int)(i % chunks) * 8;
  const long q = i / chunks;
  const int n = (int)(q / npix);
  const long pix = q % npix;
  float acc[8];
  load8(db, n, pix, c, acc);
  for (int t = 0; t * db.C < dx.C; ++t) {
    float g[8], xv[8];
    load8(dx, n, pix, t * db.C + c, g);
    load8(x, n, pix, t * db.C + c, xv);
#pragma unroll
    for (int j = 0; j < 8; ++j) { g[j] = xv[j] > 0.f ? g[j] : g[j] * slope; acc[j] += g[j]; }
    store8(dx, n, pix, t * db.C + c, g);
  }
  store8(db, n, pix, c, acc);
}

// adjoint of upsample2x_kernel (bilinear x2, align_corners = False): dx += U^T dy, gathered per input pixel
__global__ void upsample2x_backward_kernel(FMap dy, FMap dx) {
  const int chunks = dx.C / 8;
  const long npix = (long)dx.H * dx.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * chunks * dx.N) return;
  const int c = (int)(i % chunks) * 8;
  const long t = i / chunks;
  const long pix = t % npix;
  const int n = (int)(t / npix);
  const int iy = (int)(pix / dx.W), ix = (int)(pix % dx.W);
  float acc[8];
  load8(dx, n, pix, c, acc);
  // output row oy samples source coordinate sy = (oy + 0.5)/2 - 0.5 clamped to >= 0, rows y0 = floor(sy), y1 = min(y0+1, H-1)
  for (int oy = 2 * iy - 2; oy <= 2 * iy + 2; ++oy) {
    if (oy < 0 || oy >= dy.H) continue;
    float sy = (oy + 0.5f) * 0.5f - 0.5f;
    if (sy < 0.f) sy = 0.f;
    const int y0 = (int)sy, y1 = min(y0 + 1, dx.H - 1);
    const float ly = sy - y0;
    const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
    if (wy == 0.f) continue;
    for (int ox = 2 * ix - 2; ox <= 2 * ix + 2; ++ox) {
      if (ox < 0 || ox >= dy.W) continue;
      float sx = (ox + 0.5f) * 0.5f - 0.5f;
      if (sx < 0.f) sx = 0.f;
      const int x0 = (int)sx, x1 = min(x0 + 1, dx.W - 1);
      const float lx = sx - x0;
      const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
      if (wx == 0.f) continue;
      float g[8];
      load8(dy, n, (long)oy * dy.W + ox, c, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += wy * wx * g[j];
    }
  }
  store8(dx, n, pix, c, acc);
}

// adjoint of resize_bilinear_kernel (pointwise.hip: bilinear, align_corners = False, arbitrary sizes, optional per-channel
// scale): dx[iy][ix][c] += sum over the output pixels whose 2 x 2 source footprint contains (iy, ix) of wy * wx * chscale[c] *
// dy[oy][ox][c], gathered per INPUT pixel (no atomics: reproducible).  Output row oy samples sy = rh (oy + 0.5) - 0.5 clamped
// to >= 0 with rows y0 = min(floor(sy), H - 1), y1 = y0 + (y0 < H - 1): the candidates of input row iy are the oy with
// sy in (iy - 1, iy + 1), plus the clamped borders.
__global__ void resize_bilinear_backward_kernel(FMap dy, FMap dx, const float* chscale) {
  const long npix = (long)dx.H * dx.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * dx.N) return;
  const int n = (int)(i / npix);
  const long pix = i % npix;
  const int iy = (int)(pix / dx.W), ix = (int)(pix % dx.W);
  const float rh = (float)dx.H / (float)dy.H, rw = (float)dx.W / (float)dy.W;
  // oy with rh (oy + 0.5) - 0.5 in [iy - 1, iy + 1]: a generous integer bracket, every candidate re-derives its exact weights
  const int oy_lo = max(0, (int)floorf(((float)iy - 0.5f) / rh - 0.5f) - 1), oy_hi = min(dy.H - 1, (int)ceilf(((float)iy + 1.5f) / rh - 0.5f) + 1);
  const int ox_lo = max(0, (int)floorf(((float)ix - 0.5f) / rw - 0.5f) - 1), ox_hi = min(dy.W - 1, (int)ceilf(((float)ix + 1.5f) / rw - 0.5f) + 1);
  const float* gp = reinterpret_cast<const float*>(dy.p) + (long)n * dy.sn;
  float* dp = reinterpret_cast<float*>(dx.p) + (long)n * dx.sn + pix * dx.sp;
  for (int c = 0; c < dx.C; ++c) {
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const float sy = fmaxf(rh * (oy + 0.5f) - 0.5f, 0.f);
      const int y0 = (int)sy < dx.H - 1 ? (int)sy : dx.H - 1, y1 = y0 + (y0 < dx.H - 1);
      const float ly1 = sy - y0;
      const float wy = (y0 == iy ? 1.f - ly1 : 0.f) + (y1 == iy ? ly1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const float sx = fmaxf(rw * (ox + 0.5f) - 0.5f, 0.f);
        const int x0 = (int)sx < dx.W - 1 ? (int)sx : dx.W - 1, x1 = x0 + (x0 < dx.W - 1);
        const float lx1 = sx - x0;
        const float wx = (x0 == ix ? 1.f - lx1 : 0.f) + (x1 == ix ? lx1 : 0.f);
        if (wx == 0.f) continue;
        acc += wy * wx * gp[((long)oy * dy.W + ox) * dy.sp + c];
      }
    }
    dp[c] += chscale ? acc * chscale[c] : acc;
  }
}

// Backward of spynet_level_input_kernel, stage 1 (per full-resolution pixel): total gradient of the up-sampled flow
//   dflow_up <- dflow_up (conv residual) + dcat8[6:8] (flow channels of the level input) + warp gradient
// The warp is grid_sample(bilinear, border, align_corners=True): d/d(ix) of the bilinear sample, zero where the
// sampling position was clamped to the border (torch's clip_coordinates_set_grad).
__global__ void spynet_warp_backward_kernel(FMap supp, FMap flow_up, FMap dcat8, FMap dflow_up) {
  const int H = supp.H, W = supp.W;
  const long npix = (long)H * W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * supp.N) return;
  const int n = (int)(i / npix);
  const long pix = i % npix;
  const int y = (int)(pix / W), x = (int)(pix % W);
  const float* fu = reinterpret_cast<const float*>(flow_up.p) + (long)n * flow_up.sn + pix * flow_up.sp;
  const float fx = fu[0], fy = fu[1];
  const float gx = (float)x + fx, gy = (float)y + fy;
  const float wm = (float)(W - 1 > 1 ? W - 1 : 1), hm = (float)(H - 1 > 1 ? H - 1 : 1);
  const float nx = 2.0f * gx / wm - 1.0f, ny = 2.0f * gy / hm - 1.0f;
  float ix = ((nx + 1.f) / 2.f) * (float)(W - 1), iy = ((ny + 1.f) / 2.f) * (float)(H - 1);
  const float mx = (ix >= 0.f && ix <= (float)(W - 1)) ? (float)(W - 1) / wm : 0.f;      // d ix / d fx (0 where clamped)
  const float my = (iy >= 0.f && iy <= (float)(H - 1)) ? (float)(H - 1) / hm : 0.f;
  ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
  iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
  const int ix0 = (int)floorf(ix), iy0 = (int)floorf(iy), ix1 = ix0 + 1, iy1 = iy0 + 1;
  const bool x0ok = ix0 >= 0 && ix0 < W, x1ok = ix1 >= 0 && ix1 < W, y0ok = iy0 >= 0 && iy0 < H, y1ok = iy1 >= 0 && iy1 < H;
  const float* sp = reinterpret_cast<const float*>(supp.p) + (long)n * supp.sn;
  float dc[8];
  load8(dcat8, n, pix, 0, dc);
  float gix = 0.f, giy = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float nw = (y0ok && x0ok) ? sp[((long)iy0 * W + ix0) * supp.sp + c] : 0.f;
    const float ne = (y0ok && x1ok) ? sp[((long)iy0 * W + ix1) * supp.sp + c] : 0.f;
    const float sw = (y1ok && x0ok) ? sp[((long)iy1 * W + ix0) * supp.sp + c] : 0.f;
    const float se = (y1ok && x1ok) ? sp[((long)iy1 * W + ix1) * supp.sp + c] : 0.f;
    const float g = dc[3 + c];
    gix += g * ((ne - nw) * ((float)iy1 - iy) + (se - sw) * (iy - (float)iy0));
    giy += g * ((sw - nw) * ((float)ix1 - ix) + (se - ne) * (ix - (float)ix0));
  }
  float* df = reinterpret_cast<float*>(dflow_up.p) + (long)n * dflow_up.sn + pix * dflow_up.sp;
  df[0] += dc[6] + mx * gix;
  df[1] += dc[7] + my * giy;
}

// stage 2 (per half-resolution pixel): dflow_lo += 2 * U^T dflow_up, U = bilinear x2 with align_corners = True
__global__ void upsample_ac_backward_kernel(FMap dhi, FMap dlo) {
  const int H = dhi.H, W = dhi.W, h2 = dlo.H, w2 = dlo.W;
  const long npix = (long)h2 * w2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * dlo.N) return;
  const int n = (int)(i / npix);
  const long pix = i % npix;
  const int Y = (int)(pix / w2), X = (int)(pix % w2);
  const float ry = (H > 1) ? (float)(h2 - 1) / (float)(H - 1) : 0.f;
  const float rx = (W > 1) ? (float)(w2 - 1) / (float)(W - 1) : 0.f;
  const int ylo = ry > 0.f ? max(0, (int)floorf((Y - 1) / ry)) : 0, yhi = ry > 0.f ? min(H - 1, (int)ceilf((Y + 1) / ry)) : H - 1;
  const int xlo = rx > 0.f ? max(0, (int)floorf((X - 1) / rx)) : 0, xhi = rx > 0.f ? min(W - 1, (int)ceilf((X + 1) / rx)) : W - 1;
  const float* hp = reinterpret_cast<const float*>(dhi.p) + (long)n * dhi.sn;
  float sx = 0.f, sy_ = 0.f;
  for (int y = ylo; y <= yhi; ++y) {
    const float sy = ry * y;
    const int y0 = (int)sy, y1 = y0 + (y0 < h2 - 1);
    const float ly1 = sy - y0;
    const float wy = (y0 == Y ? 1.f - ly1 : 0.f) + (y1 == Y ? ly1 : 0.f);
    if (wy == 0.f) continue;
    for (int x = xlo; x <= xhi; ++x) {
      const float sxx = rx * x;
      const int x0 = (int)sxx, x1 = x0 + (x0 < w2 - 1);
      const float lx1 = sxx - x0;
      const float wx = (x0 == X ? 1.f - lx1 : 0.f) + (x1 == X ? lx1 : 0.f);
      if (wx == 0.f) continue;
      const float* g = hp + ((long)y * W + x) * dhi.sp;
      sx += wy * wx * g[0];
      sy_ += wy * wx * g[1];
    }
  }
  float* lp = reinterpret_cast<float*>(dlo.p) + (long)n * dlo.sn + pix * dlo.sp;
  lp[0] += 2.f * sx;
  lp[1] += 2.f * sy_;
}

// out = sigmoid(x) / g *= s * (1 - s) on flat fp32 arrays (DCN modulation mask, dcn_v2_amp.py:226)
__global__ void sigmoid_f32_kernel(const float* x, float* out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = 1.f / (1.f + __expf(-x[i]));
}
__global__ void sigmoid_backward_f32_kernel(float* g, const float* s, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) g[i] *= s[i] * (1.f - s[i]);
}
// dst += scale * src on flat fp32 arrays (parameter-gradient accumulation)
__global__ void axpy_f32_kernel(float* dst, const float* src, float scale, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += scale * src[i];
}

// ---- backward of match_gather_kernel (FeatureFix, pnet.py:240-255): cat[p] = [a * cor, b * cor], a = fin[p],
// b = fref[src(p)], cor = a.b / max(|a||b|, 1e-8).  Eight lanes per pixel (8 channels each), as in the forward.
struct MatchGrad { float da[8], db[8]; };
__device__ __forceinline__ MatchGrad match_pair_grad(const float a[8], const float b[8], const float ga[8], const float gb[8], int lanes) {
  float w12 = 0.f, w1 = 0.f, w2 = 0.f, dc = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { w12 += a[j] * b[j]; w1 += a[j] * a[j]; w2 += b[j] * b[j]; dc += ga[j] * a[j] + gb[j] * b[j]; }
  for (int o = 1; o < lanes; o <<= 1) {
    w12 += __shfl_xor(w12, o);
    w1 += __shfl_xor(w1, o);
    w2 += __shfl_xor(w2, o);
    dc += __shfl_xor(dc, o);
  }
  const float den = sqrtf(w1) * sqrtf(w2);
  const bool live = den > 1e-8f;
  const float D = live ? den : 1e-8f;
  const float cor = w12 / D;
  const float ka = live ? cor / w1 : 0.f, kb = live ? cor / w2 : 0.f;      // d(den)/da = den * a / |a|^2
  MatchGrad r;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    r.da[j] = ga[j] * cor + dc * (b[j] / D - ka * a[j]);
    r.db[j] = gb[j] * cor + dc * (a[j] / D - kb * b[j]);
  }
  return r;
}

// dfin[p] += da(p)
__global__ __launch_bounds__(256) void match_gather_backward_in_kernel(FMap fin, FMap fref, const int32_t* idx, int ks, int nbh, int nbw, FMap dcat, FMap dfin) {
  const int lanes = fin.C / 8;
  const long npix = (long)fin.H * fin.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long t = i / lanes;
  const int c = (int)(i % lanes) * 8;
  const bool valid = t < npix * fin.N;
  const long tc = valid ? t : 0;
  const int n = (int)(tc / npix);
  const long pix = tc % npix;
  const int y = (int)(pix / fin.W), x = (int)(pix % fin.W);
  const int m = idx[(long)n * nbh * nbw + (y / ks + 1) * nbw + (x / ks + 1)];
  const int sy = (m / nbw - 1) * ks + y % ks, sx = (m % nbw - 1) * ks + x % ks;
  float a[8], b[8], ga[8], gb[8];
  load8(fin, n, pix, c, a);
  if (sy >= 0 && sy < fref.H && sx >= 0 && sx < fref.W) {
    load8(fref, n, (long)sy * fref.W + sx, c, b);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = 0.f;
  }
  load8(dcat, n, pix, c, ga);
  load8(dcat, n, pix, fin.C + c, gb);
  const MatchGrad r = match_pair_grad(a, b, ga, gb, lanes);
  if (!valid) return;
  float d[8];
  load8(dfin, n, pix, c, d);
#pragma unroll
  for (int j = 0; j < 8; ++j) d[j] += r.da[j];
  store8(dfin, n, pix, c, d);
}

// dfref[q] += sum over the output blocks matched to q's block of db(p(q))   (gather form: no atomics, fixed order)
__global__ __launch_bounds__(256) void match_gather_backward_ref_kernel(FMap fin, FMap fref, const int32_t* idx, int ks, int nbh, int nbw, FMap dcat, FMap dfref) {
  const int lanes = fin.C / 8;
  const long npix = (long)fref.H * fref.W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long t = i / lanes;
  const int c = (int)(i % lanes) * 8;
  const bool valid = t < npix * fref.N;
  const long tc = valid ? t : 0;
  const int n = (int)(tc / npix);
  const long pix = tc % npix;
  const int qy = (int)(pix / fref.W), qx = (int)(pix % fref.W);
  const int mq = (qy / ks + 1) * nbw + (qx / ks + 1);
  float b[8], acc[8];
  load8(fref, n, pix, c, b);
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int32_t* id = idx + (long)n * nbh * nbw;
  for (int blk = 0; blk < nbh * nbw; ++blk) {
    if (id[blk] != mq) continue;                           // uniform over the 8 lanes of a pixel
    const int y = (blk / nbw - 1) * ks + qy % ks, x = (blk % nbw - 1) * ks + qx % ks;
    if (y < 0 || y >= fin.H || x < 0 || x >= fin.W) continue;
    const long p = (long)y * fin.W + x;
    float a[8], ga[8], gb[8];
    load8(fin, n, p, c, a);
    load8(dcat, n, p, c, ga);
    load8(dcat, n, p, fin.C + c, gb);
    const MatchGrad r = match_pair_grad(a, b, ga, gb, lanes);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += r.db[j];
  }
  if (!valid) return;
  float d[8];
  load8(dfref, n, pix, c, d);
#pragma unroll
  for (int j = 0; j < 8; ++j) d[j] += acc[j];
  store8(dfref, n, pix, c, d);
}

__global__ void gdn_backward_kernel(FMap g, FMap x, FMap n32, int inverse, FMap dn, FMap dx) {
  const long npix = (long)g.H * g.W;
  const int chunks = g.C / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * g.N * chunks) return;
  const int c = (int)(i % chunks) * 8;
  const long q = i / chunks;
  const int n = (int)(q / npix);
  const long pix = q % npix;
  float gv[8], xv[8], nv[8], dv[8], dnv[8];
  load8(g, n, pix, c, gv);
  load8(x, n, pix, c, xv);
  load8(n32, n, pix, c, nv);
  load8(dx, n, pix, c, dv);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (inverse) {
      const float r = sqrtf(nv[j]);
      dv[j] += gv[j] * r;
      dnv[j] = 0.5f * gv[j] * xv[j] / r;
    } else {
      const float r = rsqrtf(nv[j]);
      dv[j] += gv[j] * r;
      dnv[j] = -0.5f * gv[j] * xv[j] * r / nv[j];
    }
  }
  store8(dx, n, pix, c, dv);
  store8(dn, n, pix, c, dnv);
}

__global__ void mul2_accumulate_kernel(FMap dx, FMap x, FMap t) {
  const long npix = (long)dx.H * dx.W;
  const int chunks = dx.C / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * dx.N * chunks) return;
  const int c = (int)(i % chunks) * 8;
  const long q = i / chunks;
  const int n = (int)(q / npix);
  const long pix = q % npix;
  float dv[8], xv[8], tv[8];
  load8(dx, n, pix, c, dv);
  load8(x, n, pix, c, xv);
  load8(t, n, pix, c, tv);
#pragma unroll
  for (int j = 0; j < 8; ++j) dv[j] += 2.f * xv[j] * tv[j];
  store8(dx, n, pix, c, dv);
}

// ---------------------------------------------------------------- entropy-model rate terms (training: additive noise)
constexpr int EBP = EB_NP;  // floats per channel of the packed factorised-prior parameters (pointwise_common.h::eb_logits)

// logits_cumulative(v) and its adjoint: gp[] += gout * d logits / d params, returns gout * d logits / d v
__device__ __forceinline__ float eb_logits_backward(const float* P, float v, float gout, float* gp) {
  const float* m = P;
  const float* b = P + 33;
  const float* f = P + 46;
  float a[4][3], th[4][3], l[4][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    a[0][i] = m[i] * v + b[i];
    th[0][i] = tanhf(a[0][i]);
    l[0][i] = a[0][i] + f[i] * th[0][i];
  }
#pragma unroll
  for (int k = 1; k <= 3; ++k) {
    const float* mk = m + 3 + (k - 1) * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      a[k][i] = mk[i * 3 + 0] * l[k - 1][0] + mk[i * 3 + 1] * l[k - 1][1] + mk[i * 3 + 2] * l[k - 1][2] + b[3 * k + i];
      th[k][i] = tanhf(a[k][i]);
      l[k][i] = a[k][i] + f[3 * k + i] * th[k][i];
    }
  }
  float dl[3], da[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) { gp[30 + j] += gout * l[3][j]; dl[j] = gout * m[30 + j]; }
  gp[33 + 12] += gout;
#pragma unroll
  for (int k = 3; k >= 1; --k) {
    const float* mk = m + 3 + (k - 1) * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      da[i] = dl[i] * (1.f + f[3 * k + i] * (1.f - th[k][i] * th[k][i]));
      gp[46 + 3 * k + i] += dl[i] * th[k][i];
      gp[33 + 3 * k + i] += da[i];
    }
    float nd[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        gp[3 + (k - 1) * 9 + i * 3 + j] += da[i] * l[k - 1][j];
        nd[j] += da[i] * mk[i * 3 + j];
      }
#pragma unroll
    for (int j = 0; j < 3; ++j) dl[j] = nd[j];
  }
  float dv = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float d0 = dl[i] * (1.f + f[i] * (1.f - th[0][i] * th[0][i]));
    gp[46 + i] += dl[i] * th[0][i];
    gp[33 + i] += d0;
    gp[i] += d0 * v;
    dv += d0 * m[i];
  }
  return dv;
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// one workgroup per channel: dz += gscale * d bits / d z, dparams[c][:] += gscale * d bits / d params (bits = -log2 lik)
__global__ __launch_bounds__(256) void eb_backward_kernel(FMap z, const float* params, FMap noise, float gscale, FMap dz, float* dparams) {
  __shared__ float red[4][EBP];
  const int c = blockIdx.x;
  const float* P = params + (long)c * EBP;
  const long npix = (long)z.H * z.W;
  const long count = npix * z.N;
  float gp[EBP];
#pragma unroll
  for (int i = 0; i < EBP; ++i) gp[i] = 0.f;
  for (long e = threadIdx.x; e < count; e += 256) {
    const int n = (int)(e / npix);
    const long pix = e % npix;
    const float zv = reinterpret_cast<const float*>(z.p)[(long)n * z.sn + pix * z.sp + c];
    const float v = zv + reinterpret_cast<const float*>(noise.p)[(long)n * noise.sn + pix * noise.sp + c];
    // forward values (pointwise.hip::eb_forward_kernel); the sign is a constant of the backward pass
    const float lo = eb_logits(P, v - 0.5f), up = eb_logits(P, v + 0.5f);
    const float ssum = lo + up;
    const float sign = ssum > 0.f ? -1.f : (ssum < 0.f ? 1.f : 0.f);
    const float su = sigm(sign * up), sl = sigm(sign * lo);
    const float diff = su - sl;
    const float lik = fabsf(diff);
    float dv = 0.f;
    // likelihood floor = compressai's LowerBound (ops/bound_ops.py): the gradient passes where lik >= bound OR it is
    // negative; d(-log2 max(lik, 1e-9)) / d lik is negative whenever gscale > 0, so a latent on the floor keeps a gradient
    const float likb = fmaxf(lik, 1e-9f);
    const float dlik = -gscale / (likb * 0.69314718055994531f);            // d(-log2 max(lik, 1e-9)) w.r.t. the bounded value
    if (lik >= 1e-9f || dlik < 0.f) {
      const float sd = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
      const float gup = dlik * sd * su * (1.f - su) * sign;
      const float glo = -dlik * sd * sl * (1.f - sl) * sign;
      dv = eb_logits_backward(P, v + 0.5f, gup, gp) + eb_logits_backward(P, v - 0.5f, glo, gp);
    }
    reinterpret_cast<float*>(dz.p)[(long)n * dz.sn + pix * dz.sp + c] += dv;
  }
  // reduce the 59 parameter gradients over the workgroup (wave shuffles, then 4 waves in LDS; fixed order)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < EBP; ++i) {
    float v = gp[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < EBP) dparams[(long)c * EBP + threadIdx.x] += red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Gaussian conditional with additive noise: bits = -log2 max(Phi((.5 - v)/s) - Phi((-.5 - v)/s), 1e-9), v = |y + noise - mean|,
// s = max(scale, 0.11): dy, dmean (= -dy), dscale (below the bound only a NEGATIVE gradient passes: compressai LowerBound)
__global__ void gc_backward_kernel(FMap y, FMap gp, FMap noise, float gscale, FMap dy, FMap dgp) {
  const long npix = (long)y.H * y.W;
  const long total = npix * y.C * y.N;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % y.C);
  const long t = i / y.C;
  const long pix = t % npix;
  const int n = (int)(t / npix);
  const float yv = reinterpret_cast<const float*>(y.p)[(long)n * y.sn + pix * y.sp + c];
  const float* g = reinterpret_cast<const float*>(gp.p) + (long)n * gp.sn + pix * gp.sp;
  const float sc = g[c], mean = g[y.C + c];
  const float s = fmaxf(sc, 0.11f);
  const float u = yv + reinterpret_cast<const float*>(noise.p)[(long)n * noise.sn + pix * noise.sp + c] - mean;
  const float v = fabsf(u);
  const float k = 0.70710678118654752440f;
  const float a = (0.5f - v) / s, b = (-0.5f - v) / s;
  const float lik = 0.5f * erfcf(-k * a) - 0.5f * erfcf(-k * b);
  // both bounds follow compressai's LowerBound backward (ops/bound_ops.py): pass where x >= bound OR the gradient is negative
  const float dlik = -gscale / (fmaxf(lik, 1e-9f) * 0.69314718055994531f);
  if (!(lik >= 1e-9f || dlik < 0.f)) return;
  const float inv_sqrt_2pi = 0.39894228040143267794f;
  const float pa = inv_sqrt_2pi * expf(-0.5f * a * a), pb = inv_sqrt_2pi * expf(-0.5f * b * b);
  const float dv = dlik * (pb - pa) / s;
  const float du = dv * (u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f));
  reinterpret_cast<float*>(dy.p)[(long)n * dy.sn + pix * dy.sp + c] += du;
  float* dg = reinterpret_cast<float*>(dgp.p) + (long)n * dgp.sn + pix * dgp.sp;
  dg[y.C + c] -= du;
  const float ds = dlik * (b * pb - a * pa) / s;          // gradient w.r.t. the bounded scale
  if (sc >= 0.11f || ds < 0.f) dg[c] += ds;
}

}  // namespace

extern "C" int tdvc_act_backward(const tdvc_fmap* g, const tdvc_fmap* y, const tdvc_fmap* res, int act, float slope, const tdvc_fmap* out, void* stream) {
  TDVC_CHECK(g && y && out && fmap_any(*g) && fmap_any(*y) && fmap_any(*out) && same_geom(*g, *y) && same_geom(*g, *out) && y->C >= g->C && out->C >= g->C && (g->C % 8) == 0,
             "tdvc_act_backward: bad arguments");
  if (res) TDVC_CHECK(fmap_any(*res) && same_geom(*g, *res) && res->C >= g->C, "tdvc_act_backward: bad residual");
  TDVC_CHECK(act == TDVC_ACT_RELU || act == TDVC_ACT_LRELU, "tdvc_act_backward: activation %d has no fused backward", act);
  const long total = (long)g->N * g->H * g->W * (g->C / 8);
  hipLaunchKernelGGL(act_backward_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*g), to_dev(*y), res ? to_dev(*res) : null_fmap(),
                     act == TDVC_ACT_RELU ? 0.f : slope, to_dev(*out));
  return tdvc_launch_status("tdvc_act_backward");
}

extern "C" int tdvc_pixel_unshuffle(const tdvc_fmap* y, const tdvc_fmap* out, void* stream) {
  TDVC_CHECK(y && out && fmap_any(*y) && fmap_any(*out) && (y->C % 8) == 0 && out->N == y->N && y->H == 2 * out->H && y->W == 2 * out->W && out->C == 4 * y->C,
             "tdvc_pixel_unshuffle: bad arguments");
  const long total = (long)out->N * out->H * out->W * (y->C / 8) * 4;
  hipLaunchKernelGGL(pixel_unshuffle_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*y), to_dev(*out));
  return tdvc_launch_status("tdvc_pixel_unshuffle");
}

extern "C" int64_t tdvc_bias_grad_work_floats(int N, int C_) { return (int64_t)N * 256 * C_; }

extern "C" int tdvc_bias_grad(const tdvc_fmap* g, int nvalid, const int32_t* dst_index, float scale, float* db, float* work, int64_t work_floats, void* stream) {
  TDVC_CHECK(g && db && work && fmap_any(*g) && (g->C % 8) == 0 && nvalid >= 1 && nvalid <= g->C, "tdvc_bias_grad: bad arguments");
  // pixel ranges per image: enough workgroups to stream the map, few enough rows for the serial second stage
  const long npix = (long)g->H * g->W;
  const int nblocks = (int)(npix / 1024 < 1 ? 1 : (npix / 1024 > 64 ? 64 : npix / 1024));
  TDVC_CHECK(work_floats >= tdvc_bias_grad_work_floats(g->N, g->C), "tdvc_bias_grad: workspace too small");
  hipLaunchKernelGGL(channel_sum_wide_kernel, dim3(nblocks, g->N, (g->C + 255) / 256), dim3(256), 0, ST(stream), to_dev(*g), work, nblocks);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((nvalid + 63) / 64), dim3(256), 0, ST(stream), work, g->N * nblocks, g->C, nvalid, dst_index, scale, db);
  return tdvc_launch_status("tdvc_bias_grad");
}

extern "C" int tdvc_copy_cast(const tdvc_fmap* src, const tdvc_fmap* dst, void* stream) {
  TDVC_CHECK(src && dst && fmap_any(*src) && fmap_any(*dst) && same_geom(*src, *dst) && dst->C >= src->C && (src->dtype == TDVC_F32 || (src->C % 8) == 0),
             "tdvc_copy_cast: bad arguments");
  const long total = (long)dst->N * dst->H * dst->W * ((dst->C + 7) / 8);
  hipLaunchKernelGGL(copy_cast_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*src), to_dev(*dst));
  return tdvc_launch_status("tdvc_copy_cast");
}

extern "C" int tdvc_clamp01_backward(const tdvc_fmap* g, const tdvc_fmap* y, void* stream) {
  TDVC_CHECK(g && y && fmap_any(*g) && fmap_any(*y) && same_geom(*g, *y) && y->C >= g->C && (g->C % 8) == 0, "tdvc_clamp01_backward: bad arguments");
  const long total = (long)g->N * g->H * g->W * (g->C / 8);
  hipLaunchKernelGGL(clamp01_backward_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*g), to_dev(*y));
  return tdvc_launch_status("tdvc_clamp01_backward");
}

extern "C" int64_t tdvc_gate_backward_work_floats(int N, int C_) { return (int64_t)N * 256 * C_; }

extern "C" int tdvc_gate_backward(const tdvc_fmap* g, const tdvc_fmap* a, const float* gate, const tdvc_fmap* da, float* dgate, float* work,
                                  int64_t work_floats, void* stream) {
  TDVC_CHECK(g && a && gate && dgate && work && fmap_any(*g) && fmap_any(*a) && same_geom(*g, *a) && a->C == g->C && (g->C % 8) == 0 && g->C <= 256,
             "tdvc_gate_backward: bad arguments");
  if (da) TDVC_CHECK(fmap_any(*da) && same_geom(*g, *da) && da->C == g->C, "tdvc_gate_backward: bad da");
  TDVC_CHECK(work_floats >= tdvc_gate_backward_work_floats(g->N, g->C), "tdvc_gate_backward: workspace too small");
  const long npix = (long)g->H * g->W;
  const int nblocks = (int)(npix / 256 < 1 ? 1 : (npix / 256 > 256 ? 256 : npix / 256));
  hipLaunchKernelGGL(gate_backward_kernel, dim3(nblocks, g->N), dim3(256), 0, ST(stream), to_dev(*g), to_dev(*a), gate, da ? to_dev(*da) : null_fmap(), work, nblocks);
  hipLaunchKernelGGL(reduce_groups_kernel, dim3((g->N * g->C + 255) / 256), dim3(256), 0, ST(stream), work, g->N, nblocks, g->C, dgate);
  return tdvc_launch_status("tdvc_gate_backward");
}

extern "C" int tdvc_se_gate_backward(const float* partial, int nblocks, float inv_count, int N, int C_, int Cmid, const float* w1, const float* b1,
                                     const float* w2, const float* b2, const float* gate, const float* dgate, float scale, float* dmean,
                                     float* dw1, float* db1, float* dw2, float* db2, void* stream) {
  TDVC_CHECK(partial && w1 && b1 && w2 && b2 && gate && dgate && dmean && dw1 && db1 && dw2 && db2 && C_ <= 256 && Cmid <= 32 && N >= 1,
             "tdvc_se_gate_backward: bad arguments");
  const size_t lds = sizeof(float) * ((size_t)Cmid * C_ + (size_t)C_ * (Cmid + 1) + 2 * (size_t)N * C_ + 64);
  TDVC_CHECK(lds <= 160 * 1024, "tdvc_se_gate_backward: batch too large for one workgroup's LDS");
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(se_gate_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(se_gate_backward_kernel, dim3(1), dim3(1024), lds, ST(stream), partial, nblocks, inv_count, N, C_, Cmid, w1, b1, w2, b2, gate, dgate,
                     scale, dmean, dw1, db1, dw2, db2);
  return tdvc_launch_status("tdvc_se_gate_backward");
}

extern "C" int tdvc_bcast_channel_add(const tdvc_fmap* dx, const float* v, float scale, void* stream) {
  TDVC_CHECK(dx && v && fmap_any(*dx) && (dx->C % 8) == 0, "tdvc_bcast_channel_add: bad arguments");
  const long total = (long)dx->N * dx->H * dx->W * (dx->C / 8);
  hipLaunchKernelGGL(bcast_channel_add_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*dx), v, scale);
  return tdvc_launch_status("tdvc_bcast_channel_add");
}

extern "C" int tdvc_add_flow_backward(const tdvc_fmap* doff, const tdvc_fmap* dflow, void* stream) {
  TDVC_CHECK(doff && dflow && fmap_ok16(*doff) && fmap_ok32(*dflow) && dflow->C >= 2 && same_geom(*doff, *dflow), "tdvc_add_flow_backward: bad arguments");
  const long total = (long)doff->N * doff->H * doff->W;
  hipLaunchKernelGGL(add_flow_backward_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*doff), to_dev(*dflow));
  return tdvc_launch_status("tdvc_add_flow_backward");
}

extern "C" int tdvc_bcast_add_act_backward(const tdvc_fmap* dx, const tdvc_fmap* x, const tdvc_fmap* db, float slope, void* stream) {
  TDVC_CHECK(dx && x && db && fmap_ok16(*dx) && fmap_ok16(*x) && fmap_ok16(*db) && same_geom(*dx, *x) && same_geom(*dx, *db) && dx->C == x->C &&
                 (dx->C % db->C) == 0, "tdvc_bcast_add_act_backward: bad arguments");
  const long total = (long)dx->N * dx->H * dx->W * (db->C / 8);
  hipLaunchKernelGGL(bcast_add_act_backward_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*dx), to_dev(*x), to_dev(*db), slope);
  return tdvc_launch_status("tdvc_bcast_add_act_backward");
}

extern "C" int tdvc_upsample2x_backward(const tdvc_fmap* dy, const tdvc_fmap* dx, void* stream) {
  TDVC_CHECK(dy && dx && fmap_ok16(*dy) && fmap_ok16(*dx) && dy->H == 2 * dx->H && dy->W == 2 * dx->W && dy->C == dx->C && dy->N == dx->N,
             "tdvc_upsample2x_backward: bad arguments");
  const long total = (long)dx->N * dx->H * dx->W * (dx->C / 8);
  hipLaunchKernelGGL(upsample2x_backward_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*dy), to_dev(*dx));
  return tdvc_launch_status("tdvc_upsample2x_backward");
}

extern "C" int tdvc_resize_bilinear_backward(const tdvc_fmap* dy, const tdvc_fmap* dx, const float* chscale, void* stream) {
  TDVC_CHECK(dy && dx && fmap_ok32(*dy) && fmap_ok32(*dx) && dy->N == dx->N && dy->C == dx->C, "tdvc_resize_bilinear_backward: two fp32 fmaps of equal batch and channels expected");
  const long total = (long)dx->N * dx->H * dx->W;
  hipLaunchKernelGGL(resize_bilinear_backward_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*dy), to_dev(*dx), chscale);
  return tdvc_launch_status("tdvc_resize_bilinear_backward");
}

extern "C" int tdvc_spynet_level_input_backward(const tdvc_fmap* supp, const tdvc_fmap* flow_up, const tdvc_fmap* dcat8, const tdvc_fmap* dflow_up,
                                                const tdvc_fmap* dflow_lo, void* stream) {
  TDVC_CHECK(supp && flow_up && dcat8 && dflow_up && fmap_ok32(*supp) && supp->C >= 3 && fmap_ok32(*flow_up) && flow_up->C >= 2 &&
                 fmap_ok32(*dflow_up) && dflow_up->C >= 2 && fmap_ok16(*dcat8) && dcat8->C == 8 && same_geom(*supp, *flow_up) &&
                 same_geom(*supp, *dcat8) && same_geom(*supp, *dflow_up), "tdvc_spynet_level_input_backward: bad arguments");
  const long total = (long)supp->N * supp->H * supp->W;
  hipLaunchKernelGGL(spynet_warp_backward_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*supp), to_dev(*flow_up), to_dev(*dcat8), to_dev(*dflow_up));
  if (dflow_lo) {
    TDVC_CHECK(fmap_ok32(*dflow_lo) && dflow_lo->C >= 2 && dflow_lo->N == supp->N, "tdvc_spynet_level_input_backward: bad dflow_lo");
    const long tl = (long)dflow_lo->N * dflow_lo->H * dflow_lo->W;
    hipLaunchKernelGGL(upsample_ac_backward_kernel, grid1d(tl), dim3(EW_BLOCK), 0, ST(stream), to_dev(*dflow_up), to_dev(*dflow_lo));
  }
  return tdvc_launch_status("tdvc_spynet_level_input_backward");
}

extern "C" int tdvc_sigmoid_f32(const float* x, float* out, int64_t n, void* stream) {
  TDVC_CHECK(x && out && n >= 0, "tdvc_sigmoid_f32: bad arguments");
  if (n) hipLaunchKernelGGL(sigmoid_f32_kernel, grid1d(n), dim3(EW_BLOCK), 0, ST(stream), x, out, (long)n);
  return tdvc_launch_status("tdvc_sigmoid_f32");
}
extern "C" int tdvc_sigmoid_backward_f32(float* g, const float* s, int64_t n, void* stream) {
  TDVC_CHECK(g && s && n >= 0, "tdvc_sigmoid_backward_f32: bad arguments");
  if (n) hipLaunchKernelGGL(sigmoid_backward_f32_kernel, grid1d(n), dim3(EW_BLOCK), 0, ST(stream), g, s, (long)n);
  return tdvc_launch_status("tdvc_sigmoid_backward_f32");
}
extern "C" int tdvc_axpy_f32(float* dst, const float* src, float scale, int64_t n, void* stream) {
  TDVC_CHECK(dst && src && n >= 0, "tdvc_axpy_f32: bad arguments");
  if (n) hipLaunchKernelGGL(axpy_f32_kernel, grid1d(n), dim3(EW_BLOCK), 0, ST(stream), dst, src, scale, (long)n);
  return tdvc_launch_status("tdvc_axpy_f32");
}

extern "C" int tdvc_match_gather_backward(const tdvc_fmap* fin, const tdvc_fmap* fref, const int32_t* idx, int scale, int hp, int wp,
                                          const tdvc_fmap* dcat, const tdvc_fmap* dfin, const tdvc_fmap* dfref, void* stream) {
  TDVC_CHECK(fin && fref && idx && dcat && dfin && dfref && fmap_ok16(*fin) && fmap_ok16(*fref) && fmap_ok16(*dcat) && fmap_ok16(*dfin) && fmap_ok16(*dfref),
             "tdvc_match_gather_backward: bad fmaps");
  TDVC_CHECK(fin->C == 64 && fref->C == 64 && dcat->C == 128 && dfin->C == 64 && dfref->C == 64 && same_geom(*fin, *fref) && same_geom(*fin, *dcat) &&
                 same_geom(*fin, *dfin) && same_geom(*fin, *dfref), "tdvc_match_gather_backward: needs C=64 / dcat C=128");
  const int ks = 3 * scale;
  const int nbh = (fin->H + ks) / ks + 1, nbw = (fin->W + ks) / ks + 1;
  TDVC_CHECK(nbh == (hp + 3) / 3 + 1 && nbw == (wp + 3) / 3 + 1, "tdvc_match_gather_backward: fold grid != patch grid");
  const long total = (long)fin->N * fin->H * fin->W * 8;
  hipLaunchKernelGGL(match_gather_backward_in_kernel, grid1d(total), dim3(256), 0, ST(stream), to_dev(*fin), to_dev(*fref), idx, ks, nbh, nbw, to_dev(*dcat), to_dev(*dfin));
  hipLaunchKernelGGL(match_gather_backward_ref_kernel, grid1d(total), dim3(256), 0, ST(stream), to_dev(*fin), to_dev(*fref), idx, ks, nbh, nbw, to_dev(*dcat), to_dev(*dfref));
  return tdvc_launch_status("tdvc_match_gather_backward");
}

extern "C" int tdvc_gdn_backward(const tdvc_fmap* g, const tdvc_fmap* x, const tdvc_fmap* n32, int inverse, const tdvc_fmap* dn, const tdvc_fmap* dx, void* stream) {
  TDVC_CHECK(g && x && n32 && dn && dx && fmap_any(*g) && fmap_ok16(*x) && fmap_ok32(*n32) && fmap_ok16(*dn) && fmap_ok16(*dx) && same_geom(*g, *x) &&
                 same_geom(*g, *n32) && same_geom(*g, *dn) && same_geom(*g, *dx) && (g->C % 8) == 0 && x->C >= g->C && n32->C >= g->C && dn->C >= g->C &&
                 dx->C >= g->C, "tdvc_gdn_backward: bad arguments");
  const long total = (long)g->N * g->H * g->W * (g->C / 8);
  hipLaunchKernelGGL(gdn_backward_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*g), to_dev(*x), to_dev(*n32), inverse, to_dev(*dn), to_dev(*dx));
  return tdvc_launch_status("tdvc_gdn_backward");
}

extern "C" int tdvc_mul2_accumulate(const tdvc_fmap* dx, const tdvc_fmap* x, const tdvc_fmap* t, void* stream) {
  TDVC_CHECK(dx && x && t && fmap_ok16(*dx) && fmap_ok16(*x) && fmap_ok16(*t) && same_geom(*dx, *x) && same_geom(*dx, *t) && x->C >= dx->C && t->C >= dx->C,
             "tdvc_mul2_accumulate: bad arguments");
  const long total = (long)dx->N * dx->H * dx->W * (dx->C / 8);
  hipLaunchKernelGGL(mul2_accumulate_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*dx), to_dev(*x), to_dev(*t));
  return tdvc_launch_status("tdvc_mul2_accumulate");
}

// ---- the factorised prior's PARAMETER-SPACE work of a training step as three launches (it was ~170 torch launches per coder: softplus / tanh
// / matmul / add chains with their autograd twins, 128 x 59 values each).  Column layout of the packed table = eb_logits()'s: softplus(matrix0..4)
// [33] | bias0..4 [13] | tanh(factor0..3) [12] | median.  `raw` / `grad`: device tables of the 14 parameter tensors in that order (each [C][len]).
struct EbCol { int k, j, len; };
__device__ __forceinline__ EbCol eb_col(int col) {      // packed column -> (parameter index, element inside the channel's block, block length)
  constexpr int start[15] = {0, 3, 12, 21, 30, 33, 36, 39, 42, 45, 46, 49, 52, 55, 58};
  int k = 0;
#pragma unroll
  for (int i = 1; i < 14; ++i) k += col >= start[i] ? 1 : 0;
  return EbCol{k, col - start[k], start[k + 1] - start[k]};
}
__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }      // torch.nn.functional.softplus (beta 1, threshold 20)

__global__ void eb_pack_kernel(const float* const* raw, const float* quantiles, float* packed, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * EBP) return;
  const int c = i / EBP, col = i - c * EBP;
  if (col == 58) { packed[i] = quantiles[c * 3 + 1]; return; }
  const EbCol e = eb_col(col);
  const float x = raw[e.k][c * e.len + e.j];
  packed[i] = col < 33 ? softplus_t(x) : (col < 46 ? x : tanhf(x));
}

// grad[k][c][j] += scale * dpacked[c][col] * d packed / d raw (softplus' = sigmoid, with torch's threshold; tanh' = 1 - tanh^2); the median column
// (the quantiles) only receives the auxiliary loss
__global__ void eb_param_chain_kernel(const float* dpacked, const float* const* raw, float* const* grad, float scale, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * EBP) return;
  const int c = i / EBP, col = i - c * EBP;
  if (col == 58) return;
  const EbCol e = eb_col(col);
  const float x = raw[e.k][c * e.len + e.j];
  float d = 1.f;
  if (col < 33) d = x > 20.f ? 1.f : sigm(x);
  else if (col >= 46) { const float t = tanhf(x); d = 1.f - t * t; }
  grad[e.k][c * e.len + e.j] += scale * dpacked[i] * d;
}

// compressai's auxiliary loss: sum_c sum_j |logits_cumulative(quantiles[c][j]) - target_j| (targets -t, 0, +t), parameters held constant;
// dq[c][j] = its gradient (overwritten), loss[0] = its value.  One workgroup (3 C <= 1024 threads), fixed summation order.
__global__ __launch_bounds__(1024) void eb_aux_kernel(const float* params, const float* quantiles, float target, float* dq, float* loss, int C) {
  __shared__ float sh[16];
  const int i = threadIdx.x;
  float l = 0.f;
  if (i < 3 * C) {
    const int c = i / 3, j = i - 3 * c;
    const float* P = params + (long)c * EBP;
    const float v = quantiles[i];
    const float tj = j == 0 ? -target : (j == 2 ? target : 0.f);
    const float r = eb_logits(P, v) - tj;
    const float sg = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);
    float gp[EBP];
#pragma unroll
    for (int q = 0; q < EBP; ++q) gp[q] = 0.f;
    dq[i] = eb_logits_backward(P, v, sg, gp);
    l = fabsf(r);
  }
  for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
  if ((i & 63) == 0) sh[i >> 6] = l;
  __syncthreads();
  if (i == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    loss[0] = t;
  }
}

extern "C" int tdvc_eb_pack(const float* const* raw, const float* quantiles, float* packed, int C, void* stream) {
  TDVC_CHECK(raw && quantiles && packed && C >= 1, "tdvc_eb_pack: bad arguments");
  hipLaunchKernelGGL(eb_pack_kernel, grid1d((long)C * EBP, 256), dim3(256), 0, ST(stream), raw, quantiles, packed, C);
  return tdvc_launch_status("tdvc_eb_pack");
}

extern "C" int tdvc_eb_param_chain(const float* dpacked, const float* const* raw, float* const* grad, float scale, int C, void* stream) {
  TDVC_CHECK(dpacked && raw && grad && C >= 1, "tdvc_eb_param_chain: bad arguments");
  hipLaunchKernelGGL(eb_param_chain_kernel, grid1d((long)C * EBP, 256), dim3(256), 0, ST(stream), dpacked, raw, grad, scale, C);
  return tdvc_launch_status("tdvc_eb_param_chain");
}

extern "C" int tdvc_eb_aux(const float* params, const float* quantiles, float target, float* dq, float* loss, int C, void* stream) {
  TDVC_CHECK(params && quantiles && dq && loss && C >= 1 && 3 * C <= 1024, "tdvc_eb_aux: bad arguments (3 C <= 1024)");
  const int threads = ((3 * C + 63) / 64) * 64;
  hipLaunchKernelGGL(eb_aux_kernel, dim3(1), dim3(threads), 0, ST(stream), params, quantiles, target, dq, loss, C);
  return tdvc_launch_status("tdvc_eb_aux");
}

extern "C" int tdvc_eb_backward(const tdvc_fmap* z, const float* params, const tdvc_fmap* noise, float gscale, const tdvc_fmap* dz, float* dparams, void* stream) {
  TDVC_CHECK(z && params && noise && dz && dparams && fmap_ok32(*z) && fmap_ok32(*noise) && fmap_ok32(*dz) && same_geom(*z, *noise) && same_geom(*z, *dz) &&
                 noise->C >= z->C && dz->C >= z->C, "tdvc_eb_backward: bad arguments");
  hipLaunchKernelGGL(eb_backward_kernel, dim3(z->C), dim3(256), 0, ST(stream), to_dev(*z), params, to_dev(*noise), gscale, to_dev(*dz), dparams);
  return tdvc_launch_status("tdvc_eb_backward");
}

extern "C" int tdvc_gc_backward(const tdvc_fmap* y, const tdvc_fmap* gp, const tdvc_fmap* noise, float gscale, const tdvc_fmap* dy, const tdvc_fmap* dgp, void* stream) {
  TDVC_CHECK(y && gp && noise && dy && dgp && fmap_ok32(*y) && fmap_ok32(*gp) && fmap_ok32(*noise) && fmap_ok32(*dy) && fmap_ok32(*dgp) && same_geom(*y, *gp) &&
                 same_geom(*y, *noise) && same_geom(*y, *dy) && same_geom(*y, *dgp) && gp->C >= 2 * y->C && dgp->C >= 2 * y->C && dy->C >= y->C,
             "tdvc_gc_backward: bad arguments");
  const long total = (long)y->N * y->H * y->W * y->C;
  hipLaunchKernelGGL(gc_backward_kernel, grid1d(total), dim3(EW_BLOCK), 0, ST(stream), to_dev(*y), to_dev(*gp), to_dev(*noise), gscale, to_dev(*dy), to_dev(*dgp));
  return tdvc_launch_status("tdvc_gc_backward");
}
