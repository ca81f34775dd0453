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

// out[dst[c] or c] += scale * sum_rows partial[row][c]  (rows summed in order)
__global__ void reduce_rows_kernel(const float* partial, int rows, int C_, int nvalid, const int* dst, float scale, float* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nvalid) return;
  float s = 0.f;
  for (int r = 0; r < rows; ++r) s += partial[(long)r * C_ + c];
  const int d = dst ? dst[c] : c;
  if (d >= 0) out[d] += s * scale;
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

extern "C" int64_t tdvc_bias_grad_work_floats(int N, int C_) { return (int64_t)N * 64 * C_; }

extern "C" int tdvc_bias_grad(const tdvc_fmap* g, int nvalid, const int32_t* dst_index, float scale, float* db, float* work, int64_t work_floats, void* stream) {
  TDVC_CHECK(g && db && work && fmap_any(*g) && (g->C % 8) == 0 && nvalid >= 1 && nvalid <= g->C, "tdvc_bias_grad: bad arguments");
  const int nblocks = 64;
  TDVC_CHECK(work_floats >= tdvc_bias_grad_work_floats(g->N, g->C), "tdvc_bias_grad: workspace too small");
  hipLaunchKernelGGL(channel_sum_wide_kernel, dim3(nblocks, g->N, (g->C + 255) / 256), dim3(256), 0, ST(stream), to_dev(*g), work, nblocks);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((nvalid + 255) / 256), dim3(256), 0, ST(stream), work, g->N * nblocks, g->C, nvalid, dst_index, scale, db);
  return tdvc_launch_status("tdvc_bias_grad");
}
