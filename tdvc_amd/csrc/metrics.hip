// Quality metrics of the evaluation loop (tools/predict.py:87-100 in the reference): one level of MS-SSIM
// (main/model/ms_ssim_torch.py:33-83: valid separable Gaussian of X, Y, X^2, Y^2, XY; cs and ssim maps; per-image
// means) and the 2x2 average pooling between levels (:178-180, padding = size % 2, zeros counted).
//
// HBM-bound: one pass reads X and Y once per level (8 B per pixel per channel) -- the reference runs ten depthwise
// convolutions over five full-size fp32 products.  fp32 throughout, like the reference on a float input.
#include <hip/hip_runtime.h>

#include "common.h"
#include "../../include/tdvc_hip.h"

namespace {

constexpr int MS_TW = 32, MS_TH = 16, MS_MAXWIN = 15;

struct SsimLevel {
  const float *x, *y;
  int N, C, H, W, win;
  float taps[MS_MAXWIN];
  float c1, c2;
  int tiles_x, tiles_y;
  float* partial;               // [N][C * tiles][2]  (sum ssim_map, sum cs_map)
};

__global__ __launch_bounds__(256) void ssim_level_kernel(const SsimLevel p) {
  extern __shared__ float sm[];
  const int IW = MS_TW + p.win - 1, IH = MS_TH + p.win - 1;
  float* tx = sm;                         // [IH][IW] X tile
  float* ty = tx + IH * IW;               // [IH][IW] Y tile
  float* hz = ty + IH * IW;               // [5][IH][MS_TW] horizontally filtered X, Y, XX, YY, XY
  __shared__ float red[2][4];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const int tcol = tile % p.tiles_x, trow = tile / p.tiles_x;
  const int Hv = p.H - p.win + 1, Wv = p.W - p.win + 1;
  const int oy0 = trow * MS_TH, ox0 = tcol * MS_TW;
  const float* xp = p.x + ((long)n * p.C + c) * p.H * p.W;
  const float* yp = p.y + ((long)n * p.C + c) * p.H * p.W;
  for (int i = tid; i < IH * IW; i += 256) {
    const int r = i / IW, q = i - r * IW;
    const int iy = oy0 + r, ix = ox0 + q;
    const bool ok = iy < p.H && ix < p.W;
    tx[i] = ok ? xp[(long)iy * p.W + ix] : 0.f;
    ty[i] = ok ? yp[(long)iy * p.W + ix] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < IH * MS_TW; i += 256) {
    const int r = i / MS_TW, q = i - r * MS_TW;
    float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
    for (int k = 0; k < p.win; ++k) {
      const float w = p.taps[k], u = tx[r * IW + q + k], v = ty[r * IW + q + k];
      a += w * u; b += w * v; aa += w * (u * u); bb += w * (v * v); ab += w * (u * v);
    }
    hz[i] = a; hz[IH * MS_TW + i] = b; hz[2 * IH * MS_TW + i] = aa; hz[3 * IH * MS_TW + i] = bb; hz[4 * IH * MS_TW + i] = ab;
  }
  __syncthreads();
  float s_ssim = 0.f, s_cs = 0.f;
  for (int i = tid; i < MS_TH * MS_TW; i += 256) {
    const int r = i / MS_TW, q = i - r * MS_TW;
    if (oy0 + r >= Hv || ox0 + q >= Wv) continue;
    float mu1 = 0.f, mu2 = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
    for (int k = 0; k < p.win; ++k) {
      const float w = p.taps[k];
      const int o = (r + k) * MS_TW + q;
      mu1 += w * hz[o]; mu2 += w * hz[IH * MS_TW + o]; xx += w * hz[2 * IH * MS_TW + o]; yy += w * hz[3 * IH * MS_TW + o];
      xy += w * hz[4 * IH * MS_TW + o];
    }
    const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = xx - mu1s, s2 = yy - mu2s, s12 = xy - mu12;
    const float cs = (2.f * s12 + p.c2) / (s1 + s2 + p.c2);
    s_cs += cs;
    s_ssim += ((2.f * mu12 + p.c1) / (mu1s + mu2s + p.c1)) * cs;
  }
  // block sum in a fixed order: lanes by shuffle, waves through LDS
  for (int o = 32; o > 0; o >>= 1) {
    s_ssim += __shfl_down(s_ssim, o);
    s_cs += __shfl_down(s_cs, o);
  }
  if ((tid & 63) == 0) { red[0][tid >> 6] = s_ssim; red[1][tid >> 6] = s_cs; }
  __syncthreads();
  if (tid == 0) {
    float* o = p.partial + (((long)n * p.C + c) * (p.tiles_x * p.tiles_y) + tile) * 2;
    o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// out[n] = sum over the image's partials / count, fixed order (one workgroup per image, double accumulation)
__global__ __launch_bounds__(256) void ssim_reduce_kernel(const float* partial, int per_image, double inv_count, float* ssim, float* cs) {
  __shared__ double red[2][256];
  const int n = blockIdx.x, tid = threadIdx.x;
  double a = 0.0, b = 0.0;
  for (int i = tid; i < per_image; i += 256) {
    a += (double)partial[((long)n * per_image + i) * 2];
    b += (double)partial[((long)n * per_image + i) * 2 + 1];
  }
  red[0][tid] = a; red[1][tid] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { ssim[n] = (float)(red[0][0] * inv_count); cs[n] = (float)(red[1][0] * inv_count); }
}

// F.avg_pool2d(x, kernel_size=2, padding=(H % 2, W % 2)): stride 2, zeros of the padding counted (divisor 4)
__global__ void avgpool2_pad_kernel(const float* x, long planes, int H, int W, int ph, int pw, int Ho, int Wo, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= planes * Ho * Wo) return;
  const int ox = (int)(i % Wo);
  const long q = i / Wo;
  const int oy = (int)(q % Ho);
  const long pl = q / Ho;
  const float* xp = x + pl * H * W;
  float s = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int iy = 2 * oy - ph + dy, ix = 2 * ox - pw + dx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) s += xp[(long)iy * W + ix];
    }
  out[i] = 0.25f * s;
}

}  // namespace

extern "C" int64_t tdvc_ssim_level_work_floats(int N, int C, int H, int W, int win) {
  if (N <= 0 || C <= 0 || win < 1 || win > MS_MAXWIN || (win & 1) == 0 || H < win || W < win) return TDVC_EINVAL;
  const int tiles_x = (W - win + 1 + MS_TW - 1) / MS_TW, tiles_y = (H - win + 1 + MS_TH - 1) / MS_TH;
  return (int64_t)N * C * tiles_x * tiles_y * 2;
}

extern "C" int tdvc_ssim_level(const float* x, const float* y, int N, int C, int H, int W, const float* win, int win_size,
                               float c1, float c2, float* ssim_out, float* cs_out, float* work, int64_t work_floats, void* stream) {
  TDVC_CHECK(x && y && win && ssim_out && cs_out && work, "tdvc_ssim_level: null pointer");
  const int64_t need = tdvc_ssim_level_work_floats(N, C, H, W, win_size);
  TDVC_CHECK(need > 0, "tdvc_ssim_level: bad geometry (N %d C %d H %d W %d window %d: odd window <= %d, image >= window)", N, C, H, W, win_size, MS_MAXWIN);
  TDVC_CHECK(work_floats >= need, "tdvc_ssim_level: workspace too small");
  TDVC_CHECK(N <= 65535 && C <= 65535, "tdvc_ssim_level: too many images / channels for one launch");
  SsimLevel p;
  p.x = x; p.y = y; p.N = N; p.C = C; p.H = H; p.W = W; p.win = win_size;
  for (int k = 0; k < MS_MAXWIN; ++k) p.taps[k] = k < win_size ? win[k] : 0.f;        // `win` is HOST memory (the 1-D kernel)
  p.c1 = c1; p.c2 = c2;
  p.tiles_x = (W - win_size + 1 + MS_TW - 1) / MS_TW; p.tiles_y = (H - win_size + 1 + MS_TH - 1) / MS_TH;
  p.partial = work;
  const int IW = MS_TW + win_size - 1, IH = MS_TH + win_size - 1;
  const size_t lds = sizeof(float) * ((size_t)2 * IH * IW + (size_t)5 * IH * MS_TW);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(ssim_level_kernel, dim3(p.tiles_x * p.tiles_y, C, N), dim3(256), lds, st, p);
  const int per_image = C * p.tiles_x * p.tiles_y;
  const double inv = 1.0 / ((double)C * (H - win_size + 1) * (W - win_size + 1));
  hipLaunchKernelGGL(ssim_reduce_kernel, dim3(N), dim3(256), 0, st, work, per_image, inv, ssim_out, cs_out);
  return tdvc_launch_status("tdvc_ssim_level");
}

extern "C" int tdvc_avgpool2_pad_f32(const float* x, int64_t planes, int H, int W, float* out, void* stream) {
  TDVC_CHECK(x && out && planes > 0 && H > 0 && W > 0, "tdvc_avgpool2_pad_f32: bad arguments");
  const int ph = H % 2, pw = W % 2;
  const int Ho = (H + 2 * ph - 2) / 2 + 1, Wo = (W + 2 * pw - 2) / 2 + 1;
  const long total = planes * Ho * Wo;
  hipLaunchKernelGGL(avgpool2_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, (long)planes,
                     H, W, ph, pw, Ho, Wo, out);
  return tdvc_launch_status("tdvc_avgpool2_pad_f32");
}
