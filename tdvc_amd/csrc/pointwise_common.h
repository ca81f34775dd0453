// Helpers shared by the streaming (HBM-bound) kernels: 8-channel accesses on fp16 / fp32 fmaps, 1-D grids.
#pragma once
#include "common.h"

namespace {

constexpr int EW_BLOCK = 256;

__device__ __forceinline__ void load8(const FMap& f, int n, long pix, int c, float v[8]) {
  if (f.f32) {
    const float* p = reinterpret_cast<const float*>(f.p) + (long)n * f.sn + pix * f.sp + c;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (c + j < f.C) ? p[j] : 0.f;
  } else {
    const half8 h = *reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(f.p) + (long)n * f.sn + pix * f.sp + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
  }
}
__device__ __forceinline__ void store8(const FMap& f, int n, long pix, int c, const float v[8]) {
  if (f.f32) {
    float* p = reinterpret_cast<float*>(f.p) + (long)n * f.sn + pix * f.sp + c;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c + j < f.C) p[j] = v[j];
  } else {
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (half_t)v[j];
    *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(f.p) + (long)n * f.sn + pix * f.sp + c) = h;
  }
}

inline dim3 grid1d(long total, int block = EW_BLOCK) { return dim3((unsigned)((total + block - 1) / block)); }

// ---- factorised-prior logits (compressai EntropyBottleneck._logits_cumulative, filters (3,3,3,3)), packed parameters
constexpr int EB_NP = 59;   // floats per channel, see tdvc_amd/entropy.py::pack_eb_params

__device__ __forceinline__ float eb_logits(const float* P, float v) {
  // filters (1,3,3,3,3,1): m0[3], m1..m3[9], m4[3] | b0..b3[3], b4[1] | f0..f3[3] | median
  const float* m = P;
  const float* b = P + 33;
  const float* f = P + 46;
  float l[3], t[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    l[i] = m[i] * v + b[i];
    l[i] += f[i] * tanhf(l[i]);
  }
#pragma unroll
  for (int k = 1; k <= 3; ++k) {
    const float* mk = m + 3 + (k - 1) * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      t[i] = mk[i * 3 + 0] * l[0] + mk[i * 3 + 1] * l[1] + mk[i * 3 + 2] * l[2] + b[3 * k + i];
      t[i] += f[3 * k + i] * tanhf(t[i]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) l[i] = t[i];
  }
  return m[30] * l[0] + m[31] * l[1] + m[32] * l[2] + b[12];
}

inline bool fmap_any(const tdvc_fmap& f) { return f.dtype == TDVC_F32 ? fmap_ok32(f) : fmap_ok16(f); }
inline bool same_geom(const tdvc_fmap& a, const tdvc_fmap& b) { return a.N == b.N && a.H == b.H && a.W == b.W; }
#define ST(s) reinterpret_cast<hipStream_t>(s)

}  // namespace
