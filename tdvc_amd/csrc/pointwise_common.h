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

inline bool fmap_any(const tdvc_fmap& f) { return f.dtype == TDVC_F32 ? fmap_ok32(f) : fmap_ok16(f); }
inline bool same_geom(const tdvc_fmap& a, const tdvc_fmap& b) { return a.N == b.N && a.H == b.H && a.W == b.W; }
#define ST(s) reinterpret_cast<hipStream_t>(s)

}  // namespace
