// Shared helpers for the gfx950 kernels of libtdvc_hip.so (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/tdvc_hip.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void tdvc_set_error(const char* fmt, ...);
// per-device scratch (lib.cpp): >= 4 KB of zeros nobody writes, >= 16 KB dump page nobody reads; 0 or a HIP error code
int tdvc_scratch_pages(const void** zeros, void** dump);

#define TDVC_CHECK(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      tdvc_set_error(__VA_ARGS__);       \
      return TDVC_EINVAL;                \
    }                                    \
  } while (0)

static inline int tdvc_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    tdvc_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return TDVC_OK;
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
// "done once" flags of per-DEVICE state (hipFuncSetAttribute applies to the current device only): one bool per device index; a
// process that drives a second GPU sets the attribute there too.  Racing first launches on two threads set it twice: harmless.
struct TdvcPerDeviceFlag {
  bool done[64] = {};
  bool& flag() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return done[dev];
  }
};

// device-side view of a tdvc_fmap
struct FMap {
  void* p;
  int N, H, W, C;
  long sn;
  int sp;
  int f32;
};
static inline FMap to_dev(const tdvc_fmap& f) {
  FMap m;
  m.p = f.p; m.N = f.N; m.H = f.H; m.W = f.W; m.C = f.C; m.sn = f.sn; m.sp = f.sp;
  m.f32 = (f.dtype == TDVC_F32);
  return m;
}
static inline FMap null_fmap() {
  FMap m; memset(&m, 0, sizeof(m)); return m;
}

// fp16 fmap validity: channels / stride multiples of 8, 16-byte aligned base
static inline bool fmap_ok16(const tdvc_fmap& f) {
  return f.p && f.dtype == TDVC_F16 && (f.C % 8) == 0 && (f.sp % 8) == 0 && (f.sn % 8) == 0 && aligned16(f.p) &&
         f.sp >= f.C && f.N > 0 && f.H > 0 && f.W > 0;
}
static inline bool fmap_ok32(const tdvc_fmap& f) {
  return f.p && f.dtype == TDVC_F32 && f.sp >= f.C && f.N > 0 && f.H > 0 && f.W > 0 && (((uintptr_t)f.p) & 3) == 0;
}

// XCD-aware tile index for one-workgroup-per-tile launches: workgroups go round-robin over the 8 XCDs (each with its own
// L2), so neighbouring blockIdx.x values sit on different L2s.  Workgroup b is given tile (b % 8) * band + b / 8: the
// tiles an XCD works on are contiguous in the raster and their shared halo lines meet in one L2.  A bijection on
// [0, gridDim.x); the identity when neither gridDim.x % 8 == 0 nor the grid is one-dimensional (then b % 8 is not the XCD).
__device__ __forceinline__ int tdvc_xcd_tile(int b) {
  const int gx = (int)gridDim.x;
  const bool planar = gridDim.y * gridDim.z == 1;
  if ((gx & 7) != 0 && !planar) return b;
  const int q = gx >> 3, r = gx & 7, c = b & 7;
  return c * q + min(c, r) + (b >> 3);
}

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  switch (act) {
    case TDVC_ACT_RELU: return v > 0.f ? v : 0.f;
    case TDVC_ACT_LRELU: return v > 0.f ? v : v * slope;
    case TDVC_ACT_CLAMP01: return fminf(fmaxf(v, 0.f), 1.f);
    case TDVC_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}
