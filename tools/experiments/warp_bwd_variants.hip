// Experiment (round 3): spynet_warp_backward_kernel gives run-to-run different results while another stream runs
// weight-gradient launches (tools/race_warp_bwd.py).  Variants of the same arithmetic to find what the difference hangs on.
//   0: the kernel's expressions on raw pointers (no FMap descriptors, no runtime dtype branch)
//   1: as 0, result written to a second buffer (no read-modify-write of dflow_up)
//   2: as 0, corner loads unconditional at clamped addresses
//   3: as 0, divisions replaced by multiplications with host-computed reciprocals
//   4: as 0, the result added with one float atomic per element (no load / store pair in the kernel)
//   5: as 0, dflow_up read first (before every other load)
//   6: as 2 (no exec-masked loads), 64-thread workgroups
//   7: as 0 with idle cycles (s_nop) between the last vector-ALU instruction and the store that reads its result
//   9 / 10: the last add written as an explicit 2-vector add (v_pk_add_f32), with (9) / without (10) s_nop before the store
//   11: packed add on operands fetched with two 32-bit loads;  12: one 64-bit load, two scalar adds
//   13: as 10 with idle cycles between the wait for the load and the packed add that consumes it
//   8: as 0 with the two results kept out of packed-FP32 instructions (opaque to the vectoriser), no idle cycles
#include <hip/hip_runtime.h>

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));

template <int V>
__global__ void warp_bwd(const float* supp, const float* flow_up, const half_t* dcat8, float* dflow_up, float* out2, int N, int H, int W,
                         float rwm, float rhm) {
  const long npix = (long)H * W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * N) return;
  const int n = (int)(i / npix);
  const long pix = i % npix;
  const int y = (int)(pix / W), x = (int)(pix % W);
  float* df = dflow_up + ((long)n * npix + pix) * 2;
  float old0 = 0.f, old1 = 0.f;
  if (V == 5) { old0 = df[0]; old1 = df[1]; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  const float* fu = flow_up + ((long)n * npix + pix) * 2;
  const float fx = fu[0], fy = fu[1];
  const float gx = (float)x + fx, gy = (float)y + fy;
  const float wm = (float)(W - 1 > 1 ? W - 1 : 1), hm = (float)(H - 1 > 1 ? H - 1 : 1);
  float nx, ny, mx, my;
  if (V == 3) { nx = 2.0f * gx * rwm - 1.0f; ny = 2.0f * gy * rhm - 1.0f; }
  else { nx = 2.0f * gx / wm - 1.0f; ny = 2.0f * gy / hm - 1.0f; }
  float ix = ((nx + 1.f) / 2.f) * (float)(W - 1), iy = ((ny + 1.f) / 2.f) * (float)(H - 1);
  if (V == 3) { mx = (ix >= 0.f && ix <= (float)(W - 1)) ? (float)(W - 1) * rwm : 0.f; my = (iy >= 0.f && iy <= (float)(H - 1)) ? (float)(H - 1) * rhm : 0.f; }
  else { mx = (ix >= 0.f && ix <= (float)(W - 1)) ? (float)(W - 1) / wm : 0.f; my = (iy >= 0.f && iy <= (float)(H - 1)) ? (float)(H - 1) / hm : 0.f; }
  ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
  iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
  const int ix0 = (int)floorf(ix), iy0 = (int)floorf(iy), ix1 = ix0 + 1, iy1 = iy0 + 1;
  const bool x0ok = ix0 >= 0 && ix0 < W, x1ok = ix1 >= 0 && ix1 < W, y0ok = iy0 >= 0 && iy0 < H, y1ok = iy1 >= 0 && iy1 < H;
  const float* sp = supp + (long)n * npix * 4;
  const half8 h = *reinterpret_cast<const half8*>(dcat8 + ((long)n * npix + pix) * 8);
  float dc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dc[j] = (float)h[j];
  float gix = 0.f, giy = 0.f;
  const int cx0 = min(max(ix0, 0), W - 1), cx1 = min(max(ix1, 0), W - 1), cy0 = min(max(iy0, 0), H - 1), cy1 = min(max(iy1, 0), H - 1);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float nw, ne, sw, se;
    if (V == 2 || V == 6) {
      nw = sp[((long)cy0 * W + cx0) * 4 + c] * (float)(y0ok && x0ok);
      ne = sp[((long)cy0 * W + cx1) * 4 + c] * (float)(y0ok && x1ok);
      sw = sp[((long)cy1 * W + cx0) * 4 + c] * (float)(y1ok && x0ok);
      se = sp[((long)cy1 * W + cx1) * 4 + c] * (float)(y1ok && x1ok);
    } else {
      nw = (y0ok && x0ok) ? sp[((long)iy0 * W + ix0) * 4 + c] : 0.f;
      ne = (y0ok && x1ok) ? sp[((long)iy0 * W + ix1) * 4 + c] : 0.f;
      sw = (y1ok && x0ok) ? sp[((long)iy1 * W + ix0) * 4 + c] : 0.f;
      se = (y1ok && x1ok) ? sp[((long)iy1 * W + ix1) * 4 + c] : 0.f;
    }
    const float g = dc[3 + c];
    gix += g * ((ne - nw) * ((float)iy1 - iy) + (se - sw) * (iy - (float)iy0));
    giy += g * ((sw - nw) * ((float)ix1 - ix) + (se - ne) * (ix - (float)ix0));
  }
  float* dst = V == 1 ? out2 + ((long)n * npix + pix) * 2 : df;
  if (V == 4) {
    atomicAdd(df, dc[6] + mx * gix);
    atomicAdd(df + 1, dc[7] + my * giy);
  } else if (V == 5) {
    dst[0] = old0 + dc[6] + mx * gix;
    dst[1] = old1 + dc[7] + my * giy;
  } else if (V == 7) {
    float r0 = df[0] + dc[6] + mx * gix, r1 = df[1] + dc[7] + my * giy;
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(r0), "+v"(r1));
    dst[0] = r0;
    dst[1] = r1;
  } else if (V == 9 || V == 10 || V == 13) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a = {dc[6] + mx * gix, dc[7] + my * giy};
    f2 o = *reinterpret_cast<const f2*>(df);
    if (V == 13) asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(a), "+v"(o));
    else asm volatile("" : "+v"(a), "+v"(o));
    f2 r = o + a;
    if (V == 9) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(r));
    *reinterpret_cast<f2*>(dst) = r;
  } else if (V == 11) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a = {dc[6] + mx * gix, dc[7] + my * giy};
    float o0 = __builtin_nontemporal_load(df);
    asm volatile("" : "+v"(o0));
    float o1 = __builtin_nontemporal_load(df + 1);
    f2 o = {o0, o1};
    asm volatile("" : "+v"(a), "+v"(o));
    f2 r = o + a;
    *reinterpret_cast<f2*>(dst) = r;
  } else if (V == 12) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    float a0 = dc[6] + mx * gix, a1 = dc[7] + my * giy;
    f2 o = *reinterpret_cast<const f2*>(df);
    asm volatile("" : "+v"(o));
    float r0 = o[0] + a0;
    asm volatile("" : "+v"(r0));
    float r1 = o[1] + a1;
    asm volatile("" : "+v"(r1));
    dst[0] = r0;
    dst[1] = r1;
  } else if (V == 8) {
    float a0 = dc[6] + mx * gix, a1 = dc[7] + my * giy;
    asm volatile("" : "+v"(a0));
    float r0 = df[0] + a0;
    asm volatile("" : "+v"(r0));
    float r1 = df[1] + a1;
    dst[0] = r0;
    dst[1] = r1;
  } else {
    dst[0] = df[0] + dc[6] + mx * gix;
    dst[1] = df[1] + dc[7] + my * giy;
  }
}

extern "C" int warp_bwd_variant(int variant, const float* supp, const float* flow_up, const void* dcat8, float* dflow_up, float* out2, int N, int H,
                                int W, void* stream) {
  const long total = (long)N * H * W;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float rwm = 1.f / (float)(W - 1 > 1 ? W - 1 : 1), rhm = 1.f / (float)(H - 1 > 1 ? H - 1 : 1);
  const half_t* dc = reinterpret_cast<const half_t*>(dcat8);
  switch (variant) {
    case 0: hipLaunchKernelGGL(warp_bwd<0>, grid, block, 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
    case 1: hipLaunchKernelGGL(warp_bwd<1>, grid, block, 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
    case 2: hipLaunchKernelGGL(warp_bwd<2>, grid, block, 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
    case 3: hipLaunchKernelGGL(warp_bwd<3>, grid, block, 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
    case 4: hipLaunchKernelGGL(warp_bwd<4>, grid, block, 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
    case 9: hipLaunchKernelGGL(warp_bwd<9>, grid, block, 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
    case 10: hipLaunchKernelGGL(warp_bwd<10>, grid, block, 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
    case 13: hipLaunchKernelGGL(warp_bwd<13>, grid, block, 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
    case 11: hipLaunchKernelGGL(warp_bwd<11>, grid, block, 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
    case 12: hipLaunchKernelGGL(warp_bwd<12>, grid, block, 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
    case 7: hipLaunchKernelGGL(warp_bwd<7>, grid, block, 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
    case 8: hipLaunchKernelGGL(warp_bwd<8>, grid, block, 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
    case 5: hipLaunchKernelGGL(warp_bwd<5>, grid, block, 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
    default: hipLaunchKernelGGL(warp_bwd<6>, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, st, supp, flow_up, dc, dflow_up, out2, N, H, W, rwm, rhm); break;
  }
  return (int)hipGetLastError();
}
