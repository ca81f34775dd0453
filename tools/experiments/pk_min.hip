// Minimal probes for the packed-FP32 finding (tools/race_warp_bwd.py): how little does it take?
//   0: c = a + b on float2 (one v_pk_add_f32 between a 64-bit load pair and a 64-bit store)
//   1: c = a * b + a on float2 (v_pk_fma_f32 / v_pk_mul + v_pk_add)
//   2: variant 0 preceded by a float division per lane (v_div_scale / v_div_fmas / v_div_fixup write VCC), like the warp kernel
//   3: variant 2 with scalar adds (control)
//   4: variant 0 behind a chain of twelve dependent gathers from a 64 MB table (the wave lives for many microseconds)
//   5: variant 4 with scalar adds (control)
//   7 / 8: variant 4 with the packed op written with the operand modifiers the warp kernel's chain uses (neg_lo / neg_hi = a
//          packed subtraction; op_sel / op_sel_hi = one component broadcast), as inline assembly
//   9 / 10 / 11: ONE packed op with op_sel / op_sel_hi only;  12: variant 8 with 16 idle cycles between its two packed ops
//   6: variant 4 behind twelve exec-masked (conditional) gathers and a division, the warp kernel's shape
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ void pk_min(const f2* a, const f2* b, f2* c, long n, float d, const float* table, unsigned tmask) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f2 x = a[i], y = b[i];
  if (V == 2 || V == 3) {
    y[0] = y[0] / (d + x[1] * x[1]);       // data-dependent divisor: a full division sequence
    y[1] = y[1] / (d + x[0] * x[0]);
  }
  if (V >= 4) {
    unsigned idx = (unsigned)i * 2654435761u;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const bool take = V != 6 || ((idx >> 7) & 3) != 0;              // variant 6: a quarter of the lanes skip each load
      const float t = take ? table[idx & tmask] : 0.f;
      acc += t;
      idx = idx * 1664525u + 1013904223u + (unsigned)(int)(t * 8.f);   // the next address depends on the loaded value
    }
    if (V == 6) acc = acc / (d + x[0] * x[0]);
    y[0] += acc * 1e-3f;
    y[1] -= acc * 1e-3f;
  }
  f2 r;
  if (V == 1) r = x * y + x;
  else if (V == 7) {
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  } else if (V == 8) {
    f2 t;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "v"(y));
    asm volatile("s_nop 1\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(t), "v"(x));
  } else if (V == 9) {
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));                       // y.x broadcast
  } else if (V == 10) {
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(x), "v"(y));          // x.xx + y.yy
  } else if (V == 11) {
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));          // (x.x + y.y, x.y + y.x)
  } else if (V == 12) {
    f2 t;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "v"(y));
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(t), "v"(x));
  } else if (V == 5) {
    float r0 = x[0] + y[0];
    asm volatile("" : "+v"(r0));
    float r1 = x[1] + y[1];
    r = f2{r0, r1};
  }
  else if (V == 3) {
    float r0 = x[0] + y[0];
    asm volatile("" : "+v"(r0));
    float r1 = x[1] + y[1];
    r = f2{r0, r1};
  } else {
    asm volatile("" : "+v"(x), "+v"(y));
    r = x + y;
  }
  c[i] = r;
}

extern "C" int pk_min_launch(int variant, const void* a, const void* b, void* c, long n, const float* table, unsigned tmask, void* stream) {
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const f2 *pa = reinterpret_cast<const f2*>(a), *pb = reinterpret_cast<const f2*>(b);
  f2* pc = reinterpret_cast<f2*>(c);
  switch (variant) {
    case 0: hipLaunchKernelGGL(pk_min<0>, grid, block, 0, st, pa, pb, pc, n, 1.5f, table, tmask); break;
    case 1: hipLaunchKernelGGL(pk_min<1>, grid, block, 0, st, pa, pb, pc, n, 1.5f, table, tmask); break;
    case 2: hipLaunchKernelGGL(pk_min<2>, grid, block, 0, st, pa, pb, pc, n, 1.5f, table, tmask); break;
    case 3: hipLaunchKernelGGL(pk_min<3>, grid, block, 0, st, pa, pb, pc, n, 1.5f, table, tmask); break;
    case 4: hipLaunchKernelGGL(pk_min<4>, grid, block, 0, st, pa, pb, pc, n, 1.5f, table, tmask); break;
    case 5: hipLaunchKernelGGL(pk_min<5>, grid, block, 0, st, pa, pb, pc, n, 1.5f, table, tmask); break;
    case 6: hipLaunchKernelGGL(pk_min<6>, grid, block, 0, st, pa, pb, pc, n, 1.5f, table, tmask); break;
    case 7: hipLaunchKernelGGL(pk_min<7>, grid, block, 0, st, pa, pb, pc, n, 1.5f, table, tmask); break;
    case 8: hipLaunchKernelGGL(pk_min<8>, grid, block, 0, st, pa, pb, pc, n, 1.5f, table, tmask); break;
    case 9: hipLaunchKernelGGL(pk_min<9>, grid, block, 0, st, pa, pb, pc, n, 1.5f, table, tmask); break;
    case 10: hipLaunchKernelGGL(pk_min<10>, grid, block, 0, st, pa, pb, pc, n, 1.5f, table, tmask); break;
    case 11: hipLaunchKernelGGL(pk_min<11>, grid, block, 0, st, pa, pb, pc, n, 1.5f, table, tmask); break;
    default: hipLaunchKernelGGL(pk_min<12>, grid, block, 0, st, pa, pb, pc, n, 1.5f, table, tmask); break;
  }
  return (int)hipGetLastError();
}
