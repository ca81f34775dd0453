// conv_mfma_v2 — implicit-GEMM conv for the dominant layer class of the TDVC path (stride 1, >= 2 taps,
// Cin >= 32, Cout >= 64: the 3x3 64/128-channel transforms, the 7x7 SPyNet middle layers, the masked
// 5x5 context conv).
//
// Why a second kernel: PMC on v1 showed 67 % of wave cycles waiting and ~3 GB of L2
// requests per 3x3 64->64 launch — every wave re-fetched the layer's weights from L2 at every k-step.
// Here
//   * a workgroup (4 waves) owns a 16x32-pixel tile x 64 output channels; a wave owns 64 channels x
//     128 pixels (2x4 MFMA 32x32 tiles): 8 MFMAs per 6 LDS reads;
//   * weights stream L2 -> registers -> a 2-slot LDS ring, one 4 KB (tap, 32-channel chunk) slice per
//     iteration, issued two taps ahead and shared by the 4 waves (one 16-byte load per thread per
//     slice): 8x less L2 weight traffic than v1; one barrier per tap;
//   * the halo tile of a 32-channel chunk is staged with all of a thread's loads in flight at once
//     (vmcnt returns in order, so a deeper cross-stage prefetch would collide with the just-in-time
//     weight stream); two workgroups per CU overlap one's staging with the other's matrix work;
//   * LDS: (16+kh-1)(32+kw-1) x 80 B tile (pixel stride 80 B = 5 x 16 B: ds_read_b128 conflict-free)
//     + 8 KB weight ring = 57 KB for 3x3.
#include "conv_common.h"

using convk::ConvParams;

namespace {

constexpr int TH2 = 16, TW2 = 32, NT = 4, CK = 32, CK8 = 4, PS = 80;
constexpr int WSLICE = 4096;   // bytes of one (cout-block, chunk, tap) weight slice: 2 mt x 2 k-steps x 1 KiB

// diagnostic only (never launched unless tdvc_debug_set_stamp_buffer() was called): per-phase
// s_memtime stamps of wave 0 of every workgroup, written to a buffer no kernel code reads.
static long long* g_stamp_buf = nullptr;
static int g_stamp_cap = 0;

template <int SIMPLE, bool STAMP = false>      // 0: per-register epilogue4, 1: transposed generic, 2: transposed lean
__global__ __launch_bounds__(256, 2) void conv_mfma_v2_kernel(const ConvParams p, long long* stamps = nullptr, int stamp_cap = 0) {
  long long st[8];
  int nst = 0;
#define TDVC_STAMP() do { if constexpr (STAMP) { if (nst < 8) st[nst++] = clock64(); } } while (0)
  TDVC_STAMP();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* tapoff = reinterpret_cast<int*>(smem);
  unsigned char* wring = smem + 256;
  unsigned char* tbuf = smem + 256 + 2 * WSLICE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, r = lane & 31;
  const int tile_id = p.reverse ? (int)gridDim.x - 1 - tdvc_xcd_tile(blockIdx.x) : tdvc_xcd_tile(blockIdx.x);
  const int tx = tile_id % p.tiles_x, ty = tile_id / p.tiles_x;
  const int cb = blockIdx.y, n = blockIdx.z;
  const int TIW = TW2 + p.kw - 1;
  const int total_items = (TH2 + p.kh - 1) * TIW * CK8;
  const int ntaps = p.ntaps, nchunks = p.nchunks;
  const int total_taps = ntaps * nchunks;

  if (tid < ntaps) tapoff[tid] = (p.tap_dy[tid] * TIW + p.tap_dx[tid]) * PS;

  const half_t* xn = p.x + (long)n * p.x_sn;
  const int iy0 = ty * TH2 - p.pad, ix0 = tx * TW2 - p.pad;
  // weight slice source of this thread: q = tid>>6 -> (mt = q>>1, k-step s2 = q&1); flat tap index T
  // -> (chunk = T / ntaps, tap = T % ntaps) -> packed step (chunk*ntaps + tap)*2 + s2 of tile cb*2+mt
  const int q = tid >> 6;
  const half_t* wthread = p.w + ((long)(cb * 2 + (q >> 1)) * nchunks * p.steps * 64 + lane) * 8 + (long)(q & 1) * 512;
  auto wsrc = [&](int T) -> const half_t* {
    const int Tc = T < total_taps ? T : total_taps - 1;      // beyond the end: harmless valid address
    return wthread + (long)Tc * 1024;
  };

  *reinterpret_cast<half8*>(wring + tid * 16) = *reinterpret_cast<const half8*>(wsrc(0));
  half8 wreg = *reinterpret_cast<const half8*>(wsrc(1));

  int base[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) base[nt] = ((wave * NT + nt) * TIW + r) * PS + hh * 16;

  f32x16 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  int T = 0;
  for (int ch = 0; ch < nchunks; ++ch) {
    // ---- stage the halo tile of channels [ch*32, ch*32+32): every load of a thread in flight at once
    for (int idx = tid; idx < total_items; idx += 256) {
      const int c8 = idx & 3, pix = idx >> 2;
      const int c = pix % TIW, rr = pix / TIW;
      const int iy = iy0 + rr, ix = ix0 + c;
      const int cg = ch * CK + c8 * 8;
      half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && cg < p.Cin)
        v = *reinterpret_cast<const half8*>(xn + ((long)iy * p.W + ix) * p.x_sp + cg);
      *reinterpret_cast<half8*>(tbuf + pix * PS + c8 * 16) = v;
    }
    __syncthreads();
    TDVC_STAMP();
    for (int t = 0; t < ntaps; ++t, ++T) {
      // (a) publish the slice loaded one iteration ago; (b) issue the slice needed two taps ahead
      *reinterpret_cast<half8*>(wring + ((T + 1) & 1) * WSLICE + tid * 16) = wreg;
      wreg = *reinterpret_cast<const half8*>(wsrc(T + 2));
      // (c) matrix work of tap t
      const unsigned char* wslot = wring + (T & 1) * WSLICE + lane * 16;
      const int toff = tapoff[t];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        half8 a[2], b[NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const half8*>(wslot + (mt * 2 + s2) * 1024);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const half8*>(tbuf + base[nt] + toff + s2 * 32);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
      }
      // (d) everyone is done with tap t's slot (and, after the last tap, with the tile)
      __syncthreads();
    }
    TDVC_STAMP();
  }

  if constexpr (SIMPLE) {
    // Transposed epilogue: bias + activation in the MFMA layout (4 consecutive channels per lane), fp16
    // through a wave-private LDS region (the tile buffer is free after the last barrier), then every
    // lane owns 8 consecutive channels of a pixel: residuals and the output move as full 128-byte lines
    // (8 lanes x 16 B per pixel) instead of 32 partial-line requests per store instruction.
    convk::epilogue_simple_rows<NT, false, SIMPLE>(p, acc, p.bias + cb * 64, tbuf + wave * (32 * 144), n, cb * 64,
                                     ty * TH2 + wave * NT, tx * TW2, lane, false);
  } else {
    const int ox = tx * TW2 + r;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int oy = ty * TH2 + wave * NT + nt;
      if (oy >= p.Ho || ox >= p.Wo) continue;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[mt][nt][4 * g + i];
          convk::epilogue4(p, n, oy, ox, (cb * 2 + mt) * 32 + 8 * g + 4 * hh, v);
        }
      }
    }
  }
  if constexpr (STAMP) {
    __builtin_amdgcn_s_waitcnt(0);
    TDVC_STAMP();
    const int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (threadIdx.x == 0 && bid < stamp_cap)
      for (int i = 0; i < 8; ++i) stamps[(long)bid * 8 + i] = i < nst ? st[i] : 0;
  }
#undef TDVC_STAMP
}

inline int v2_lds_bytes(int kh, int kw) { return 256 + 2 * WSLICE + (TH2 + kh - 1) * (TW2 + kw - 1) * PS; }

}  // namespace

extern "C" void tdvc_debug_set_stamp_buffer(void* buf, int cap_blocks) { g_stamp_buf = (long long*)buf; g_stamp_cap = cap_blocks; }

bool conv_v2_eligible(const tdvc_conv_desc* d, int Ho, int Wo) {
  static const bool force_v1 = getenv("TDVC_CONV_V1") != nullptr;
  if (force_v1) return false;
  return d->ck == 32 && d->stride == 1 && d->ntaps >= 2 && d->cout >= 64 && d->x.C >= 32 && !d->square_input &&
         v2_lds_bytes(d->kh, d->kw) <= 80 * 1024 && (long)Ho * Wo >= 2048;
}

int launch_conv_v2(const ConvParams& p, int ntiles_unused, int cout_blocks, int N, hipStream_t st) {
  ConvParams q = p;
  q.tiles_x = (p.Wo + TW2 - 1) / TW2;
  const int tiles_y = (p.Ho + TH2 - 1) / TH2;
  const int lds = v2_lds_bytes(p.kh, p.kw);
  // compact epilogue when the layer is the common case: fp16 NHWC out, bias, none/ReLU/LeakyReLU, fp16 residuals
  const bool simple = convk::conv_is_simple(p);
  if (simple) q.slope = convk::conv_simple_slope(p);
  dim3 grid(q.tiles_x * tiles_y, cout_blocks, N);
  hipError_t err = hipSuccess;
  static TdvcPerDeviceFlag attr_flags;
  bool& attr_done = attr_flags.flag();
  if (!attr_done) {
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (err == hipSuccess)
      err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (err == hipSuccess)
      err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (err != hipSuccess) { tdvc_set_error("conv v2: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    attr_done = true;
  }
  const bool lean = simple && convk::conv_is_lean(p);
  if (g_stamp_buf && simple) {
    static bool a2 = false;
    if (!a2) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v2_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); a2 = true; }
    hipLaunchKernelGGL((conv_mfma_v2_kernel<1, true>), grid, dim3(256), lds, st, q, g_stamp_buf, g_stamp_cap);
  } else if (lean) hipLaunchKernelGGL((conv_mfma_v2_kernel<2>), grid, dim3(256), lds, st, q, (long long*)nullptr, 0);
  else if (simple) hipLaunchKernelGGL((conv_mfma_v2_kernel<1>), grid, dim3(256), lds, st, q, (long long*)nullptr, 0);
  else hipLaunchKernelGGL((conv_mfma_v2_kernel<0>), grid, dim3(256), lds, st, q, (long long*)nullptr, 0);
  return tdvc_launch_status("tdvc_conv2d(v2)");
}
