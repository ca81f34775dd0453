// conv_mfma_v5 — barrier-free, weight-stationary 1x1 conv (stride 1, Cin <= ~1000, Cout >= 64): the
// multi-frame fusion conv (256 -> 64), the GDN / inverse-GDN norm pools (128 -> 128 with the squared
// -input prologue), OffsetGen's 1x1 fusions.
//
// A 1x1 conv has no halo and no reuse between pixels: under the workgroup-tiled kernels a stage is one
// tap = 8 MFMAs per wave between two barriers, i.e. barrier-bound (the shared-tile weight-stationary kernel
// this replaced: 491 us for 256 -> 64 at 1080p = 2.7 TB/s algorithmic).  Here every wave owns a PRIVATE 2x32-pixel tile in LDS (5 KB) next to the
// resident weights, so after the one-time weight load there is NO barrier: waves drift apart and one
// wave's publish / epilogue / stores overlap its SIMD partner's MFMAs.  Measured 296 us = 4.5 TB/s
// algorithmic (0.57 of HBM peak) for the same layer.
// (The same scheme was tried for 3x3 windows — each wave loading its own 4-row halo — and measured 20 %
// SLOWER than a shared tile, so v5 only takes 1x1; 3x3 is conv_mfma_v7.)
#include <type_traits>

#include "conv_common.h"

using convk::ConvParams;

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WR = 2, TW5 = 32, CK5 = 32, PS5 = 80, NTHR5 = 512, NWAVE = 8;
constexpr int WSL5 = 4096;

struct V5Extra {
  int nbtiles;        // workgroup tiles (16 x 32 output pixels = 8 wave tiles)
  int wtile_bytes;    // private LDS bytes per wave
};

// TL = 16-byte loads per lane per stage: 4 for 1x1 (2 x 32 pixels x 4 chunks / 64 lanes)
template <int SIMPLE, int TL>      // SIMPLE 0: per-register epilogue4, 1: transposed generic, 2: transposed lean
__global__ __launch_bounds__(NTHR5, 1) void conv_mfma_v5_kernel(const ConvParams p, const V5Extra e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* tapoff = reinterpret_cast<int*>(smem);
  float* bias_s = reinterpret_cast<float*>(smem + 256);
  unsigned char* wlds = smem + 512;                                   // nchunks x ntaps x 4 KB, shared, read-only
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* tbuf = wlds + p.nchunks * p.ntaps * WSL5 + wave * e.wtile_bytes;   // this wave's private tile

  const int hh = lane >> 5, r = lane & 31;
  const int cb = blockIdx.y, n = blockIdx.z;
  const int TIW = TW5 + p.kw - 1, TIH = WR + p.kh - 1;
  const int total_items = TIH * TIW * 4;
  const int ntaps = p.ntaps, nchunks = p.nchunks;

  const int first = blockIdx.x, stride = gridDim.x;
  const int my_tiles = (e.nbtiles - first + stride - 1) / stride;
  const int nstages = my_tiles * nchunks;
  if (nstages <= 0) return;

  if (tid < 64) bias_s[tid] = p.bias ? p.bias[blockIdx.y * 64 + tid] : 0.f;
  if (tid < ntaps) tapoff[tid] = (p.tap_dy[tid] * TIW + p.tap_dx[tid]) * PS5;
  {
    const int nslices = nchunks * ntaps;
    for (int i = tid; i < nslices * 256; i += NTHR5) {
      const int sl = i >> 8, u = i & 255;            // u = q*64 + lane, q = mt*2 + s2
      const int qq = u >> 6, ln = u & 63;
      const half_t* src = p.w + ((((long)(cb * 2 + (qq >> 1)) * nchunks * ntaps + sl) * 2 + (qq & 1)) * 64 + ln) * 8;
      *reinterpret_cast<half8*>(wlds + sl * WSL5 + u * 16) = *reinterpret_cast<const half8*>(src);
    }
  }
  __syncthreads();          // the only barrier: weights, bias and tap offsets are visible

  // ---- per-lane constant staging geometry (items of the wave's private tile) ----------------------
  int it_rr[TL], it_c[TL], it_dst[TL];
#pragma unroll
  for (int j = 0; j < TL; ++j) {
    const int idx = j * 64 + lane;
    const int c8 = idx & 3, pix = idx >> 2;
    it_rr[j] = pix / TIW;
    it_c[j] = pix - it_rr[j] * TIW;
    it_dst[j] = idx < total_items ? pix * PS5 + c8 * 16 : -1;
  }
  const int c8off = (lane & 3) * 8;
  const half_t* xn = p.x + (long)n * p.x_sn;

  u32x4 treg[2][TL];
  unsigned okmask[2] = {0u, 0u};
  auto issue = [&](auto setc, int S) {
    constexpr int SET = decltype(setc)::value;
    const int tile_i = S / nchunks, ch = S - tile_i * nchunks;
    const int tile = p.reverse ? e.nbtiles - 1 - (first + tile_i * stride) : first + tile_i * stride;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const int iy0 = ty * (WR * NWAVE) + wave * WR - p.pad, ix0 = tx * TW5 - p.pad;
    const int cg = ch * CK5 + c8off;
    const bool cok = cg < p.Cin;
    const int cgc = cok ? cg : 0;
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < TL; ++j) {
      const int iy = (iy0 + it_rr[j]) * p.in_stride, ix = (ix0 + it_c[j]) * p.in_stride;    // in_stride > 1 only for 1x1, pad 0
      const bool ok = cok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
      const half_t* src = xn + ((long)iyc * p.W + ixc) * p.x_sp + cgc;
      treg[SET][j] = *reinterpret_cast<const u32x4*>(src);
      m |= (ok ? 1u : 0u) << j;
    }
    okmask[SET] = m;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  int base[WR];
#pragma unroll
  for (int nt = 0; nt < WR; ++nt) base[nt] = (nt * TIW + r) * PS5 + hh * 16;

  f32x16 acc[2][WR];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < WR; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // tdvc_conv_desc::chan_sum: this lane's channels 8 (lane & 7) .. + 8 over the pixels it stored
  issue(I0{}, 0);
  if (nstages > 1) issue(I1{}, 1);

  auto stage = [&](auto setc, int S) {
    constexpr int SET = decltype(setc)::value;
    const int tile_i = S / nchunks, ch = S - tile_i * nchunks;
    // publish: wave-private tile, in-order LDS within the wave -> no barrier
#pragma unroll
    for (int j = 0; j < TL; ++j) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      u32x4 val = ((okmask[SET] >> j) & 1u) ? treg[SET][j] : z;
      if (p.square) {
        half8 hv = __builtin_bit_cast(half8, val);
        hv = hv * hv;
        val = __builtin_bit_cast(u32x4, hv);
      }
      if (it_dst[j] >= 0) *reinterpret_cast<u32x4*>(tbuf + it_dst[j]) = val;
    }

    const unsigned char* wch = wlds + ch * ntaps * WSL5 + lane * 16;
    for (int t = 0; t < ntaps; ++t) {
      const unsigned char* wslot = wch + t * WSL5;
      const int toff = tapoff[t];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        half8 a[2], b[WR];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const half8*>(wslot + (mt * 2 + s2) * 1024);
#pragma unroll
        for (int nt = 0; nt < WR; ++nt) b[nt] = *reinterpret_cast<const half8*>(tbuf + base[nt] + toff + s2 * 32);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < WR; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
      }
    }

    if (ch == nchunks - 1) {
      const int tile = p.reverse ? e.nbtiles - 1 - (first + tile_i * stride) : first + tile_i * stride;
      const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
      const int oy0 = ty * (WR * NWAVE) + wave * WR;
      if constexpr (SIMPLE == 4) {
        // lean + broadcast-add over the four slices at y (tdvc_conv_desc::bcast_T): the wave overwrites exactly the pixels whose
        // input channels it has finished reading (1x1: no halo; the prefetched stages belong to other pixels)
        convk::epilogue_lean_seq<WR, false, true>(p, acc, bias_s, tbuf, n, cb * 64, oy0, tx * TW5, lane, true, false);
      } else if constexpr (SIMPLE) {
        // the private tile doubles as the transpose scratch (this wave's reads of it are complete:
        // every fragment read was waited for before its MFMA)
        if (SIMPLE == 2 && p.csum) convk::epilogue_lean_seq<WR, false, false, true>(p, acc, bias_s, tbuf, n, cb * 64, oy0, tx * TW5, lane, true, false, cs);
        else convk::epilogue_simple_rows<WR, false, SIMPLE>(p, acc, bias_s, tbuf, n, cb * 64, oy0, tx * TW5, lane, true);
      } else {
        const int ox = tx * TW5 + r;
#pragma unroll
        for (int nt = 0; nt < WR; ++nt) {
          const int oy = oy0 + nt;
          const bool ok = oy < p.Ho && ox < p.Wo;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float v[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) { v[i] = acc[mt][nt][4 * g + i]; acc[mt][nt][4 * g + i] = 0.f; }
              if (ok) convk::epilogue4(p, n, oy, ox, (cb * 2 + mt) * 32 + 8 * g + 4 * hh, v);
            }
          }
        }
      }
    }
    if (S + 2 < nstages) issue(setc, S + 2);          // last vector-memory work of the stage
  };
  for (int S = 0; S < nstages; S += 2) {
    stage(I0{}, S);
    if (S + 1 < nstages) stage(I1{}, S + 1);
  }
  if (SIMPLE == 2 && p.csum) {
    // the eight lanes with the same lane & 7 hold the same channels: fixed-order butterfly, then one row per (workgroup, wave)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cs[j] += __shfl_xor(cs[j], 8);
      cs[j] += __shfl_xor(cs[j], 16);
      cs[j] += __shfl_xor(cs[j], 32);
    }
    if (lane < 8) {
      float* o = p.csum + (((long)n * gridDim.x + blockIdx.x) * NWAVE + wave) * 64 + lane * 8;
      *reinterpret_cast<f32x4*>(o) = f32x4{cs[0], cs[1], cs[2], cs[3]};
      *reinterpret_cast<f32x4*>(o + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
    }
  }
}

inline int v5_wtile_bytes(int kh, int kw) {
  const int b = (WR + kh - 1) * (TW5 + kw - 1) * PS5;
  return b < 32 * 144 ? 32 * 144 : b;               // also the epilogue's transpose scratch
}
inline int v5_lds_bytes(int kh, int kw, int ntaps, int nchunks) { return 512 + nchunks * ntaps * WSL5 + NWAVE * v5_wtile_bytes(kh, kw); }
inline int v5_tl(int kh, int kw) { return ((WR + kh - 1) * (TW5 + kw - 1) * 4 + 63) / 64; }

template <int TL>
int launch_tl(const ConvParams& q, const V5Extra& e, int mode, dim3 grid, int lds, hipStream_t st) {
  static TdvcPerDeviceFlag attr_flags;
  bool& attr_done = attr_flags.flag();
  if (!attr_done) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v5_kernel<1, TL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err == hipSuccess)
      err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v5_kernel<2, TL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err == hipSuccess)
      err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v5_kernel<0, TL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err == hipSuccess)
      err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v5_kernel<4, TL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) { tdvc_set_error("conv v5: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    attr_done = true;
  }
  if (mode == 4) hipLaunchKernelGGL((conv_mfma_v5_kernel<4, TL>), grid, dim3(NTHR5), lds, st, q, e);
  else if (mode == 2) hipLaunchKernelGGL((conv_mfma_v5_kernel<2, TL>), grid, dim3(NTHR5), lds, st, q, e);
  else if (mode == 1) hipLaunchKernelGGL((conv_mfma_v5_kernel<1, TL>), grid, dim3(NTHR5), lds, st, q, e);
  else hipLaunchKernelGGL((conv_mfma_v5_kernel<0, TL>), grid, dim3(NTHR5), lds, st, q, e);
  return tdvc_launch_status("tdvc_conv2d(v5)");
}

}  // namespace

bool conv_v5_eligible(const tdvc_conv_desc* d, int Ho, int Wo) {
  static const bool off = getenv("TDVC_CONV_NO_V5") != nullptr || getenv("TDVC_CONV_V1") != nullptr;
  if (off) return false;
  const int nchunks = (d->x.C + CK5 - 1) / CK5;
  const int tl = v5_tl(d->kh, d->kw);
  const bool stride_ok = d->stride == 1 || (d->stride == 2 && d->kh == 1 && d->kw == 1 && d->pad == 0);   // ResidualBlockWithStride skips
  return d->ck == 32 && stride_ok && d->ntaps >= 1 && d->ntaps <= 9 && d->kh <= 3 && d->kw <= 3 && d->cout >= 64 &&
         d->x.C >= 32 && !d->s2d && ((long)Ho * Wo >= 8192 || d->stride == 2) && tl == 4 &&     // stride 2: the only ck = 32 kernel that takes it

         v5_lds_bytes(d->kh, d->kw, d->ntaps, nchunks) <= 160 * 1024;
}

// workgroups along x for a launch over `nbtiles` tiles (also the row count of tdvc_conv_desc::chan_sum: 8 waves per workgroup)
static int v5_grid_x(int nbtiles, int cout_blocks, int N) {
  int gx = 256 / (cout_blocks * N);
  if (gx < 1) gx = 1;
  return gx > nbtiles ? nbtiles : gx;
}
int conv_v5_chan_sum_rows(int Ho, int Wo, int cout_blocks, int N) {
  const int nbtiles = ((Wo + TW5 - 1) / TW5) * ((Ho + WR * NWAVE - 1) / (WR * NWAVE));
  return v5_grid_x(nbtiles, cout_blocks, N) * NWAVE;
}

int launch_conv_v5(const ConvParams& p, int cout_blocks, int N, hipStream_t st) {
  ConvParams q = p;
  q.tiles_x = (p.Wo + TW5 - 1) / TW5;
  const int tiles_y = (p.Ho + WR * NWAVE - 1) / (WR * NWAVE);
  V5Extra e;
  e.nbtiles = q.tiles_x * tiles_y;
  e.wtile_bytes = v5_wtile_bytes(p.kh, p.kw);
  const int lds = v5_lds_bytes(p.kh, p.kw, p.ntaps, p.nchunks);
  const bool simple = convk::conv_is_simple(p);
  if (simple) q.slope = convk::conv_simple_slope(p);
  dim3 grid(v5_grid_x(e.nbtiles, cout_blocks, N), cout_blocks, N);
  const int tl = v5_tl(p.kh, p.kw);
  if (tl == 4) return launch_tl<4>(q, e, p.bcast_T ? 4 : (simple ? (convk::conv_is_lean(p) ? 2 : 1) : 0), grid, lds, st);
  tdvc_set_error("conv v5: unsupported window %dx%d", p.kh, p.kw);
  return TDVC_EINVAL;
}
