// Conv backward building blocks (training path, SURVEY §8a "Training multiplies a2-a15 by ~3").
//
//   * tdvc_pack_conv_weights_indexed: fp32 master weights (the nn.Parameter, on the device) -> the fp16 MFMA
//     fragment order of the conv kernels, through three small index tables (row / channel / tap).  The same
//     kernel produces the forward packing (with PixelShuffle row permutation, concatenation channel
//     permutation, the space-to-depth form of stride-2 convs) and the DATA-GRADIENT packing (rows and
//     channels swapped, taps mirrored), so dgrad runs on the forward conv kernels: dX = conv(dY, W^T flipped).
//   * tdvc_conv_wgrad: dW[co][ci][dy][dx] += sum_{n,oy,ox} dY[n,oy,ox,co] * X[n, oy*s+dy-pad, ox*s+dx-pad, ci].
//     A contraction over PIXELS, while both tensors are stored channel-innermost: the K index of the MFMA
//     operands is the slow index in memory.  Tiles are staged pixel-major in LDS exactly like the forward
//     tiles and read with gfx950's transposing LDS read (ds_read_b64_tr_b16: a 4 x 16 block delivered
//     column-major), two reads per 32x16 operand fragment.  One workgroup owns a (64 co x 32 ci x <= 12 taps)
//     block of dW in registers and walks a strided set of 8x32-pixel blocks; partial results go to a
//     workspace and are reduced in a fixed order (bitwise reproducible, no float atomics).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ packing
__device__ __forceinline__ void pack_indexed_item(const float* __restrict__ w, const int* __restrict__ row_off, const int* __restrict__ chan_off,
                                                  const int* __restrict__ tap_off, const unsigned char* __restrict__ row_mask,
                                                  const unsigned char* __restrict__ chan_mask, const unsigned char* __restrict__ tap_mask,
                                                  int cout, int cin, int ntaps, int ck,
                                                  int nchunks, int steps, long e, half_t* __restrict__ out) {
  const int lane = (int)(e & 63);
  long q = e >> 6;
  const int s = (int)(q % steps); q /= steps;
  const int ch = (int)(q % nchunks);
  const int t = (int)(q / nchunks);
  const int ck8 = ck >> 3;
  const int r = lane & 31, h = lane >> 5;
  const int co = t * 32 + r, kc = 2 * s + h;
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (half_t)0.f;
  if (kc < ntaps * ck8 && co < cout) {
    const int tap = kc / ck8, c8 = kc - tap * ck8;
    const int ro = row_off[co], to = tap_off[tap];               // tap offsets may be negative (relative forms)
    const unsigned tm = tap_mask[tap];
    if (ro >= 0 && (tm & row_mask[co]) == 0u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ci = ch * ck + c8 * 8 + j;
        if (ci < cin) {
          const int cof = chan_off[ci];
          if (cof >= 0 && (tm & chan_mask[ci]) == 0u) o[j] = (half_t)w[(long)ro + cof + to];
        }
      }
    }
  }
  *reinterpret_cast<half8*>(out + e * 8) = o;
}

__global__ void pack_indexed_kernel(const float* __restrict__ w, const int* __restrict__ row_off, const int* __restrict__ chan_off,
                                    const int* __restrict__ tap_off, const unsigned char* __restrict__ row_mask,
                                    const unsigned char* __restrict__ chan_mask, const unsigned char* __restrict__ tap_mask,
                                    int cout, int cin, int ntaps, int ck,
                                    int nchunks, int steps, long total, half_t* __restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per 8 packed halves
  if (e >= total) return;
  pack_indexed_item(w, row_off, chan_off, tap_off, row_mask, chan_mask, tap_mask, cout, cin, ntaps, ck, nchunks, steps, e, out);
}

// Every packed layer of a model in one launch: block_start[j] .. block_start[j+1] are the 256-item blocks of job j.
__global__ void pack_batch_kernel(const tdvc_pack_job* __restrict__ jobs, const int* __restrict__ block_start, int njobs) {
  int lo = 0, hi = njobs;                                              // last job with block_start <= blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (block_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const tdvc_pack_job jb = jobs[lo];
  const int lb = blockIdx.x - block_start[lo];
  const int ck8 = jb.ck >> 3, nchunks = (jb.cin + jb.ck - 1) / jb.ck, steps = (jb.ntaps * ck8 + 1) / 2;
  const int cot = jb.cout <= 32 ? 1 : 2 * ((jb.cout + 63) / 64);      // 32-row tiles of the padded cout
  const long total = (long)cot * nchunks * steps * 64;
  const long e = (long)lb * 256 + threadIdx.x;
  if (e < total)
    pack_indexed_item(jb.w, jb.row_off, jb.chan_off, jb.tap_off, jb.row_mask, jb.chan_mask, jb.tap_mask, jb.cout, jb.cin, jb.ntaps, jb.ck, nchunks,
                      steps, e, reinterpret_cast<half_t*>(jb.dst));
  if (lb == 0 && jb.bias_src)
    for (int i = threadIdx.x; i < jb.cout; i += 256) jb.bias_dst[i] = jb.bias_src[jb.bias_perm ? jb.bias_perm[i] : i];
}

// ------------------------------------------------------------------------------------------------ wgrad
constexpr int WG_TH = 8, WG_TW = 32;            // output pixels per block
constexpr int WG_CO = 64, WG_CI = 32;           // dW block per workgroup
constexpr int WG_MAXT = 12;                     // taps per workgroup (3 per wave)
constexpr int PSG = WG_CO * 2 + 16;             // LDS bytes per pixel of the dY tile (144)
constexpr int PSX = WG_CI * 2 + 16;             // ... of the X tile (80)

struct WgradParams {
  FMap g, x;
  int Ho, Wo, stride, pad;
  int ntaps_all;                                // taps of the layer; a workgroup takes one group of <= WG_MAXT of them
  int8_t tap_dy[TDVC_MAX_TAPS], tap_dx[TDVC_MAX_TAPS];
  int co_tiles, ci_tiles;
  int blocks_x, blocks_y, nblocks;              // pixel blocks per image, total over the batch
  int tih, tiw;                                 // X tile extent in pixels
  int square_x;                                 // contract with x^2 (GDN norm pool: dgamma = sum dnorm * x^2)
  float* work;                                  // [P][ntaps_all][co_tiles*64][ci_tiles*32]
  float* bwork;                                 // [P][co_tiles*64] column sums of dY (bias gradient), or null
  long long* stamps;                            // diagnostic: per (workgroup, wave) cycles per phase, or null
  int nworkers, ngroups;                        // launch shape (the grid is one-dimensional and XCD-aware)
};

typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));

__device__ __forceinline__ half4 tr_read(const unsigned char* p) {
  const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p));
  return __builtin_bit_cast(half4, v);
}

// XL = 16-byte X-tile pieces per thread (tile pixels x 4 / 256, rounded up): 6 for 3x3 stride 1, 9 for 7x7, 18 for stride 2
template <int XL>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* gt = smem;                                         // dY tile: WG_TH*WG_TW pixels x PSG
  unsigned char* xt = smem + WG_TH * WG_TW * PSG;                   // X tile: tih*tiw pixels x PSX
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware one-dimensional grid: workgroup b runs on XCD b % 8; all (dW tile, tap group) workgroups of a worker walk
  // the same pixel blocks, so they are placed on the same XCD and share the dY / X tiles in one L2.  The j-th workgroup
  // of XCD c is (worker c + 8 (j / T), item j % T), T = tiles x tap groups; the grid is padded to 8 ceil(workers / 8) T.
  const int T = p.co_tiles * p.ci_tiles * p.ngroups;
  const int jx = (int)blockIdx.x >> 3;
  const int worker = ((int)blockIdx.x & 7) + 8 * (jx / T), nworkers = p.nworkers;
  if (worker >= nworkers) return;                                   // padding workgroups (whole workgroup, before any barrier)
  const int item = jx % T, tile = item % (p.co_tiles * p.ci_tiles), zgroup = item / (p.co_tiles * p.ci_tiles);
  const int cit = tile % p.ci_tiles, cot = tile / p.ci_tiles;
  const int tap0 = zgroup * WG_MAXT, ntaps_g = min(WG_MAXT, p.ntaps_all - tap0);         // this workgroup's tap group
  const int co0 = cot * WG_CO, ci0 = cit * WG_CI;

  // taps of this wave: wave, wave + 4, wave + 8 (inside the launch's group)
  f32x16 acc[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][m][i] = 0.f;

  // transposing-read lane geometry: 16-lane group gq, lane i = 4q + pp supplies (row q, columns 4pp .. 4pp+3)
  const int gq = lane >> 4, li = lane & 15, rq = li >> 2, cp = li & 3;
  const int khalf = gq >> 1;                                        // k = 8*khalf + {0..7}
  const int colbase = 16 * (gq & 1) + 4 * cp;                       // column (channel) this lane ADDRESSES

  // Tiles travel global -> registers -> LDS; the NEXT block's loads are issued before this block's contraction and
  // land under it (the first version loaded synchronously and spent most of a block waiting on HBM latency).
  constexpr int GL = WG_TH * WG_TW * (WG_CO / 8) / 256;             // 8 dY pieces per thread
  half8 gr[GL], xr[XL];
  const int x_items = p.tih * p.tiw * (WG_CI / 8);
  // per-thread piece geometry is block-invariant: dY piece j = pixel (row j, column tid >> 3), channels 8 * (tid & 7);
  // X piece j = tile pixel (tid >> 2) + 64 j -> (row, column) computed once
  // Loads are unconditional (clamped addresses) and the out-of-range pieces are zeroed when they are written to LDS:
  // a load under a branch made the compiler wait for it at the join (s_waitcnt vmcnt(0) after every pair of loads).
  const int g_ox = tid >> 3;
  const bool g_cok = co0 + (tid & 7) * 8 < p.g.C;
  const int g_c = g_cok ? co0 + (tid & 7) * 8 : 0;
  unsigned gmask = 0, xmask = 0;                                    // validity bits of the pieces in flight
  int x_yy[XL], x_xx[XL];
#pragma unroll
  for (int j = 0; j < XL; ++j) {
    const int px = (tid >> 2) + 64 * j;
    x_yy[j] = px / p.tiw;
    x_xx[j] = px - x_yy[j] * p.tiw;
    if (tid + j * 256 >= x_items) x_yy[j] = 1 << 20;                // beyond the tile: never in bounds
  }
  const bool x_cok = ci0 + (tid & 3) * 8 < p.x.C;
  const int x_c = x_cok ? ci0 + (tid & 3) * 8 : 0;
  const half_t *gbase = nullptr, *xbase = nullptr;                  // next block: image base + first tile pixel
  int n_oy0 = 0, n_ox0 = 0, n_iy0 = 0, n_ix0 = 0;
  auto next_block = [&](int blk) {
    const int n = blk / (p.blocks_x * p.blocks_y);
    const int rem = blk - n * (p.blocks_x * p.blocks_y);
    const int by = rem / p.blocks_x, bx = rem - by * p.blocks_x;
    n_oy0 = by * WG_TH; n_ox0 = bx * WG_TW;
    n_iy0 = n_oy0 * p.stride - p.pad; n_ix0 = n_ox0 * p.stride - p.pad;
    gbase = reinterpret_cast<const half_t*>(p.g.p) + (long)n * p.g.sn + g_c;
    xbase = reinterpret_cast<const half_t*>(p.x.p) + (long)n * p.x.sn + x_c;
  };
  auto load_g = [&](int j) {
    const int oy = n_oy0 + j, ox = n_ox0 + g_ox;
    const bool ok = oy < p.Ho && ox < p.Wo && g_cok;
    gr[j] = *reinterpret_cast<const half8*>(gbase + (min(oy, p.Ho - 1) * p.Wo + min(ox, p.Wo - 1)) * p.g.sp);
    gmask |= (ok ? 1u : 0u) << j;
  };
  auto load_x = [&](int j) {
    const int iy = n_iy0 + x_yy[j], ix = n_ix0 + x_xx[j];
    const bool ok = iy >= 0 && iy < p.x.H && ix >= 0 && ix < p.x.W && x_cok;
    xr[j] = *reinterpret_cast<const half8*>(xbase + (min(max(iy, 0), p.x.H - 1) * p.x.W + min(max(ix, 0), p.x.W - 1)) * p.x.sp);
    xmask |= (ok ? 1u : 0u) << j;
  };
  auto prefetch = [&](int blk) {
    next_block(blk);
    gmask = 0; xmask = 0;
#pragma unroll
    for (int j = 0; j < GL; ++j) load_g(j);
#pragma unroll
    for (int j = 0; j < XL; ++j) load_x(j);
  };

  // this wave's taps: wave, wave + 4, wave + 8 of the group; an absent tap reads tap 0's window and skips its MFMAs
  int xoff[3];
  bool tv[3];
#pragma unroll
  for (int ai = 0; ai < 3; ++ai) {
    const int t = wave + 4 * ai;
    tv[ai] = t < ntaps_g;
    const int tt = tap0 + (tv[ai] ? t : 0);
    xoff[ai] = (p.tap_dy[tt] * p.tiw + p.tap_dx[tt]) * PSX;
  }
  const unsigned char* abase = gt + (8 * khalf + rq) * PSG + colbase * 2;
  const unsigned char* bbase = xt + (8 * khalf + rq) * p.stride * PSX + colbase * 2;
  // bias gradient rides along: the workgroups of input-channel tile 0 / tap group 0 also sum their dY tiles per channel
  const bool do_bias = p.bwork != nullptr && cit == 0 && zgroup == 0;
  float bs[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bs[q] = 0.f;
  if (worker < p.nblocks) prefetch(worker);
  long long ph[6] = {0, 0, 0, 0, 0, 0}, tc = p.stamps ? clock64() : 0;
  auto mark = [&](int k) { if (p.stamps) { const long long now = clock64(); ph[k] += now - tc; tc = now; } };
  for (int blk = worker; blk < p.nblocks; blk += nworkers) {
    __syncthreads();                                                // previous block's reads are done
    mark(0);
    half8 zero8;
#pragma unroll
    for (int q = 0; q < 8; ++q) zero8[q] = (half_t)0.f;
#pragma unroll
    for (int j = 0; j < GL; ++j) {
      const int it = tid + j * 256;
      gr[j] = ((gmask >> j) & 1u) ? gr[j] : zero8;
      *reinterpret_cast<half8*>(gt + (it >> 3) * PSG + (it & 7) * 16) = gr[j];
    }
    if (do_bias) {                                                  // this thread's pieces all carry channels 8*(tid&7) ..
#pragma unroll
      for (int j = 0; j < GL; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) bs[q] += (float)gr[j][q];
    }
#pragma unroll
    for (int j = 0; j < XL; ++j) {
      const int it = tid + j * 256;
      half8 v = ((xmask >> j) & 1u) ? xr[j] : zero8;
      if (p.square_x) v = v * v;
      if (it < x_items) *reinterpret_cast<half8*>(xt + (it >> 2) * PSX + (it & 3) * 16) = v;
    }
    gmask = 0; xmask = 0;
    mark(1);
    __syncthreads();
    mark(2);
    const bool more = blk + nworkers < p.nblocks;                   // wave-uniform
    if (more) next_block(blk + nworkers);
    mark(3);
    // ---- contraction: k-steps of 16 consecutive output pixels of one row.  Fragments of step ks+1 are read while the
    // MFMAs of step ks run (with one workgroup per CU nothing else hides the LDS latency), and the next block's global
    // loads are issued piece by piece between the k-steps: their address arithmetic fills the MFMA shadow instead of
    // running as a 4 k-cycle prologue in front of it.
    {
      constexpr int KS = WG_TH * (WG_TW / 16);
      static_assert(KS == 2 * GL, "one dY piece per pair of k-steps");
      auto load = [&](int ks, half8 (&a)[2], half8 (&b)[3]) {
        const int yy = ks / (WG_TW / 16), kh = ks % (WG_TW / 16);
        const unsigned char* ap = abase + (yy * WG_TW + kh * 16) * PSG;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const half4 lo = tr_read(ap + m * 64), hi = tr_read(ap + m * 64 + 4 * PSG);
          a[m] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        const unsigned char* bp = bbase + ((yy * p.stride) * p.tiw + kh * 16 * p.stride) * PSX;
#pragma unroll
        for (int ai = 0; ai < 3; ++ai) {
          const half4 lo = tr_read(bp + xoff[ai]), hi = tr_read(bp + xoff[ai] + 4 * p.stride * PSX);
          b[ai] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      };
      auto fma = [&](const half8 (&a)[2], const half8 (&b)[3]) {
#pragma unroll
        for (int ai = 0; ai < 3; ++ai)
          if (tv[ai]) {                                              // wave-uniform
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[ai][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[ai], acc[ai][m], 0, 0, 0);
          }
      };
      half8 a0[2], b0[3], a1[2], b1[3];
      load(0, a0, b0);
#pragma unroll
      for (int ks = 0; ks < KS; ks += 2) {
        load(ks + 1, a1, b1);
        fma(a0, b0);
        if (more && ks < KS / 2) {                                   // all pieces in the first half: they land before the next store
          load_g(ks);
          load_g(ks + 1);
#pragma unroll
          for (int j = ks / 2; j < XL; j += GL / 2) load_x(j);
        }
        if (ks + 2 < KS) load(ks + 2, a0, b0);
        fma(a1, b1);
      }
    }
    mark(4);
  }
  if (do_bias) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                    // [32 pixel lanes][64 channels]
#pragma unroll
    for (int q = 0; q < 8; ++q) red[(tid >> 3) * 64 + (tid & 7) * 8 + q] = bs[q];
    __syncthreads();
    if (tid < 64) {
      float t = 0.f;
      for (int r = 0; r < 32; ++r) t += red[r * 64 + tid];
      p.bwork[(long)worker * p.co_tiles * WG_CO + co0 + tid] = t;
    }
  }
  // ---- partial dW block -> workspace[worker][co][ci][tap]
  const int CIW = p.ci_tiles * WG_CI, COW = p.co_tiles * WG_CO;
  float* wk = p.work + (long)worker * p.ntaps_all * COW * CIW;       // [worker][tap][co][ci]: a half wave stores 128 contiguous bytes
  const int col = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int ai = 0; ai < 3; ++ai) {
    const int t = wave + 4 * ai;
    if (t < ntaps_g) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
          wk[((long)(tap0 + t) * COW + (co0 + m * 32 + row)) * CIW + (ci0 + col)] = acc[ai][m][i];
        }
    }
  }
  if (p.stamps && lane == 0) {
    mark(5);
    long long* o = p.stamps + (((long)(zgroup * nworkers + worker) * (p.co_tiles * p.ci_tiles) + tile) * 4 + wave) * 8;
    for (int k = 0; k < 6; ++k) o[k] = ph[k];
  }
}

// dW[row_off[co] + chan_off[ci] + tap_off[t]] += scale * sum over workers (fixed order); the tables are the layer's
// forward packing tables (convpack.forward_tables), so PixelShuffle row order, concatenation channel order,
// zero-padded channels and Conv3d holders scatter to the right parameter element
__device__ __forceinline__ void wgrad_reduce_block(int blk, const float* __restrict__ work, int nworkers, int COW, int CIW, int ntaps,
                                                   const int* __restrict__ row_off, const int* __restrict__ chan_off, const int* __restrict__ tap_off,
                                                   int cout, int cin, float scale, float* __restrict__ dw, int wblocks, const float* __restrict__ bwork,
                                                   const int* __restrict__ bias_index, float* __restrict__ db) {
  // 64 elements per workgroup, 4 worker slices per element (fixed order: the sum is reproducible)
  __shared__ float part[4][64];
  const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
  if (blk >= wblocks) {                                     // bias gradient: db[index[co]] += scale * sum_w bwork[w][co]
    const int co = (blk - wblocks) * 64 + e;
    float s = 0.f;
    if (co < cout)
      for (int w = sl; w < nworkers; w += 4) s += bwork[(long)w * COW + co];
    part[sl][e] = s;
    __syncthreads();
    if (sl != 0 || co >= cout) return;
    const int d = bias_index ? bias_index[co] : co;
    if (d >= 0) db[d] += ((part[0][e] + part[1][e]) + (part[2][e] + part[3][e])) * scale;
    return;
  }
  const long i = (long)blk * 64 + e;
  const long total = (long)cout * cin * ntaps;
  const bool live = i < total;
  const int ci = live ? (int)(i % cin) : 0;               // fastest index = ci: coalesced reads of the partials
  const long q = live ? i / cin : 0;
  const int co = (int)(q % cout), t = (int)(q / cout);
  const long stride = (long)ntaps * COW * CIW;
  const long off = ((long)t * COW + co) * CIW + ci;
  float s = 0.f;
  if (live) {
#pragma unroll 8
    for (int w = sl; w < nworkers; w += 4) s += work[w * stride + off];
  }
  part[sl][e] = s;
  __syncthreads();
  if (sl != 0 || !live) return;
  const int ro = row_off[co], cf = chan_off[ci];
  if (ro < 0 || cf < 0) return;
  s = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
  dw[(long)ro + cf + tap_off[t]] += s * scale;
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ work, int nworkers, int COW, int CIW, int ntaps, const int* __restrict__ row_off,
                                    const int* __restrict__ chan_off, const int* __restrict__ tap_off, int cout, int cin, float scale,
                                    float* __restrict__ dw, int wblocks, const float* __restrict__ bwork,
                                    const int* __restrict__ bias_index, float* __restrict__ db) {
  wgrad_reduce_block((int)blockIdx.x, work, nworkers, COW, CIW, ntaps, row_off, chan_off, tap_off, cout, cin, scale, dw, wblocks, bwork, bias_index, db);
}

// The second stage of MANY layers in one launch: block_start[j] .. block_start[j + 1] are the blocks of job j (same arithmetic and
// summation order as wgrad_reduce_kernel: bit-identical gradients)
__global__ void wgrad_reduce_batch_kernel(const tdvc_wgrad_reduce_job* __restrict__ jobs, const int* __restrict__ block_start, int njobs) {
  int lo = 0, hi = njobs;                                              // last job with block_start <= blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (block_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const tdvc_wgrad_reduce_job j = jobs[lo];
  wgrad_reduce_block((int)blockIdx.x - block_start[lo], j.work, j.nworkers, j.cow, j.ciw, j.ntaps, j.row_off, j.chan_off, j.tap_off, j.cout, j.cin,
                     j.scale, j.dw, j.wblocks, j.bwork, j.bias_index, j.db);
}

}  // namespace

extern "C" int tdvc_pack_conv_weights_indexed(const float* w, const int32_t* row_off, const int32_t* chan_off, const int32_t* tap_off,
                                              const uint8_t* row_mask, const uint8_t* chan_mask, const uint8_t* tap_mask,
                                              int cout, int cin, int ntaps, int ck, void* dst, void* stream) {
  TDVC_CHECK(w && row_off && chan_off && tap_off && row_mask && chan_mask && tap_mask && dst && aligned16(dst),
             "tdvc_pack_conv_weights_indexed: null / unaligned pointer");
  const int64_t bytes = tdvc_conv_packed_bytes(cout, cin, ntaps, ck);
  TDVC_CHECK(bytes > 0, "tdvc_pack_conv_weights_indexed: bad geometry");
  const int ck8 = ck / 8, nchunks = (cin + ck - 1) / ck, steps = (ntaps * ck8 + 1) / 2;
  const long total = bytes / 16;
  hipLaunchKernelGGL(pack_indexed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     w, row_off, chan_off, tap_off, row_mask, chan_mask, tap_mask, cout, cin, ntaps, ck, nchunks, steps, total, reinterpret_cast<half_t*>(dst));
  return tdvc_launch_status("tdvc_pack_conv_weights_indexed");
}

extern "C" int64_t tdvc_pack_job_blocks(int cout, int cin, int ntaps, int ck) {
  const int64_t bytes = tdvc_conv_packed_bytes(cout, cin, ntaps, ck);
  return bytes > 0 ? (bytes / 16 + 255) / 256 : bytes;
}

extern "C" int tdvc_pack_conv_weights_batch(const tdvc_pack_job* jobs, const int32_t* block_start, int njobs, int total_blocks, void* stream) {
  TDVC_CHECK(jobs && block_start && njobs >= 1 && total_blocks >= 1, "tdvc_pack_conv_weights_batch: bad arguments");
  hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), jobs, block_start, njobs);
  return tdvc_launch_status("tdvc_pack_conv_weights_batch");
}

static long long* g_wg_stamp = nullptr;
static long g_wg_stamp_cap = 0;
extern "C" void tdvc_debug_set_stamp_buffer_wgrad(void* buf, int cap_workgroups) { g_wg_stamp = (long long*)buf; g_wg_stamp_cap = cap_workgroups; }

static int g_wg_max_workers = 256;
static long g_wg_partial_cap = 16L << 20;
extern "C" void tdvc_debug_set_wgrad_max_workers(int n) { g_wg_max_workers = n < 1 ? 1 : n; }
extern "C" void tdvc_debug_set_wgrad_partial_cap_mb(int mb) { g_wg_partial_cap = (long)(mb < 1 ? 1 : mb) << 20; }

static int wgrad_workers(int co_tiles, int ci_tiles, int groups, int nblocks, int ntaps) {
  int w = 1024 / (co_tiles * ci_tiles * groups);      // ~4 workgroups per CU over the whole launch
  const long dw_bytes = (long)co_tiles * WG_CO * ci_tiles * WG_CI * ntaps * 4;
  const long cap = g_wg_partial_cap / dw_bytes;            // partial sums are written and re-read once: keep them <= 16 MB per layer
  if (w > cap) w = (int)cap;
  if (w > nblocks) w = nblocks;
  if (w > g_wg_max_workers) w = g_wg_max_workers;
  if (w < 1) w = 1;
  return w;
}

extern "C" int64_t tdvc_conv_wgrad_work_floats(int cout, int cin, int ntaps, int N, int Ho, int Wo) {
  if (cout <= 0 || cin <= 0 || ntaps <= 0 || ntaps > TDVC_MAX_TAPS || N <= 0 || Ho <= 0 || Wo <= 0) return TDVC_EINVAL;
  const int co_tiles = (cout + WG_CO - 1) / WG_CO, ci_tiles = (cin + WG_CI - 1) / WG_CI, groups = (ntaps + WG_MAXT - 1) / WG_MAXT;
  const int nblocks = N * ((Ho + WG_TH - 1) / WG_TH) * ((Wo + WG_TW - 1) / WG_TW);
  return (int64_t)wgrad_workers(co_tiles, ci_tiles, groups, nblocks, ntaps) * co_tiles * WG_CO * (ci_tiles * WG_CI * ntaps + 1);   // + bias partials
}

extern "C" int tdvc_conv_wgrad_partials(const tdvc_fmap* g, const tdvc_fmap* x, int cout, int kh, int kw, int stride, int pad,
                                        int ntaps, const int8_t* tap_dy, const int8_t* tap_dx, const int32_t* row_off, const int32_t* chan_off,
                                        const int32_t* tap_off, int square_x, float scale, float* dw, const int32_t* bias_index, float* db,
                                        float* work, int64_t work_floats, tdvc_wgrad_reduce_job* job, void* stream) {
  TDVC_CHECK(job, "tdvc_conv_wgrad_partials: null job");
  TDVC_CHECK(g && x && dw && work && tap_dy && tap_dx && row_off && chan_off && tap_off, "tdvc_conv_wgrad: null pointer");
  TDVC_CHECK(fmap_ok16(*g) && fmap_ok16(*x) && g->N == x->N, "tdvc_conv_wgrad: fmaps must be fp16 with matching batch");
  TDVC_CHECK(stride == 1 || stride == 2, "tdvc_conv_wgrad: stride %d", stride);
  TDVC_CHECK(ntaps >= 1 && ntaps <= TDVC_MAX_TAPS && kh >= 1 && kh <= 7 && kw >= 1 && kw <= 7, "tdvc_conv_wgrad: bad window");
  TDVC_CHECK(cout >= 1 && cout <= g->C, "tdvc_conv_wgrad: cout %d > dY channels %d", cout, g->C);
  const int cin = x->C;
  const int Ho = (x->H + 2 * pad - kh) / stride + 1, Wo = (x->W + 2 * pad - kw) / stride + 1;
  TDVC_CHECK(Ho == g->H && Wo == g->W, "tdvc_conv_wgrad: dY is %dx%d, conv output is %dx%d", g->H, g->W, Ho, Wo);
  TDVC_CHECK((long)x->H * x->W * x->sp < 2147483647L && (long)g->H * g->W * g->sp < 2147483647L, "tdvc_conv_wgrad: image too large for 32-bit element offsets");
  TDVC_CHECK(work_floats >= tdvc_conv_wgrad_work_floats(cout, cin, ntaps, x->N, Ho, Wo), "tdvc_conv_wgrad: workspace too small");
  WgradParams p;
  p.g = to_dev(*g); p.x = to_dev(*x);
  p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.pad = pad;
  p.ntaps_all = ntaps;
  for (int t = 0; t < ntaps; ++t) {
    TDVC_CHECK(tap_dy[t] >= 0 && tap_dy[t] < kh && tap_dx[t] >= 0 && tap_dx[t] < kw, "tdvc_conv_wgrad: tap %d outside the window", t);
    p.tap_dy[t] = tap_dy[t]; p.tap_dx[t] = tap_dx[t];
  }
  p.co_tiles = (cout + WG_CO - 1) / WG_CO; p.ci_tiles = (cin + WG_CI - 1) / WG_CI;
  p.blocks_x = (Wo + WG_TW - 1) / WG_TW; p.blocks_y = (Ho + WG_TH - 1) / WG_TH;
  p.nblocks = x->N * p.blocks_x * p.blocks_y;
  p.tih = (WG_TH - 1) * stride + kh; p.tiw = (WG_TW - 1) * stride + kw;
  const size_t lds = (size_t)WG_TH * WG_TW * PSG + (size_t)p.tih * p.tiw * PSX;
  TDVC_CHECK(lds <= 160 * 1024, "tdvc_conv_wgrad: LDS plan %zu bytes too large", lds);
  p.work = work;
  p.square_x = square_x;
  const int groups = (ntaps + WG_MAXT - 1) / WG_MAXT;
  const int workers = wgrad_workers(p.co_tiles, p.ci_tiles, groups, p.nblocks, ntaps);
  p.nworkers = workers; p.ngroups = groups;
  p.bwork = db ? work + (long)workers * p.co_tiles * WG_CO * p.ci_tiles * WG_CI * ntaps : nullptr;
  p.stamps = (g_wg_stamp && (long)p.co_tiles * p.ci_tiles * workers * groups <= g_wg_stamp_cap) ? g_wg_stamp : nullptr;
  const int xl = (p.tih * p.tiw * (WG_CI / 8) + 255) / 256;
  TDVC_CHECK(xl <= 18, "tdvc_conv_wgrad: X tile of %dx%d pixels needs %d pieces per thread (max 18)", p.tih, p.tiw, xl);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  auto go = [&](auto kern) -> int {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) { tdvc_set_error("tdvc_conv_wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(8 * ((workers + 7) / 8) * p.co_tiles * p.ci_tiles * groups)), dim3(256), lds, st, p);
    return 0;
  };
  const int rc = xl <= 6 ? go(&conv_wgrad_kernel<6>) : (xl <= 9 ? go(&conv_wgrad_kernel<9>) : go(&conv_wgrad_kernel<18>));
  if (rc) return rc;
  const long total = (long)cout * cin * ntaps;
  const int wblocks = (int)((total + 63) / 64), bblocks = db ? (cout + 63) / 64 : 0;
  job->work = work; job->bwork = p.bwork;
  job->row_off = row_off; job->chan_off = chan_off; job->tap_off = tap_off; job->bias_index = bias_index;
  job->dw = dw; job->db = db; job->scale = scale;
  job->nworkers = workers; job->cow = p.co_tiles * WG_CO; job->ciw = p.ci_tiles * WG_CI; job->ntaps = ntaps; job->cout = cout; job->cin = cin;
  job->wblocks = wblocks; job->nblocks = wblocks + bblocks;
  return tdvc_launch_status("tdvc_conv_wgrad");
}

extern "C" int tdvc_wgrad_reduce_batch(const tdvc_wgrad_reduce_job* jobs, const int32_t* block_start, int njobs, int total_blocks, void* stream) {
  TDVC_CHECK(jobs && block_start && njobs >= 1 && total_blocks >= 1, "tdvc_wgrad_reduce_batch: bad arguments");
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), jobs, block_start, njobs);
  return tdvc_launch_status("tdvc_wgrad_reduce_batch");
}

extern "C" int tdvc_conv_wgrad_bias(const tdvc_fmap* g, const tdvc_fmap* x, int cout, int kh, int kw, int stride, int pad,
                                    int ntaps, const int8_t* tap_dy, const int8_t* tap_dx, const int32_t* row_off, const int32_t* chan_off,
                                    const int32_t* tap_off, int square_x, float scale, float* dw, const int32_t* bias_index, float* db,
                                    float* work, int64_t work_floats, void* stream) {
  tdvc_wgrad_reduce_job j;
  const int rc = tdvc_conv_wgrad_partials(g, x, cout, kh, kw, stride, pad, ntaps, tap_dy, tap_dx, row_off, chan_off, tap_off, square_x, scale, dw,
                                          bias_index, db, work, work_floats, &j, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)j.nblocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), j.work, j.nworkers, j.cow,
                     j.ciw, j.ntaps, j.row_off, j.chan_off, j.tap_off, j.cout, j.cin, j.scale, j.dw, j.wblocks, j.bwork, j.bias_index, j.db);
  return tdvc_launch_status("tdvc_conv_wgrad");
}

extern "C" int tdvc_conv_wgrad(const tdvc_fmap* g, const tdvc_fmap* x, int cout, int kh, int kw, int stride, int pad,
                               int ntaps, const int8_t* tap_dy, const int8_t* tap_dx, const int32_t* row_off, const int32_t* chan_off,
                               const int32_t* tap_off, int square_x, float scale, float* dw, float* work, int64_t work_floats, void* stream) {
  return tdvc_conv_wgrad_bias(g, x, cout, kh, kw, stride, pad, ntaps, tap_dy, tap_dx, row_off, chan_off, tap_off, square_x, scale, dw, nullptr, nullptr,
                              work, work_floats, stream);
}
