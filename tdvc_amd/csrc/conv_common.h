// Shared between the MFMA conv kernels: launch parameters and the fused epilogue.
#pragma once
#include "common.h"

namespace convk {

struct ConvParams {
  const half_t* x; long x_sn; int x_sp; int H, W, Cin;
  const half_t* w; const float* bias;
  FMap y; int Ho, Wo; int cout;
  FMap aux; FMap res; FMap res2;
  int ntaps, kh, kw, pad;
  int in_stride;        // input pixels per output pixel (conv_mfma_v5: 1x1 stride-2 skip convs; 1 elsewhere)
  int nchunks, steps;   // steps per chunk
  int square, gdn, act; float slope; int round16, out_mode;
  int tiles_x;
  int simple;           // host decision: transposed fast epilogue (conv_is_simple)
  int s2d, Corig;       // space-to-depth view of a stride-2 conv (v3 only): Corig = channels per parity
  int reverse;          // walk the tile raster backwards (alternate launches: see tdvc_conv2d, "Infinity Cache")
  int bcast_T; float bcast_slope;   // conv_mfma_v5 only: y[:, t] = lrelu(y[:, t] + conv) over bcast_T slices (tdvc_conv_desc::bcast_T)
  float* csum;                      // conv_mfma_v5 only: partial channel sums of the stored values (tdvc_conv_desc::chan_sum), or null
  int8_t tap_dy[TDVC_MAX_TAPS], tap_dx[TDVC_MAX_TAPS];
};


// Epilogue for 4 consecutive output channels co..co+3 of output pixel (oy, ox) of image n:
// bias -> GDN -> fp16 rounding (DCN quirk) -> activation -> residual(s) -> store (NHWC fp16 / fp32,
// PixelShuffle(2), planar fp32).
// F32MAPS: fp32 GDN multiplicand / fp32 second residual (only conv_f32 passes them; keeping the branches out of the fp16
// kernels saves them 16 VGPRs: conv_mfma<4,1,1> 104 -> 88)
template <bool F32MAPS = false>
__device__ __forceinline__ void epilogue4(const ConvParams& p, int n, int oy, int ox, int co, float v[4]) {
  if (p.bias) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += b4[i];
  }
  // output geometry
  int py = oy, px = ox, pc = co, PW = p.Wo;
  if (p.out_mode == TDVC_OUT_SHUFFLE2) {
    const int cq = p.cout >> 2;            // channels after the shuffle
    const int sub = co / cq;               // host permutes rows: packed = (i*2+j)*cq + c
    pc = co - sub * cq;
    py = 2 * oy + (sub >> 1);
    px = 2 * ox + (sub & 1);
    PW = 2 * p.Wo;
  }
  if (p.gdn) {
    float a[4];
    if (F32MAPS && p.aux.f32) {            // fp32 islands (conv_f32): exact 1 / sqrtf like the reference's torch.rsqrt on CPU
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.aux.p) + (long)n * p.aux.sn + ((long)oy * p.Wo + ox) * p.aux.sp + co);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = a4[i] * (p.gdn == TDVC_GDN_FWD ? 1.0f / sqrtf(v[i]) : sqrtf(v[i]));
    } else {
      const half4 a4 = *reinterpret_cast<const half4*>(reinterpret_cast<const half_t*>(p.aux.p) + (long)n * p.aux.sn + ((long)oy * p.Wo + ox) * p.aux.sp + co);
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = (float)a4[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = a[i] * (p.gdn == TDVC_GDN_FWD ? rsqrtf(v[i]) : sqrtf(v[i]));
    }
  }
  if (p.round16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)(half_t)v[i];
  }
  if (p.act) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = act_apply(v[i], p.act, p.slope);
  }
  const long opix = (long)py * PW + px;
  if (p.res.p) {
    if (p.res.f32) {
      const float* rp = reinterpret_cast<const float*>(p.res.p) + (long)n * p.res.sn + opix * p.res.sp + pc;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (pc + i < p.res.C) v[i] += rp[i];
    } else if (pc < p.res.C) {
      const half4 r4 = *reinterpret_cast<const half4*>(reinterpret_cast<const half_t*>(p.res.p) + (long)n * p.res.sn + opix * p.res.sp + pc);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += (float)r4[i];
    }
  }
  if (p.res2.p && pc < p.res2.C) {
    if (F32MAPS && p.res2.f32) {
      const float* rp = reinterpret_cast<const float*>(p.res2.p) + (long)n * p.res2.sn + opix * p.res2.sp + pc;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (pc + i < p.res2.C) v[i] += rp[i];
    } else {
      const half4 r4 = *reinterpret_cast<const half4*>(reinterpret_cast<const half_t*>(p.res2.p) + (long)n * p.res2.sn + opix * p.res2.sp + pc);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += (float)r4[i];
    }
  }
  if (p.out_mode == TDVC_OUT_NCHW_F32) {
    float* yp = reinterpret_cast<float*>(p.y.p);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (co + i < p.cout) yp[(((long)n * p.cout + co + i) * p.Ho + oy) * p.Wo + ox] = v[i];
  } else if (p.y.f32) {
    float* yp = reinterpret_cast<float*>(p.y.p) + (long)n * p.y.sn + opix * p.y.sp + pc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (pc + i < p.y.C) yp[i] = v[i];
  } else if (pc < p.y.C) {
    half4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (half_t)v[i];
    *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(p.y.p) + (long)n * p.y.sn + opix * p.y.sp + pc) = o;
  }
}

// XCD-aware tile walk of the persistent kernels (v3, v7).  Workgroups are dispatched round-robin over the 8 XCDs, each
// with its own L2, so workgroup x of a launch runs on XCD x % 8.  The raster of tiles is cut into 8 contiguous bands,
// one per XCD, and a workgroup walks its XCD's band with the stride of that XCD's workgroup count: neighbouring tiles
// (shared halo rows / columns) and consecutive tile rows are fetched through ONE L2 instead of eight.  Needs
// gridDim.x % 8 == 0 (then x % 8 is the XCD whatever blockIdx.y / z are); otherwise the plain interleaved walk.
__device__ __forceinline__ void xcd_tile_walk(int ntiles, int& first, int& stride, int& my_tiles) {
  const int gx = (int)gridDim.x, bx = (int)blockIdx.x;
  if ((gx & 7) == 0 && ntiles >= gx) {
    const int xcd = bx & 7, slot = bx >> 3, per = gx >> 3;
    const int nb = (ntiles + 7) >> 3;                      // tiles per band (the last band may be shorter)
    const int band0 = xcd * nb;
    const int band_n = min(nb, ntiles - band0);
    first = band0 + slot;
    stride = per;
    my_tiles = slot < band_n ? (band_n - slot + per - 1) / per : 0;
  } else {
    first = bx;
    stride = gx;
    my_tiles = (ntiles - first + stride - 1) / stride;
  }
}

// Which layers take the transposed epilogue below (everything else uses epilogue4).
inline bool conv_is_simple(const ConvParams& p) {
  const bool shuf = p.out_mode == TDVC_OUT_SHUFFLE2;
  if (!(p.out_mode == TDVC_OUT_NHWC || shuf) || p.y.f32 || p.round16 || !p.bias) return false;
  if (p.gdn ? p.act != TDVC_ACT_NONE : !(p.act == TDVC_ACT_NONE || p.act == TDVC_ACT_RELU || p.act == TDVC_ACT_LRELU)) return false;
  if (p.act == TDVC_ACT_LRELU && !(p.slope >= 0.f && p.slope <= 1.f)) return false;     // max(v, v*slope) form
  if (p.res.p && (p.res.f32 || p.res.C < p.y.C)) return false;
  if (p.res2.p && p.res2.C < p.y.C) return false;
  if (shuf && (((p.cout >> 2) % 64) != 0 || p.gdn)) return false;
  if (p.gdn && (p.aux.C < p.y.C || shuf)) return false;
  return true;
}
// slope that turns `max(v, v * slope)` into none / ReLU / LeakyReLU
inline float conv_simple_slope(const ConvParams& p) {
  return p.act == TDVC_ACT_NONE ? 1.f : (p.act == TDVC_ACT_RELU ? 0.f : p.slope);
}

// Transposed "simple" epilogue shared by every conv kernel (v1 .. v7): fp16 NHWC output (optionally through
// PixelShuffle(2)), bias, none / ReLU / LeakyReLU via a slope select, GDN / inverse GDN
// (aux * rsqrt(v) | aux * sqrt(v)), up to two fp16 residuals.  One pass = one output row of 32 pixels x
// 64 channels of this wave: bias + activation in the MFMA layout (4 consecutive channels per lane), fp16
// through a wave-private LDS region `ew` (32 x 144 B), then every lane owns 8 consecutive channels of a
// pixel, so aux / residuals / output move as full 128-byte lines (8 lanes x 16 B per pixel).  All loads
// of a pass are issued before any is consumed (one memory round trip per pass).
// --- the two halves of the transposed epilogue ------------------------------------------------------
// (1) pack: bias + activation in the MFMA layout -> fp16, 4 consecutive channels per (mt, g) slot.
struct PackedRow { half4 v[8]; };           // [mt*4 + g] -> channels mt*32 + 8g + 4*(lane>>5) .. +3

// BIAS_IN_ACC: the caller initialised the accumulators with the bias (no add here, no re-zeroing).
// The activation is max(v, v * slope): none (slope 1) / ReLU (0) / LeakyReLU (0 <= slope <= 1, conv_is_simple).
template <int NTX, bool BIAS_IN_ACC = false>
__device__ __forceinline__ void epilogue_pack(const ConvParams& p, f32x16 (&acc)[2][NTX], const float* bias64, int lane,
                                              PackedRow (&out)[NTX], bool zero_acc) {
  const int hh = lane >> 5;
#pragma unroll
  for (int nt = 0; nt < NTX; ++nt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cl = mt * 32 + 8 * g + 4 * hh;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!BIAS_IN_ACC) b4 = *reinterpret_cast<const f32x4*>(bias64 + cl);
        half4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = acc[mt][nt][4 * g + i];
          if constexpr (!BIAS_IN_ACC) {
            v += b4[i];
            if (zero_acc) acc[mt][nt][4 * g + i] = 0.f;
          }
          v = __builtin_fmaxf(v, v * p.slope);
          o[i] = (half_t)v;
        }
        out[nt].v[mt * 4 + g] = o;
      }
}

// (2) store one output row (32 pixels x 64 channels of this wave) through the wave-private LDS region
// `ew` (32 x 144 B): afterwards every lane owns 8 consecutive channels of a pixel, so GDN multiplicand,
// residuals and the output move as full 128-byte lines; all loads are issued before any is consumed.
// `full` (wave-uniform): the caller guarantees that the row and all 32 pixels are inside the image, so the
// four stores are always issued (conv_mfma_v7 counts them in its s_waitcnt vmcnt bookkeeping).
__device__ __forceinline__ void epilogue_store_row(const ConvParams& p, const PackedRow& row, unsigned char* ew, int n, int cbase,
                                                   int oy, int ox_first, int lane, bool full = false) {
  constexpr int EPS = 144;
  const int hh = lane >> 5, r = lane & 31;
  const int chunk = lane & 7, prow = lane >> 3;
  const int co = cbase + chunk * 8;
  const bool has1 = p.res.p != nullptr, has2 = p.res2.p != nullptr, gdn = p.gdn != 0;
  int pc = co, sub_y = 0, sub_x = 0, mul = 1, PW = p.Wo;
  if (p.out_mode == TDVC_OUT_SHUFFLE2) {     // host permutes rows: packed = (i*2+j)*cq + c
    const int cq = p.cout >> 2;
    const int sub = co / cq;
    pc = co - sub * cq;
    sub_y = sub >> 1; sub_x = sub & 1; mul = 2; PW = 2 * p.Wo;
  }
  const bool ch_ok = pc < p.y.C && co < ((p.cout + 63) & ~63);
  const bool row_ok = oy < p.Ho && ch_ok;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<half4*>(ew + r * EPS + (mt * 32 + 8 * g + 4 * hh) * 2) = row.v[mt * 4 + g];
  const int pcc = ch_ok ? pc : 0;
  constexpr int KB = 4;                         // pixel groups per load batch (all four: one round trip per row)
#pragma unroll
  for (int k0 = 0; k0 < 4; k0 += KB) {
    half8 r0[KB], r1[KB], r2[KB];
    int opix[KB], apix[KB];               // pixel indices fit 32 bits (checked by the host); widened once per access
    bool ok[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int ox = ox_first + (k0 + k) * 8 + prow;
      ok[k] = full ? ch_ok : (row_ok && ox < p.Wo);
      opix[k] = ok[k] ? (mul * oy + sub_y) * PW + (mul * ox + sub_x) : 0;
      apix[k] = ok[k] ? oy * p.Wo + ox : 0;
    }
    if (gdn) {
      const half_t* ab = reinterpret_cast<const half_t*>(p.aux.p) + (long)n * p.aux.sn + (ch_ok ? co : 0);
#pragma unroll
      for (int k = 0; k < KB; ++k) r0[k] = *reinterpret_cast<const half8*>(ab + (long)apix[k] * p.aux.sp);
    }
    if (has1) {
      const half_t* rb = reinterpret_cast<const half_t*>(p.res.p) + (long)n * p.res.sn + pcc;
#pragma unroll
      for (int k = 0; k < KB; ++k) r1[k] = *reinterpret_cast<const half8*>(rb + (long)opix[k] * p.res.sp);
    }
    if (has2) {
      const half_t* rb = reinterpret_cast<const half_t*>(p.res2.p) + (long)n * p.res2.sn + pcc;
#pragma unroll
      for (int k = 0; k < KB; ++k) r2[k] = *reinterpret_cast<const half8*>(rb + (long)opix[k] * p.res2.sp);
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      half8 h = *reinterpret_cast<const half8*>(ew + ((k0 + k) * 8 + prow) * EPS + chunk * 16);
      if (gdn) {                          // coder path: fp32 arithmetic, one rounding at the end
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)r0[k][j] * (p.gdn == TDVC_GDN_FWD ? rsqrtf((float)h[j]) : sqrtf((float)h[j]));
        if (has1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (float)r1[k][j];
        }
        if (has2) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (float)r2[k][j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (half_t)v[j];
      } else {
        // residuals as packed fp16 adds (4 instructions per 8 channels instead of ~28 through fp32), which is
        // also what the reference computes under AMP: the conv result is an fp16 tensor, `out + x` an fp16 add
        // (utils.py:52-56), and an outer skip is a second fp16 add
        if (has1) h = h + r1[k];
        if (has2) h = h + r2[k];
      }
      if (ok[k]) *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(p.y.p) + (long)n * p.y.sn + (long)opix[k] * p.y.sp + pc) = h;
    }
  }
}

// Lean form of the transposed epilogue (host: conv_is_lean -> ConvParams::simple == 2): fp16 NHWC output, no GDN, no
// PixelShuffle.  The generic form above carries every mode as run-time branches: ~700 vector instructions per two rows
// plus ~450 s_nop of hazard padding between its fp32 conversions (listing of conv_mfma_v10 before the split), serial
// time in which the wave issues no MFMA.  Here one row is: residual loads first (their latency runs under the packing),
// 32 accumulators (+ bias) -> packed fp16 (v_cvt_pk_f16_f32), activation as packed fp16 math (what the reference's
// autocast computes: the conv result is an fp16 tensor and ReLU / LeakyReLU runs on it), 8 ds_write_b64, 4
// ds_read_b128, packed fp16 residual adds, 4 full-line stores: ~90 vector instructions.
// BCAST (conv_mfma_v5, tdvc_conv_desc::bcast_T == 4): the row is not stored; it is added to the four 64-channel slices that
// start at y and LeakyReLU'd in place, with the arithmetic of bcast_add_act_kernel (fp32 add of the two fp16 values,
// v > 0 ? v : v * slope, one rounding) -- bit-identical to the conv followed by tdvc_bcast_add_act.
// CSUM (conv_mfma_v5, tdvc_conv_desc::chan_sum): the lane also adds the values it stores (channels 8 (lane & 7) .. + 8 of four pixels per row)
// into cs[0..7].
template <int NTX, bool BIAS_IN_ACC, bool BCAST = false, bool CSUM = false>
__device__ __forceinline__ void epilogue_lean_seq(const ConvParams& p, f32x16 (&acc)[2][NTX], const float* bias64, unsigned char* ew, int n,
                                                  int cbase, int oy_first, int ox_first, int lane, bool zero_acc, bool full, float* cs = nullptr) {
  constexpr int EPS = 144;
  const int hh = lane >> 5, r = lane & 31;
  const int chunk = lane & 7, prow = lane >> 3;
  const int co = cbase + chunk * 8;
  const bool ch_ok = co < p.y.C && co < ((p.cout + 63) & ~63);
  const int cc = ch_ok ? co : 0;
  const bool has1 = p.res.p != nullptr, has2 = p.res2.p != nullptr;       // wave-uniform
  const half_t sl = (half_t)p.slope;
  const half2v sl2 = {sl, sl}, zero2 = {(half_t)0.f, (half_t)0.f};
  const bool act = p.slope != 1.f, relu = p.slope == 0.f;                  // wave-uniform
  half_t* yb = reinterpret_cast<half_t*>(p.y.p) + (long)n * p.y.sn + cc;
  const half_t* rb1 = reinterpret_cast<const half_t*>(p.res.p) + (long)n * p.res.sn + cc;
  const half_t* rb2 = reinterpret_cast<const half_t*>(p.res2.p) + (long)n * p.res2.sn + cc;
  f32x4 b4[2][4];
  if constexpr (!BIAS_IN_ACC) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) b4[mt][g] = *reinterpret_cast<const f32x4*>(bias64 + mt * 32 + 8 * g + 4 * hh);
  }
#pragma unroll
  for (int nt = 0; nt < NTX; ++nt) {
    const int oy = oy_first + nt;
    int opix[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ox = ox_first + k * 8 + prow;
      ok[k] = full ? ch_ok : (ch_ok && oy < p.Ho && ox < p.Wo);
      opix[k] = ok[k] ? oy * p.Wo + ox : 0;
    }
    half8 r1[4], r2[4];
    if (has1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) r1[k] = *reinterpret_cast<const half8*>(rb1 + (long)opix[k] * p.res.sp);
    }
    if (has2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) r2[k] = *reinterpret_cast<const half8*>(rb2 + (long)opix[k] * p.res2.sp);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = acc[mt][nt][4 * g + i];
          if constexpr (!BIAS_IN_ACC) {
            v[i] += b4[mt][g][i];
            if (zero_acc) acc[mt][nt][4 * g + i] = 0.f;
          }
        }
        half2v lo = {(half_t)v[0], (half_t)v[1]}, hi = {(half_t)v[2], (half_t)v[3]};
        if (relu) {
          lo = __builtin_elementwise_max(lo, zero2);
          hi = __builtin_elementwise_max(hi, zero2);
        } else if (act) {
          lo = __builtin_elementwise_max(lo, lo * sl2);
          hi = __builtin_elementwise_max(hi, hi * sl2);
        }
        const half4 o = {lo[0], lo[1], hi[0], hi[1]};
        *reinterpret_cast<half4*>(ew + r * EPS + (mt * 32 + 8 * g + 4 * hh) * 2) = o;
      }
    if constexpr (BCAST) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const half8 v = *reinterpret_cast<const half8*>(ew + (k * 8 + prow) * EPS + chunk * 16);
        half_t* yp = yb + (long)opix[k] * p.y.sp;
        half8 xs[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) xs[t] = *reinterpret_cast<const half8*>(yp + t * 64);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          half8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float s = (float)xs[t][j] + (float)v[j];
            o[j] = (half_t)(s > 0.f ? s : s * p.bcast_slope);
          }
          if (ok[k]) *reinterpret_cast<half8*>(yp + t * 64) = o;
        }
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      half8 v = *reinterpret_cast<const half8*>(ew + (k * 8 + prow) * EPS + chunk * 16);
      if (has1) v = v + r1[k];
      if (has2) v = v + r2[k];
      if (ok[k]) *reinterpret_cast<half8*>(yb + (long)opix[k] * p.y.sp) = v;
      if constexpr (CSUM) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] += ok[k] ? (float)v[j] : 0.f;
      }
    }
  }
}

// immediate form used by conv_mfma (v1), v2, v3, v5 and v7 (v7: generic only, its counted wait is checked against ONE
// epilogue's store count)
// MODE: 1 = generic form only, 2 = lean form only (the kernel instantiation was chosen on the host), 3 = both, chosen at run
// time from ConvParams::simple (conv_mfma v1, whose many instantiations are not multiplied again)
template <int NTX, bool BIAS_IN_ACC = false, int MODE = 3>
__device__ __forceinline__ void epilogue_simple_rows(const ConvParams& p, f32x16 (&acc)[2][NTX], const float* bias64,
                                                     unsigned char* ew, int n, int cbase, int oy_first, int ox_first,
                                                     int lane, bool zero_acc, bool full = false) {
  if constexpr (MODE == 2) {
    epilogue_lean_seq<NTX, BIAS_IN_ACC>(p, acc, bias64, ew, n, cbase, oy_first, ox_first, lane, zero_acc, full);
    return;
  }
  if constexpr (MODE == 3) {
    if (p.simple == 2) {                 // wave-uniform
      epilogue_lean_seq<NTX, BIAS_IN_ACC>(p, acc, bias64, ew, n, cbase, oy_first, ox_first, lane, zero_acc, full);
      return;
    }
  }
  PackedRow rows[NTX];
  epilogue_pack<NTX, BIAS_IN_ACC>(p, acc, bias64, lane, rows, zero_acc);
#pragma unroll
  for (int nt = 0; nt < NTX; ++nt) epilogue_store_row(p, rows[nt], ew, n, cbase, oy_first + nt, ox_first, lane, full);
}

// Lean epilogue of the weight-stationary 3x3 kernel (conv_mfma_v10): what `conv_is_lean` admits -- fp16 NHWC output,
// no GDN, no PixelShuffle, bias already in the accumulators, activation max(v, v * slope), NRES fp16 residuals known at
// compile time.  The generic transposed epilogue above carries every mode as run-time branches (1.4 k vector instructions
// per 16x32 tile per wave, 0.9 k s_nop of hazard padding between its fp32 conversions); with one wave per SIMD those
// instructions are serial time in which the matrix pipe idles.  Here a 2-row round is: residual loads first (their
// latency runs under the packing), 64 accumulators -> packed fp16 (v_cvt_pk_f16_f32), activation as packed fp16 math
// (what the reference's autocast computes: the conv result is an fp16 tensor, LeakyReLU runs on it), 16 ds_write_b64 into
// two wave-private 32 x 144 B regions, 8 ds_read_b128, packed fp16 residual adds, 8 full-line stores.
inline bool conv_is_lean(const ConvParams& p) {
  return conv_is_simple(p) && p.out_mode == TDVC_OUT_NHWC && !p.gdn;
}

template <int NRES, int NR, bool STAMPED = false>
__device__ __forceinline__ void epilogue_lean_rows(const ConvParams& p, const f32x16 (&acc)[2][NR], unsigned char* ew, int n,
                                                   int cbase, int oy, int ox_first, int lane, bool full, long long* st = nullptr) {
#define EST(i) do { if constexpr (STAMPED) { if (st) st[i] = clock64(); } } while (0)
  static_assert(NR % 2 == 0, "two rows per round");
  constexpr int EPS = 144, EROW = 32 * EPS;
  const int hh = lane >> 5, r = lane & 31;
  const int chunk = lane & 7, prow = lane >> 3;
  const int co = cbase + chunk * 8;
  const bool ch_ok = co < p.y.C && co < ((p.cout + 63) & ~63);
  const int cc = ch_ok ? co : 0;
  int opix[NR][4];
  bool ok[NR][4];
#pragma unroll
  for (int j = 0; j < NR; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ox = ox_first + k * 8 + prow;
      ok[j][k] = full ? ch_ok : (ch_ok && oy + j < p.Ho && ox < p.Wo);
      opix[j][k] = ok[j][k] ? (oy + j) * p.Wo + ox : 0;
    }
  // every residual load of the tile goes out before the first store: the compiler's waits on them are then counted
  // (vmcnt(k), k = younger stores) instead of draining the previous round's stores
  half8 r1[NR][4], r2[NR][4];
  if constexpr (NRES >= 1) {
    const half_t* rb = reinterpret_cast<const half_t*>(p.res.p) + (long)n * p.res.sn + cc;
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) r1[j][k] = *reinterpret_cast<const half8*>(rb + (long)opix[j][k] * p.res.sp);
  }
  if constexpr (NRES >= 2) {
    const half_t* rb = reinterpret_cast<const half_t*>(p.res2.p) + (long)n * p.res2.sn + cc;
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) r2[j][k] = *reinterpret_cast<const half8*>(rb + (long)opix[j][k] * p.res2.sp);
  }
  const half_t sl = (half_t)p.slope;
  const half2v sl2 = {sl, sl};
  const bool act = p.slope != 1.f;                       // wave-uniform: no activation -> no packed math at all
  half_t* yb = reinterpret_cast<half_t*>(p.y.p) + (long)n * p.y.sn + cc;
  EST(0);
#pragma unroll
  for (int j0 = 0; j0 < NR; j0 += 2) {
#pragma unroll
    for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          half2v lo = {(half_t)acc[mt][j][4 * g + 0], (half_t)acc[mt][j][4 * g + 1]};
          half2v hi = {(half_t)acc[mt][j][4 * g + 2], (half_t)acc[mt][j][4 * g + 3]};
          if (act) {
            lo = __builtin_elementwise_max(lo, lo * sl2);
            hi = __builtin_elementwise_max(hi, hi * sl2);
          }
          half4 o = {lo[0], lo[1], hi[0], hi[1]};
          *reinterpret_cast<half4*>(ew + (j & 1) * EROW + r * EPS + (mt * 32 + 8 * g + 4 * hh) * 2) = o;
        }
    EST(1 + 3 * (j0 / 2));                 // packed + written to LDS (issue)
    half8 h[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) h[j][k] = *reinterpret_cast<const half8*>(ew + j * EROW + (k * 8 + prow) * EPS + chunk * 16);
    EST(2 + 3 * (j0 / 2));                 // transposed values back (clock64 drains lgkmcnt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        half8 v = h[j][k];
        if constexpr (NRES >= 1) v = v + r1[j0 + j][k];
        if constexpr (NRES >= 2) v = v + r2[j0 + j][k];
        if (ok[j0 + j][k]) *reinterpret_cast<half8*>(yb + (long)opix[j0 + j][k] * p.y.sp) = v;
      }
    EST(3 + 3 * (j0 / 2));                 // stores issued
  }
#undef EST
}

}  // namespace convk
