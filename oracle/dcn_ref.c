/* CPU ORACLE (test infrastructure only) — modulated deformable conv (DCNv2), forward sampling and
 * the three backward kernels, restated in plain C from the reference's CPU sources:
 *   bilinear sample ............ main/utils/dcnv2/src/cpu/dcn_v2_im2col_cpu.cpp:27-54
 *   gradient weight ............ :56-80      coordinate weight ... :82-123
 *   im2col (sampling * mask) ... :125-194    col2im (dInput) ..... :196-251
 *   col2im_coord (dOffset,dMask) :253-323
 * and the wrapper arithmetic of src/cpu/dcn_v2_cpu.cpp:132-272 (backward) with the forward bias rule
 * of the CUDA path (src/cuda/dcn_v2_cuda.cu:69-92; the CPU forward adds bias into uninitialised memory).
 * The reference file itself needs <TH/TH.h>, which this torch no longer ships, so it is not compiled
 * (DESIGN.md §5).  Pinned by the reference's own tests: check_zero_offset (testcpu.py:34-69) and the
 * gradcheck settings of check_gradient_dconv (testcpu.py:71-99) — tests/test_oracle_dcn.py.
 * All tensors fp32, contiguous NCHW; one sample at a time exactly like the reference's batch loop. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float bilinear(const float* im, int W, int H, float h, float w) {
  int h_low = (int)floorf(h), w_low = (int)floorf(w);
  int h_high = h_low + 1, w_high = w_low + 1;
  float lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
  float v1 = (h_low >= 0 && w_low >= 0) ? im[h_low * W + w_low] : 0.f;
  float v2 = (h_low >= 0 && w_high <= W - 1) ? im[h_low * W + w_high] : 0.f;
  float v3 = (h_high <= H - 1 && w_low >= 0) ? im[h_high * W + w_low] : 0.f;
  float v4 = (h_high <= H - 1 && w_high <= W - 1) ? im[h_high * W + w_high] : 0.f;
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

static float grad_weight(float ah, float aw, int h, int w, int H, int W) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0.f;
  int hl = (int)floorf(ah), wl = (int)floorf(aw), hh = hl + 1, wh = wl + 1;
  float wt = 0.f;
  if (h == hl && w == wl) wt = (h + 1 - ah) * (w + 1 - aw);
  if (h == hl && w == wh) wt = (h + 1 - ah) * (aw + 1 - w);
  if (h == hh && w == wl) wt = (ah + 1 - h) * (w + 1 - aw);
  if (h == hh && w == wh) wt = (ah + 1 - h) * (aw + 1 - w);
  return wt;
}

static float coord_weight(float ah, float aw, int H, int W, const float* im, int dir) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0.f;
  int hl = (int)floorf(ah), wl = (int)floorf(aw), hh = hl + 1, wh = wl + 1;
  float wt = 0.f;
  if (dir == 0) {
    if (hl >= 0 && wl >= 0) wt += -1 * (wl + 1 - aw) * im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) wt += -1 * (aw - wl) * im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) wt += (wl + 1 - aw) * im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) wt += (aw - wl) * im[hh * W + wh];
  } else {
    if (hl >= 0 && wl >= 0) wt += -1 * (hl + 1 - ah) * im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) wt += (hl + 1 - ah) * im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) wt += -1 * (ah - hl) * im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) wt += (ah - hl) * im[hh * W + wh];
  }
  return wt;
}

typedef struct { int C, H, W, Ho, Wo, kh, kw, ph, pw, sh, sw, dh, dw, G; } geo_t;

/* columns[(c*K + t)][ho*Wo + wo] = sample(im[c], pos) * mask   (one sample) */
void dcn_ref_im2col(const float* im, const float* off, const float* msk, const geo_t* g, float* col) {
  const int K = g->kh * g->kw, P = g->Ho * g->Wo, cpg = g->C / g->G;
  for (int c = 0; c < g->C; ++c) {
    const int dg = c / cpg;
    for (int ho = 0; ho < g->Ho; ++ho)
      for (int wo = 0; wo < g->Wo; ++wo)
        for (int i = 0; i < g->kh; ++i)
          for (int j = 0; j < g->kw; ++j) {
            const int t = i * g->kw + j, p = ho * g->Wo + wo;
            const float oh = off[(dg * 2 * K + 2 * t) * P + p], ow = off[(dg * 2 * K + 2 * t + 1) * P + p];
            const float m = msk[(dg * K + t) * P + p];
            const float h = ho * g->sh - g->ph + i * g->dh + oh, w = wo * g->sw - g->pw + j * g->dw + ow;
            float v = 0.f;
            if (h > -1 && w > -1 && h < g->H && w < g->W) v = bilinear(im + (long)c * g->H * g->W, g->W, g->H, h, w);
            col[(long)(c * K + t) * P + p] = v * m;
          }
  }
}

/* grad_im[c][y][x] += dcol * mask * bilinear weight (one sample; grad_im pre-zeroed by the caller) */
void dcn_ref_col2im(const float* dcol, const float* off, const float* msk, const geo_t* g, float* grad_im) {
  const int K = g->kh * g->kw, P = g->Ho * g->Wo, cpg = g->C / g->G;
  for (int c = 0; c < g->C; ++c) {
    const int dg = c / cpg;
    for (int i = 0; i < g->kh; ++i)
      for (int j = 0; j < g->kw; ++j)
        for (int ho = 0; ho < g->Ho; ++ho)
          for (int wo = 0; wo < g->Wo; ++wo) {
            const int t = i * g->kw + j, p = ho * g->Wo + wo;
            const float oh = off[(dg * 2 * K + 2 * t) * P + p], ow = off[(dg * 2 * K + 2 * t + 1) * P + p];
            const float m = msk[(dg * K + t) * P + p];
            const float ih = ho * g->sh - g->ph + i * g->dh + oh, iw = wo * g->sw - g->pw + j * g->dw + ow;
            const float top = dcol[(long)(c * K + t) * P + p] * m;
            const int ch = (int)ih, cw = (int)iw;
            for (int dy = -2; dy <= 2; ++dy)
              for (int dx = -2; dx <= 2; ++dx)
                if (ch + dy >= 0 && ch + dy < g->H && cw + dx >= 0 && cw + dx < g->W && fabsf(ih - (ch + dy)) < 1 &&
                    fabsf(iw - (cw + dx)) < 1)
                  grad_im[((long)c * g->H + ch + dy) * g->W + cw + dx] += grad_weight(ih, iw, ch + dy, cw + dx, g->H, g->W) * top;
          }
  }
}

/* grad_offset[(dg*2K + 2t + dir)][p], grad_mask[(dg*K + t)][p]   (one sample) */
void dcn_ref_col2im_coord(const float* dcol, const float* im, const float* off, const float* msk, const geo_t* g,
                          float* grad_off, float* grad_msk) {
  const int K = g->kh * g->kw, P = g->Ho * g->Wo, cpg = g->C / g->G;
  for (int dg = 0; dg < g->G; ++dg)
    for (int t = 0; t < K; ++t)
      for (int dir = 0; dir < 2; ++dir)
        for (int ho = 0; ho < g->Ho; ++ho)
          for (int wo = 0; wo < g->Wo; ++wo) {
            const int i = t / g->kw, j = t % g->kw, p = ho * g->Wo + wo;
            const float oh = off[(dg * 2 * K + 2 * t) * P + p], ow = off[(dg * 2 * K + 2 * t + 1) * P + p];
            const float m = msk[(dg * K + t) * P + p];
            float val = 0.f, mval = 0.f;
            for (int cl = 0; cl < cpg; ++cl) {
              const int c = dg * cpg + cl;
              const float* imc = im + (long)c * g->H * g->W;
              float ih = ho * g->sh - g->ph + i * g->dh + oh, iw = wo * g->sw - g->pw + j * g->dw + ow;
              const float d = dcol[(long)(c * K + t) * P + p];
              if (ih <= -1 || iw <= -1 || ih >= g->H || iw >= g->W) ih = iw = -2;
              else mval += d * bilinear(imc, g->W, g->H, ih, iw);
              val += coord_weight(ih, iw, g->H, g->W, imc, dir) * d * m;
            }
            grad_off[(long)(dg * 2 * K + 2 * t + dir) * P + p] = val;
            if (dir == 0) grad_msk[(long)(dg * K + t) * P + p] = mval;
          }
}

/* Full forward / backward over a batch with the wrapper's GEMMs written out (row-major loops). */
void dcn_ref_forward(const float* x, const float* w, const float* b, const float* off, const float* msk, float* out,
                     int B, int Cout, const geo_t* g) {
  const int K = g->kh * g->kw, P = g->Ho * g->Wo, CK = g->C * K;
  float* col = (float*)malloc(sizeof(float) * (size_t)CK * P);
  for (int n = 0; n < B; ++n) {
    dcn_ref_im2col(x + (long)n * g->C * g->H * g->W, off + (long)n * g->G * 2 * K * P, msk + (long)n * g->G * K * P, g, col);
    for (int co = 0; co < Cout; ++co)
      for (int p = 0; p < P; ++p) {
        float s = b[co];
        for (int k = 0; k < CK; ++k) s += w[(long)co * CK + k] * col[(long)k * P + p];
        out[((long)n * Cout + co) * P + p] = s;
      }
  }
  free(col);
}

void dcn_ref_backward(const float* x, const float* w, const float* off, const float* msk, const float* gout,
                      float* gx, float* goff, float* gmsk, float* gw, float* gb, int B, int Cout, const geo_t* g) {
  const int K = g->kh * g->kw, P = g->Ho * g->Wo, CK = g->C * K;
  float* col = (float*)malloc(sizeof(float) * (size_t)CK * P);
  memset(gx, 0, sizeof(float) * (size_t)B * g->C * g->H * g->W);
  memset(gw, 0, sizeof(float) * (size_t)Cout * CK);
  memset(gb, 0, sizeof(float) * (size_t)Cout);
  for (int n = 0; n < B; ++n) {
    const float* xn = x + (long)n * g->C * g->H * g->W;
    const float* on = off + (long)n * g->G * 2 * K * P;
    const float* mn = msk + (long)n * g->G * K * P;
    const float* gn = gout + (long)n * Cout * P;
    for (int k = 0; k < CK; ++k)                       /* columns = W^T @ dY */
      for (int p = 0; p < P; ++p) {
        float s = 0.f;
        for (int co = 0; co < Cout; ++co) s += w[(long)co * CK + k] * gn[(long)co * P + p];
        col[(long)k * P + p] = s;
      }
    dcn_ref_col2im_coord(col, xn, on, mn, g, goff + (long)n * g->G * 2 * K * P, gmsk + (long)n * g->G * K * P);
    dcn_ref_col2im(col, on, mn, g, gx + (long)n * g->C * g->H * g->W);
    dcn_ref_im2col(xn, on, mn, g, col);
    for (int co = 0; co < Cout; ++co) {                /* dW += dY @ columns^T ; db += dY @ 1 */
      for (int k = 0; k < CK; ++k) {
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += gn[(long)co * P + p] * col[(long)k * P + p];
        gw[(long)co * CK + k] += s;
      }
      float s = 0.f;
      for (int p = 0; p < P; ++p) s += gn[(long)co * P + p];
      gb[co] += s;
    }
  }
  free(col);
}
