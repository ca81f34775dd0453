// Host-side range coder of the entropy stage: rANS with a 64-bit state and 32-bit renormalisation
// words, 16-bit CDF precision and 4-bit "bypass" digits for out-of-table symbols — the published
// scheme of CompressAI's `ans` extension (BufferedRansEncoder / RansDecoder over ryg_rans' rans64),
// which `Cheng2020Anchor.compress` drives from main/model/pnet.py:46-49,70-73.  It is host code
// in the reference as well; the GPU produces the symbols and CDF indexes, this packs them.
#include <stdint.h>
#include <string.h>
#include <vector>

#include "../../include/tdvc_hip.h"

void tdvc_set_error(const char* fmt, ...);

namespace {
constexpr uint64_t kL = 1ull << 31;
constexpr int kPrec = 16;
constexpr int kBypassBits = 4;
constexpr int kMaxBypass = (1 << kBypassBits) - 1;

struct Sym { uint16_t start, range; bool bypass; };
}  // namespace

extern "C" int64_t tdvc_rans_encode(const int32_t* symbols, const int32_t* indexes, int64_t n,
                                    const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                                    const int32_t* offsets, uint8_t* out, int64_t cap) {
  if (!symbols || !indexes || !cdfs || !cdf_sizes || !offsets || !out || n < 0) {
    tdvc_set_error("tdvc_rans_encode: null argument");
    return TDVC_EINVAL;
  }
  std::vector<Sym> syms;
  syms.reserve((size_t)n + 16);
  for (int64_t i = 0; i < n; ++i) {
    const int32_t ci = indexes[i];
    const int32_t* cdf = cdfs + (int64_t)ci * cdf_stride;
    const int32_t max_value = cdf_sizes[ci] - 2;
    int32_t value = symbols[i] - offsets[ci];
    uint32_t raw = 0;
    if (value < 0) {
      raw = (uint32_t)(-2 * value - 1);
      value = max_value;
    } else if (value >= max_value) {
      raw = (uint32_t)(2 * (value - max_value));
      value = max_value;
    }
    syms.push_back({(uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), false});
    if (value == max_value) {
      int32_t nb = 0;
      while ((raw >> (nb * kBypassBits)) != 0) ++nb;
      int32_t v = nb;
      while (v >= kMaxBypass) {
        syms.push_back({(uint16_t)kMaxBypass, (uint16_t)(kMaxBypass + 1), true});
        v -= kMaxBypass;
      }
      syms.push_back({(uint16_t)v, (uint16_t)(v + 1), true});
      for (int32_t j = 0; j < nb; ++j) {
        const int32_t d = (raw >> (j * kBypassBits)) & kMaxBypass;
        syms.push_back({(uint16_t)d, (uint16_t)(d + 1), true});
      }
    }
  }
  std::vector<uint32_t> words;
  words.reserve(syms.size() / 2 + 4);
  uint64_t x = kL;
  for (size_t k = syms.size(); k-- > 0;) {
    const Sym s = syms[k];
    if (!s.bypass) {
      const uint64_t x_max = ((kL >> kPrec) << 32) * s.range;
      if (x >= x_max) { words.push_back((uint32_t)x); x >>= 32; }
      x = ((x / s.range) << kPrec) + (x % s.range) + s.start;
    } else {
      const uint32_t freq = 1u << (16 - kBypassBits);
      const uint64_t x_max = ((kL >> 16) << 32) * freq;
      if (x >= x_max) { words.push_back((uint32_t)x); x >>= 32; }
      x = (x << kBypassBits) | s.start;
    }
  }
  words.push_back((uint32_t)(x >> 32));
  words.push_back((uint32_t)x);
  const int64_t nbytes = (int64_t)words.size() * 4;
  if (nbytes > cap) {
    tdvc_set_error("tdvc_rans_encode: output buffer too small (%lld > %lld)", (long long)nbytes, (long long)cap);
    return TDVC_EINVAL;
  }
  // stream order = reverse emission order, little-endian words
  for (size_t k = 0; k < words.size(); ++k) {
    const uint32_t w = words[words.size() - 1 - k];
    out[4 * k + 0] = (uint8_t)(w);
    out[4 * k + 1] = (uint8_t)(w >> 8);
    out[4 * k + 2] = (uint8_t)(w >> 16);
    out[4 * k + 3] = (uint8_t)(w >> 24);
  }
  return nbytes;
}

namespace {
struct Dec {
  const uint8_t* d; int64_t nwords; int64_t pos; uint64_t x; bool bad;
  uint32_t word() {
    if (pos >= nwords) { bad = true; return 0; }
    const uint8_t* p = d + 4 * pos++;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
  }
  void renorm() { if (x < kL) x = (x << 32) | word(); }
  uint32_t bits(int nb) { const uint32_t v = (uint32_t)(x & ((1u << nb) - 1)); x >>= nb; renorm(); return v; }
};
}  // namespace

static int dec_run(Dec& dc, const int32_t* indexes, int64_t n, const int32_t* cdfs, int32_t cdf_stride,
                   const int32_t* cdf_sizes, const int32_t* offsets, int32_t* symbols_out);

extern "C" int tdvc_rans_decode(const uint8_t* data, int64_t nbytes, const int32_t* indexes, int64_t n,
                                const int32_t* cdfs, int32_t cdf_stride, const int32_t* cdf_sizes,
                                const int32_t* offsets, int32_t* symbols_out) {
  if (!data || !indexes || !cdfs || !cdf_sizes || !offsets || !symbols_out || nbytes < 8 || (nbytes & 3)) {
    tdvc_set_error("tdvc_rans_decode: bad arguments (stream must be >= 8 bytes, multiple of 4)");
    return TDVC_EINVAL;
  }
  Dec dc{data, nbytes / 4, 0, 0, false};
  const uint64_t lo = dc.word(), hi = dc.word();
  dc.x = lo | (hi << 32);
  return dec_run(dc, indexes, n, cdfs, cdf_stride, cdf_sizes, offsets, symbols_out);
}

// stateful decoder (compressai RansDecoder.set_stream / decode_stream): the autoregressive decoder
// learns the CDF indexes of a position only after the previous positions are decoded
extern "C" void* tdvc_rans_decoder_create(const uint8_t* data, int64_t nbytes) {
  if (!data || nbytes < 8 || (nbytes & 3)) {
    tdvc_set_error("tdvc_rans_decoder_create: stream must be >= 8 bytes, multiple of 4");
    return nullptr;
  }
  uint8_t* copy = new uint8_t[(size_t)nbytes];
  memcpy(copy, data, (size_t)nbytes);
  Dec* dc = new Dec{copy, nbytes / 4, 0, 0, false};
  const uint64_t lo = dc->word(), hi = dc->word();
  dc->x = lo | (hi << 32);
  return dc;
}
extern "C" int tdvc_rans_decoder_decode(void* handle, const int32_t* indexes, int64_t n, const int32_t* cdfs,
                                        int32_t cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets,
                                        int32_t* symbols_out) {
  if (!handle || !indexes || !cdfs || !cdf_sizes || !offsets || !symbols_out) {
    tdvc_set_error("tdvc_rans_decoder_decode: null argument");
    return TDVC_EINVAL;
  }
  return dec_run(*static_cast<Dec*>(handle), indexes, n, cdfs, cdf_stride, cdf_sizes, offsets, symbols_out);
}
extern "C" void tdvc_rans_decoder_destroy(void* handle) {
  if (!handle) return;
  Dec* dc = static_cast<Dec*>(handle);
  delete[] dc->d;
  delete dc;
}

static int dec_run(Dec& dc, const int32_t* indexes, int64_t n, const int32_t* cdfs, int32_t cdf_stride,
                   const int32_t* cdf_sizes, const int32_t* offsets, int32_t* symbols_out) {
  const uint32_t mask = (1u << kPrec) - 1;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t ci = indexes[i];
    const int32_t* cdf = cdfs + (int64_t)ci * cdf_stride;
    const int32_t max_value = cdf_sizes[ci] - 2;
    const uint32_t cum = (uint32_t)(dc.x & mask);
    int32_t s = 0;
    while (s <= max_value && (uint32_t)cdf[s + 1] <= cum) ++s;
    if (s > max_value) { tdvc_set_error("tdvc_rans_decode: corrupt stream at symbol %lld", (long long)i); return TDVC_EINVAL; }
    dc.x = (uint64_t)(cdf[s + 1] - cdf[s]) * (dc.x >> kPrec) + cum - (uint32_t)cdf[s];
    dc.renorm();
    int32_t value = s;
    if (value == max_value) {
      int32_t v = (int32_t)dc.bits(kBypassBits);
      int32_t nb = v;
      while (v == kMaxBypass) { v = (int32_t)dc.bits(kBypassBits); nb += v; if (dc.bad) break; }
      uint32_t raw = 0;
      for (int32_t j = 0; j < nb && j < 8; ++j) raw |= dc.bits(kBypassBits) << (j * kBypassBits);
      value = (int32_t)(raw >> 1);
      if (raw & 1) value = -value - 1; else value += max_value;
    }
    if (dc.bad) { tdvc_set_error("tdvc_rans_decode: stream exhausted at symbol %lld", (long long)i); return TDVC_EINVAL; }
    symbols_out[i] = value + offsets[ci];
  }
  return TDVC_OK;
}

// compressai `_CXX.pmf_to_quantized_cdf` (cpp_exts/ops/ops.cpp): round to 2^precision, renormalise with
// integer division, prefix-sum, force the last entry, then repair zero-width bins by stealing one count
// from the narrowest bin wider than 1.  cdf_out has n + 1 entries.
extern "C" int tdvc_pmf_to_quantized_cdf(const float* pmf, int n, int precision, int32_t* cdf_out) {
  if (!pmf || !cdf_out || n < 1 || precision < 1 || precision > 16) {
    tdvc_set_error("tdvc_pmf_to_quantized_cdf: bad arguments");
    return TDVC_EINVAL;
  }
  std::vector<uint32_t> cdf((size_t)n + 1);
  cdf[0] = 0;
  for (int i = 0; i < n; ++i) {
    const float v = pmf[i] * (float)(1 << precision);
    cdf[i + 1] = (uint32_t)(v + 0.5f >= 0.f ? (long long)(v + 0.5f) : 0);     // std::round for v >= 0
    if ((float)cdf[i + 1] - v > 0.5f) cdf[i + 1] -= 1;                          // guard the float add
  }
  uint64_t total = 0;
  for (uint32_t c : cdf) total += c;
  if (total == 0) { tdvc_set_error("tdvc_pmf_to_quantized_cdf: empty pmf"); return TDVC_EINVAL; }
  for (auto& c : cdf) c = (uint32_t)((((uint64_t)1 << precision) * c) / total);
  for (size_t i = 1; i < cdf.size(); ++i) cdf[i] += cdf[i - 1];
  cdf.back() = 1u << precision;
  const int m = (int)cdf.size();
  for (int i = 0; i < m - 1; ++i) {
    if (cdf[i] == cdf[i + 1]) {
      uint32_t best_freq = ~0u;
      int best = -1;
      for (int j = 0; j < m - 1; ++j) {
        const uint32_t f = cdf[j + 1] - cdf[j];
        if (f > 1 && f < best_freq) { best_freq = f; best = j; }
      }
      if (best == -1) { tdvc_set_error("tdvc_pmf_to_quantized_cdf: cannot repair zero-width bin"); return TDVC_EINVAL; }
      if (best < i) { for (int j = best + 1; j <= i; ++j) cdf[j]--; }
      else { for (int j = i + 1; j <= best; ++j) cdf[j]++; }
    }
  }
  for (int i = 0; i < m; ++i) cdf_out[i] = (int32_t)cdf[i];
  return TDVC_OK;
}
