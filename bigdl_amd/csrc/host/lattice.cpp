// 2-bit / 1.6-bit vector-quantized formats (names gguf_iq2_xxs, gguf_iq2_xs,
// gguf_iq1_s / gguf_iq1_m in the qtype table).
//
// The reference reaches these through a closed quantizer
// (python/llm/src/ipex_llm/transformers/low_bit_linear.py:146,163-171) whose codebooks
// are not in its tree, so the codebooks here are generated deterministically by this
// library: groups of 8 weights are coded as an index into a grid of 8-dimensional
// magnitude patterns plus sign bits.  Byte sizes per 256 weights match the public ggml
// formats (66 / 74 / 50 bytes); codebook contents do not, so files are only
// interchangeable with this library.
#include "qformats.h"
#include <algorithm>
#include <cfloat>
#include <mutex>
#include <vector>

namespace bigdl {

namespace {

constexpr int N_COMBO = 6561;  // 3^8

struct Grids {
  uint8_t g2[512][8];       // magnitudes in {1,3,5}; first 256 entries are the xxs grid
  uint16_t map2_256[N_COMBO];  // base-3 combo -> nearest entry among the first 256
  uint16_t map2_512[N_COMBO];
  int8_t g1[2048][8];       // ternary patterns in {-1,0,1}
  uint16_t map1[N_COMBO];
};

Grids* G = nullptr;
std::once_flag g_once;

inline void combo_digits(int c, int* dgt) {
  for (int i = 0; i < 8; ++i) { dgt[i] = c % 3; c /= 3; }
}

void build() {
  G = new Grids();
  // order all 6561 patterns by energy, ties by index: low-energy patterns are the most
  // probable ones for bell-shaped weights
  std::vector<int> order(N_COMBO);
  std::vector<int> e2(N_COMBO), e1(N_COMBO);
  for (int c = 0; c < N_COMBO; ++c) {
    int d[8];
    combo_digits(c, d);
    int s2 = 0, s1 = 0;
    for (int i = 0; i < 8; ++i) { const int m = 2 * d[i] + 1; s2 += m * m; s1 += d[i] != 1; }
    e2[c] = s2; e1[c] = s1;
    order[c] = c;
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return e2[a] < e2[b]; });
  std::vector<int> sel2(order.begin(), order.begin() + 512);
  for (int k = 0; k < 512; ++k) {
    int d[8];
    combo_digits(sel2[k], d);
    for (int i = 0; i < 8; ++i) G->g2[k][i] = (uint8_t)(2 * d[i] + 1);
  }
  for (int c = 0; c < N_COMBO; ++c) {
    int d[8];
    combo_digits(c, d);
    int best256 = 0, best512 = 0, bd256 = 1 << 30, bd512 = 1 << 30;
    for (int k = 0; k < 512; ++k) {
      int dist = 0;
      for (int i = 0; i < 8; ++i) { const int t = (2 * d[i] + 1) - G->g2[k][i]; dist += t * t; }
      if (k < 256 && dist < bd256) { bd256 = dist; best256 = k; }
      if (dist < bd512) { bd512 = dist; best512 = k; }
    }
    G->map2_256[c] = (uint16_t)best256;
    G->map2_512[c] = (uint16_t)best512;
  }
  // ternary grid: digit 0,1,2 -> -1,0,+1 ; fewest non-zeros first
  for (int c = 0; c < N_COMBO; ++c) order[c] = c;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return e1[a] < e1[b]; });
  for (int k = 0; k < 2048; ++k) {
    int d[8];
    combo_digits(order[k], d);
    for (int i = 0; i < 8; ++i) G->g1[k][i] = (int8_t)(d[i] - 1);
  }
  for (int c = 0; c < N_COMBO; ++c) {
    int d[8];
    combo_digits(c, d);
    int best = 0, bd = 1 << 30;
    for (int k = 0; k < 2048; ++k) {
      int dist = 0;
      for (int i = 0; i < 8; ++i) { const int t = (d[i] - 1) - G->g1[k][i]; dist += t * t; }
      if (dist < bd) { bd = dist; best = k; if (!dist) break; }
    }
    G->map1[c] = (uint16_t)best;
  }
}

inline const Grids& grids() {
  std::call_once(g_once, build);
  return *G;
}

// code 8 magnitudes |x|/db to the nearest {1,3,5} pattern; returns base-3 combo
inline int combo_of_mags(const float* x, float idb) {
  int c = 0, mul = 1;
  for (int i = 0; i < 8; ++i) {
    const float m = std::fabs(x[i]) * idb;
    const int d = m < 2.0f ? 0 : (m < 4.0f ? 1 : 2);
    c += d * mul;
    mul *= 3;
  }
  return c;
}

// 7 explicit sign bits; the 8th is implied by even parity.  When the true signs have odd
// parity the sign of the smallest-magnitude element is flipped.
inline uint8_t signs7(const float* x) {
  uint8_t s = 0;
  int nneg = 0, imin = 0;
  float amin = FLT_MAX;
  for (int i = 0; i < 8; ++i) {
    if (x[i] < 0) { s |= (uint8_t)(1u << i); ++nneg; }
    if (std::fabs(x[i]) < amin) { amin = std::fabs(x[i]); imin = i; }
  }
  if (nneg & 1) s ^= (uint8_t)(1u << imin);
  return s & 0x7f;
}

inline uint8_t full_signs(uint8_t s7) {
  return (uint8_t)(s7 | ((__builtin_popcount(s7) & 1) << 7));
}

}  // namespace

void quantize_row_iq2_xxs(const float* x, block_iq2_xxs* y, int64_t n, const float* w) {
  (void)w;
  const Grids& g = grids();
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const float* xb = x + i * QK_K;
    float db[8], dmax = 0.0f;
    for (int b = 0; b < 8; ++b) {
      float amax = 0.0f;
      for (int l = 0; l < 32; ++l) amax = std::max(amax, std::fabs(xb[32 * b + l]));
      db[b] = amax / 5.5f;
      dmax = std::max(dmax, db[b]);
    }
    const float d = dmax / (15.5f * 0.25f);
    y[i].d = fp32_to_fp16(d);
    const float dq = fp16_to_fp32(y[i].d);
    for (int b = 0; b < 8; ++b) {
      int s = dq > 0 ? (int)std::nearbyint(db[b] / (dq * 0.25f) - 0.5f) : 0;
      s = std::max(0, std::min(15, s));
      const float dl = dq * (0.5f + s) * 0.25f;
      const float idl = dl > 0 ? 1.0f / dl : 0.0f;
      uint32_t idx = 0, sg = 0;
      for (int k = 0; k < 4; ++k) {
        const float* xg = xb + 32 * b + 8 * k;
        idx |= (uint32_t)g.map2_256[combo_of_mags(xg, idl)] << (8 * k);
        sg |= (uint32_t)signs7(xg) << (7 * k);
      }
      sg |= (uint32_t)s << 28;
      y[i].qs[4 * b + 0] = (uint16_t)(idx & 0xffff);
      y[i].qs[4 * b + 1] = (uint16_t)(idx >> 16);
      y[i].qs[4 * b + 2] = (uint16_t)(sg & 0xffff);
      y[i].qs[4 * b + 3] = (uint16_t)(sg >> 16);
    }
  }
}

void dequantize_row_iq2_xxs(const block_iq2_xxs* x, float* y, int64_t n) {
  const Grids& g = grids();
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const float d = fp16_to_fp32(x[i].d);
    for (int b = 0; b < 8; ++b) {
      const uint32_t idx = x[i].qs[4 * b] | ((uint32_t)x[i].qs[4 * b + 1] << 16);
      const uint32_t sg = x[i].qs[4 * b + 2] | ((uint32_t)x[i].qs[4 * b + 3] << 16);
      const float dl = d * (0.5f + (sg >> 28)) * 0.25f;
      for (int k = 0; k < 4; ++k) {
        const uint8_t* grid = g.g2[(idx >> (8 * k)) & 255];
        const uint8_t signs = full_signs((sg >> (7 * k)) & 127);
        for (int j = 0; j < 8; ++j) *y++ = dl * grid[j] * ((signs >> j) & 1 ? -1.0f : 1.0f);
      }
    }
  }
}

void quantize_row_iq2_xs(const float* x, block_iq2_xs* y, int64_t n, const float* w) {
  (void)w;
  const Grids& g = grids();
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const float* xb = x + i * QK_K;
    float db[16], dmax = 0.0f;
    for (int b = 0; b < 16; ++b) {
      float amax = 0.0f;
      for (int l = 0; l < 16; ++l) amax = std::max(amax, std::fabs(xb[16 * b + l]));
      db[b] = amax / 5.5f;
      dmax = std::max(dmax, db[b]);
    }
    const float d = dmax / (15.5f * 0.25f);
    y[i].d = fp32_to_fp16(d);
    const float dq = fp16_to_fp32(y[i].d);
    for (int b = 0; b < 16; ++b) {
      int s = dq > 0 ? (int)std::nearbyint(db[b] / (dq * 0.25f) - 0.5f) : 0;
      s = std::max(0, std::min(15, s));
      if (b & 1) y[i].scales[b / 2] |= (uint8_t)(s << 4);
      else y[i].scales[b / 2] = (uint8_t)s;
      const float dl = dq * (0.5f + s) * 0.25f;
      const float idl = dl > 0 ? 1.0f / dl : 0.0f;
      for (int k = 0; k < 2; ++k) {
        const float* xg = xb + 16 * b + 8 * k;
        y[i].qs[2 * b + k] = (uint16_t)(g.map2_512[combo_of_mags(xg, idl)] | ((uint16_t)signs7(xg) << 9));
      }
    }
  }
}

void dequantize_row_iq2_xs(const block_iq2_xs* x, float* y, int64_t n) {
  const Grids& g = grids();
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const float d = fp16_to_fp32(x[i].d);
    for (int b = 0; b < 16; ++b) {
      const int s = (b & 1) ? (x[i].scales[b / 2] >> 4) : (x[i].scales[b / 2] & 0xF);
      const float dl = d * (0.5f + s) * 0.25f;
      for (int k = 0; k < 2; ++k) {
        const uint16_t q = x[i].qs[2 * b + k];
        const uint8_t* grid = g.g2[q & 511];
        const uint8_t signs = full_signs((uint8_t)(q >> 9));
        for (int j = 0; j < 8; ++j) *y++ = dl * grid[j] * ((signs >> j) & 1 ? -1.0f : 1.0f);
      }
    }
  }
}

static constexpr float IQ1_DELTA = 0.125f;

void quantize_row_iq1_s(const float* x, block_iq1_s* y, int64_t n, const float* w) {
  (void)w;
  const Grids& g = grids();
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const float* xb = x + i * QK_K;
    float db[8], dmax = 0.0f;
    for (int b = 0; b < 8; ++b) {
      // ternary levels sit at +-dl: match the rms of the block rather than its max
      float s2 = 0.0f;
      for (int l = 0; l < 32; ++l) s2 += xb[32 * b + l] * xb[32 * b + l];
      db[b] = 1.25f * std::sqrt(s2 / 32.0f);
      dmax = std::max(dmax, db[b]);
    }
    const float d = dmax / 15.0f;
    y[i].d = fp32_to_fp16(d);
    const float dq = fp16_to_fp32(y[i].d);
    for (int b = 0; b < 8; ++b) {
      int s = dq > 0 ? (int)std::nearbyint((db[b] / dq - 1.0f) * 0.5f) : 0;
      s = std::max(0, std::min(7, s));
      const float dl = dq * (2 * s + 1);
      const float idl = dl > 0 ? 1.0f / dl : 0.0f;
      float mean = 0.0f;
      for (int l = 0; l < 32; ++l) mean += xb[32 * b + l];
      const bool neg_delta = mean < 0.0f;
      const float delta = neg_delta ? -IQ1_DELTA : IQ1_DELTA;
      uint16_t qh = (uint16_t)((s << 12) | (neg_delta ? 0x8000 : 0));
      for (int k = 0; k < 4; ++k) {
        const float* xg = xb + 32 * b + 8 * k;
        int c = 0, mul = 1;
        for (int j = 0; j < 8; ++j) {
          const float t = xg[j] * idl - delta;
          const int dgt = t < -0.5f ? 0 : (t > 0.5f ? 2 : 1);
          c += dgt * mul;
          mul *= 3;
        }
        const uint16_t idx = g.map1[c];
        y[i].qs[4 * b + k] = (uint8_t)(idx & 255);
        qh |= (uint16_t)((idx >> 8) << (3 * k));
      }
      y[i].qh[b] = qh;
    }
  }
}

void dequantize_row_iq1_s(const block_iq1_s* x, float* y, int64_t n) {
  const Grids& g = grids();
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const float d = fp16_to_fp32(x[i].d);
    for (int b = 0; b < 8; ++b) {
      const uint16_t qh = x[i].qh[b];
      const float dl = d * (2 * ((qh >> 12) & 7) + 1);
      const float delta = (qh & 0x8000) ? -IQ1_DELTA : IQ1_DELTA;
      for (int k = 0; k < 4; ++k) {
        const int8_t* grid = g.g1[x[i].qs[4 * b + k] | (((qh >> (3 * k)) & 7) << 8)];
        for (int j = 0; j < 8; ++j) *y++ = dl * (grid[j] + delta);
      }
    }
  }
}

}  // namespace bigdl
