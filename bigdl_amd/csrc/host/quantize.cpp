// Host block quantizers (OpenMP over rows happens in api.cpp).
//
// Role: the counterpart of the reference's closed `ggml_quantize_tensor` /
// `ggml_quantize_tensor_with_weights` entry points
// (python/llm/src/ipex_llm/ggml/model/llama/llama_cpp.py:946-991), written from the
// public ggml format definitions; the k-quant search loops are a compact weighted
// least-squares search, not a transcription of ggml's.
#include "qformats.h"
#include <algorithm>
#include <cfloat>
#include <mutex>
#include <vector>

namespace bigdl {

const float NF4_TABLE[16] = {
    -1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
    -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
    0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
    0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};
// 3-bit normal-float: quantiles of N(0,1) normalised to [-1, 1], with an exact zero
const float NF3_TABLE[8] = {-1.0f, -0.5350227355957031f, -0.2469314038753510f, 0.0f,
                            0.1833375245332718f, 0.3819939494132996f, 0.6229856610298157f, 1.0f};
const float E2M1_TABLE[16] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f,
                              -0.0f, -0.5f, -1.0f, -1.5f, -2.0f, -3.0f, -4.0f, -6.0f};
float E3M2_TABLE[64];
float E4M3_TABLE[256];
float E5M2_TABLE[256];

static std::once_flag g_tables_once;

void init_tables() {
  std::call_once(g_tables_once, [] {
    for (int c = 0; c < 64; ++c) {  // e3m2, bias 3, no inf/nan
      const int s = c >> 5, e = (c >> 2) & 7, m = c & 3;
      const float v = e == 0 ? std::ldexp((float)m / 4.0f, -2) : std::ldexp(1.0f + m / 4.0f, e - 3);
      E3M2_TABLE[c] = s ? -v : v;
    }
    for (int c = 0; c < 256; ++c) {  // OCP e4m3fn, bias 7; 0x7f/0xff = NaN (decoded as 0 here, never produced)
      const int s = c >> 7, e = (c >> 3) & 15, m = c & 7;
      float v = e == 0 ? std::ldexp((float)m / 8.0f, -6) : std::ldexp(1.0f + m / 8.0f, e - 7);
      if ((c & 0x7f) == 0x7f) v = 0.0f;
      E4M3_TABLE[c] = s ? -v : v;
    }
    for (int c = 0; c < 256; ++c) {  // OCP e5m2, bias 15; e==31 is inf/nan (decoded as 0, never produced)
      const int s = c >> 7, e = (c >> 2) & 31, m = c & 3;
      float v = e == 0 ? std::ldexp((float)m / 4.0f, -14) : std::ldexp(1.0f + m / 4.0f, e - 15);
      if (e == 31) v = 0.0f;
      E5M2_TABLE[c] = s ? -v : v;
    }
  });
}

int qk_size(int qtype) {
  switch (qtype) {
    case Q_SYM_INT4: case Q_ASYM_INT4: case Q_SYM_INT5: case Q_ASYM_INT5: case Q_SYM_INT8:
    case Q_NF3: case Q_FP4: case Q_FP6: case Q_FP8_E4M3: case Q_FP8_E5M2:
      return QK;
    case Q_NF4: return QK_NF4;
    case Q_WOQ_INT4: return QK_WOQ;
    case Q_Q2_K: case Q_Q4_K: case Q_Q5_K: case Q_Q6_K: case Q_FP6_K:
    case Q_IQ2_XXS: case Q_IQ2_XS: case Q_IQ1_S: case Q_IQ1_M:
      return QK_K;
    case Q_FP16: case Q_BF16: return 1;
    default: return 0;
  }
}

int type_size(int qtype) {
  switch (qtype) {
    case Q_SYM_INT4: return sizeof(block_q4_0);
    case Q_ASYM_INT4: return sizeof(block_q4_1);
    case Q_SYM_INT5: return sizeof(block_q5_0);
    case Q_ASYM_INT5: return sizeof(block_q5_1);
    case Q_SYM_INT8: return sizeof(block_q8_0);
    case Q_NF4: return sizeof(block_nf4);
    case Q_NF3: return sizeof(block_nf3);
    case Q_FP4: return sizeof(block_fp4);
    case Q_FP6: return sizeof(block_fp6);
    case Q_FP8_E4M3: case Q_FP8_E5M2: return sizeof(block_fp8);
    case Q_WOQ_INT4: return sizeof(block_woq4);
    case Q_Q2_K: return sizeof(block_q2_K);
    case Q_Q4_K: return sizeof(block_q4_K);
    case Q_Q5_K: return sizeof(block_q5_K);
    case Q_Q6_K: return sizeof(block_q6_K);
    case Q_FP6_K: return sizeof(block_fp6_K);
    case Q_IQ2_XXS: return sizeof(block_iq2_xxs);
    case Q_IQ2_XS: return sizeof(block_iq2_xs);
    case Q_IQ1_S: case Q_IQ1_M: return sizeof(block_iq1_s);
    case Q_FP16: case Q_BF16: return 2;
    default: return 0;
  }
}

// ---------------------------------------------------------------------------
// nearest-code encoders for the small float formats
// ---------------------------------------------------------------------------
template <int N>
static inline int nearest_sorted(const float (&tab)[N], float x) {
  int best = 0;
  float bd = std::fabs(tab[0] - x);
  for (int i = 1; i < N; ++i) {
    const float d = std::fabs(tab[i] - x);
    if (d < bd) { bd = d; best = i; }
  }
  return best;
}

uint8_t encode_nf4(float x) { return (uint8_t)nearest_sorted(NF4_TABLE, x); }
uint8_t encode_nf3(float x) { return (uint8_t)nearest_sorted(NF3_TABLE, x); }

uint8_t encode_e2m1(float x) {
  const uint8_t s = std::signbit(x) ? 8 : 0;
  const float a = std::fabs(x);
  // midpoints between {0,.5,1,1.5,2,3,4,6}; ties go to the even mantissa
  uint8_t c;
  if (a <= 0.25f) c = 0; else if (a < 0.75f) c = 1; else if (a <= 1.25f) c = 2;
  else if (a < 1.75f) c = 3; else if (a <= 2.5f) c = 4; else if (a < 3.5f) c = 5;
  else if (a <= 5.0f) c = 6; else c = 7;
  return s | c;
}

// generic sign|exp|mantissa encoder with saturation, round-to-nearest-even
static inline uint32_t encode_small_float(float x, int ebits, int mbits, int bias, float maxv) {
  const uint32_t s = std::signbit(x) ? 1u : 0u;
  float a = std::fabs(x);
  if (!(a == a)) a = 0.0f;
  if (a > maxv) a = maxv;
  uint32_t code;
  const float min_normal = std::ldexp(1.0f, 1 - bias);
  if (a < min_normal) {
    const float q = a / std::ldexp(1.0f, 1 - bias - mbits);  // in units of the subnormal step
    uint32_t m = (uint32_t)std::nearbyint(q);
    code = m;  // m == 2^mbits rolls into exponent 1, mantissa 0 -- correct by construction
  } else {
    int e;
    const float fr = std::frexp(a, &e);  // a = fr * 2^e, fr in [0.5,1)
    int E = e - 1;
    float mant = fr * 2.0f - 1.0f;
    uint32_t m = (uint32_t)std::nearbyint(mant * (float)(1 << mbits));
    if (m == (1u << mbits)) { m = 0; ++E; }
    code = ((uint32_t)(E + bias) << mbits) | m;
  }
  (void)ebits;
  return (s << (ebits + mbits)) | code;
}

uint8_t encode_e3m2(float x) { return (uint8_t)encode_small_float(x, 3, 2, 3, 28.0f); }
uint8_t encode_fp8_e4m3(float x) { return (uint8_t)encode_small_float(x, 4, 3, 7, 448.0f); }
uint8_t encode_fp8_e5m2(float x) { return (uint8_t)encode_small_float(x, 5, 2, 15, 57344.0f); }

// ---------------------------------------------------------------------------
// simple 32-element block formats
// ---------------------------------------------------------------------------
static void quantize_row_q4_0(const float* x, block_q4_0* y, int64_t n) {
  for (int64_t i = 0; i < n / QK; ++i) {
    float amax = 0.0f, mx = 0.0f;
    for (int j = 0; j < QK; ++j) {
      const float v = x[i * QK + j];
      if (amax < std::fabs(v)) { amax = std::fabs(v); mx = v; }
    }
    const float d = mx / -8.0f;
    const float id = d ? 1.0f / d : 0.0f;
    y[i].d = fp32_to_fp16(d);
    for (int j = 0; j < QK / 2; ++j) {
      const float x0 = x[i * QK + j] * id, x1 = x[i * QK + QK / 2 + j] * id;
      const uint8_t q0 = (uint8_t)std::min(15, (int)(x0 + 8.5f));
      const uint8_t q1 = (uint8_t)std::min(15, (int)(x1 + 8.5f));
      y[i].qs[j] = q0 | (q1 << 4);
    }
  }
}

static void quantize_row_q4_1(const float* x, block_q4_1* y, int64_t n) {
  for (int64_t i = 0; i < n / QK; ++i) {
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int j = 0; j < QK; ++j) { mn = std::min(mn, x[i * QK + j]); mx = std::max(mx, x[i * QK + j]); }
    const float d = (mx - mn) / 15.0f;
    const float id = d ? 1.0f / d : 0.0f;
    y[i].d = fp32_to_fp16(d);
    y[i].m = fp32_to_fp16(mn);
    for (int j = 0; j < QK / 2; ++j) {
      const float x0 = (x[i * QK + j] - mn) * id, x1 = (x[i * QK + QK / 2 + j] - mn) * id;
      const uint8_t q0 = (uint8_t)std::min(15, (int)(x0 + 0.5f));
      const uint8_t q1 = (uint8_t)std::min(15, (int)(x1 + 0.5f));
      y[i].qs[j] = q0 | (q1 << 4);
    }
  }
}

static void quantize_row_q5_0(const float* x, block_q5_0* y, int64_t n) {
  for (int64_t i = 0; i < n / QK; ++i) {
    float amax = 0.0f, mx = 0.0f;
    for (int j = 0; j < QK; ++j) {
      const float v = x[i * QK + j];
      if (amax < std::fabs(v)) { amax = std::fabs(v); mx = v; }
    }
    const float d = mx / -16.0f;
    const float id = d ? 1.0f / d : 0.0f;
    y[i].d = fp32_to_fp16(d);
    uint32_t qh = 0;
    for (int j = 0; j < QK / 2; ++j) {
      const float x0 = x[i * QK + j] * id, x1 = x[i * QK + QK / 2 + j] * id;
      const uint8_t q0 = (uint8_t)std::min(31, (int)(x0 + 16.5f));
      const uint8_t q1 = (uint8_t)std::min(31, (int)(x1 + 16.5f));
      y[i].qs[j] = (q0 & 0x0F) | ((q1 & 0x0F) << 4);
      qh |= ((q0 & 0x10u) >> 4) << (j + 0);
      qh |= ((q1 & 0x10u) >> 4) << (j + QK / 2);
    }
    std::memcpy(y[i].qh, &qh, 4);
  }
}

static void quantize_row_q5_1(const float* x, block_q5_1* y, int64_t n) {
  for (int64_t i = 0; i < n / QK; ++i) {
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int j = 0; j < QK; ++j) { mn = std::min(mn, x[i * QK + j]); mx = std::max(mx, x[i * QK + j]); }
    const float d = (mx - mn) / 31.0f;
    const float id = d ? 1.0f / d : 0.0f;
    y[i].d = fp32_to_fp16(d);
    y[i].m = fp32_to_fp16(mn);
    uint32_t qh = 0;
    for (int j = 0; j < QK / 2; ++j) {
      const float x0 = (x[i * QK + j] - mn) * id, x1 = (x[i * QK + QK / 2 + j] - mn) * id;
      const uint8_t q0 = (uint8_t)std::min(31, (int)(x0 + 0.5f));
      const uint8_t q1 = (uint8_t)std::min(31, (int)(x1 + 0.5f));
      y[i].qs[j] = (q0 & 0x0F) | ((q1 & 0x0F) << 4);
      qh |= ((q0 & 0x10u) >> 4) << (j + 0);
      qh |= ((q1 & 0x10u) >> 4) << (j + QK / 2);
    }
    std::memcpy(y[i].qh, &qh, 4);
  }
}

static void quantize_row_q8_0(const float* x, block_q8_0* y, int64_t n) {
  for (int64_t i = 0; i < n / QK; ++i) {
    float amax = 0.0f;
    for (int j = 0; j < QK; ++j) amax = std::max(amax, std::fabs(x[i * QK + j]));
    const float d = amax / 127.0f;
    const float id = d ? 1.0f / d : 0.0f;
    y[i].d = fp32_to_fp16(d);
    for (int j = 0; j < QK; ++j) y[i].qs[j] = (int8_t)std::nearbyint(x[i * QK + j] * id);
  }
}

static void quantize_row_nf4(const float* x, block_nf4* y, int64_t n) {
  for (int64_t i = 0; i < n / QK_NF4; ++i) {
    float amax = 0.0f;
    for (int j = 0; j < QK_NF4; ++j) amax = std::max(amax, std::fabs(x[i * QK_NF4 + j]));
    // the stored scale is the fp16-rounded absmax; codes are chosen against that value
    const fp16_t dh = fp32_to_fp16(amax);
    const float d = fp16_to_fp32(dh);
    const float id = d ? 1.0f / d : 0.0f;
    y[i].d = dh;
    for (int j = 0; j < QK_NF4 / 2; ++j) {
      const uint8_t q0 = encode_nf4(x[i * QK_NF4 + j] * id);
      const uint8_t q1 = encode_nf4(x[i * QK_NF4 + QK_NF4 / 2 + j] * id);
      y[i].qs[j] = q0 | (q1 << 4);
    }
  }
}

static void quantize_row_nf3(const float* x, block_nf3* y, int64_t n) {
  for (int64_t i = 0; i < n / QK; ++i) {
    float amax = 0.0f;
    for (int j = 0; j < QK; ++j) amax = std::max(amax, std::fabs(x[i * QK + j]));
    const fp16_t dh = fp32_to_fp16(amax);
    const float d = fp16_to_fp32(dh);
    const float id = d ? 1.0f / d : 0.0f;
    y[i].d = dh;
    std::memset(y[i].ql, 0, sizeof(y[i].ql));
    std::memset(y[i].qh, 0, sizeof(y[i].qh));
    for (int j = 0; j < QK; ++j) {
      const uint8_t q = encode_nf3(x[i * QK + j] * id);
      y[i].ql[j / 4] |= (q & 3) << (2 * (j % 4));
      y[i].qh[j / 8] |= (q >> 2) << (j % 8);
    }
  }
}

// pick the block scale for a small-float format: absmax/maxcode, optionally refined by
// an MSE search over a few shrink factors (the reference enables this for fp6 and
// fp8_e4m3: python/llm/src/ipex_llm/transformers/convert.py:456-466)
template <typename Enc, typename Dec>
static float search_scale(const float* x, int n, float maxcode, bool search, Enc enc, Dec dec) {
  float amax = 0.0f;
  for (int j = 0; j < n; ++j) amax = std::max(amax, std::fabs(x[j]));
  float d = fp16_to_fp32(fp32_to_fp16(amax / maxcode));
  if (!search || d == 0.0f) return d;
  float best_d = d, best_err = FLT_MAX;
  for (int s = 0; s <= 8; ++s) {
    const float cand = fp16_to_fp32(fp32_to_fp16(amax / maxcode * (1.0f - 0.03f * s)));
    if (cand == 0.0f) continue;
    const float id = 1.0f / cand;
    float err = 0.0f;
    for (int j = 0; j < n; ++j) {
      const float r = dec(enc(x[j] * id)) * cand - x[j];
      err += r * r;
    }
    if (err < best_err) { best_err = err; best_d = cand; }
  }
  return best_d;
}

static void quantize_row_fp4(const float* x, block_fp4* y, int64_t n, bool search) {
  for (int64_t i = 0; i < n / QK; ++i) {
    const float d = search_scale(x + i * QK, QK, 6.0f, search, encode_e2m1,
                                 [](uint8_t c) { return E2M1_TABLE[c]; });
    const float id = d ? 1.0f / d : 0.0f;
    y[i].d = fp32_to_fp16(d);
    for (int j = 0; j < QK / 2; ++j) {
      const uint8_t q0 = encode_e2m1(x[i * QK + j] * id);
      const uint8_t q1 = encode_e2m1(x[i * QK + QK / 2 + j] * id);
      y[i].qs[j] = q0 | (q1 << 4);
    }
  }
}

static inline void pack_fp6(const uint8_t* codes, uint8_t* ql, uint8_t* qh, int nblk32) {
  // per 32 elements: 16 bytes of low nibbles (q4_0 order) + 8 bytes of 2 high bits
  for (int b = 0; b < nblk32; ++b) {
    const uint8_t* c = codes + b * 32;
    uint8_t* l = ql + b * 16;
    uint8_t* h = qh + b * 8;
    std::memset(h, 0, 8);
    for (int j = 0; j < 16; ++j) l[j] = (c[j] & 0x0F) | ((c[j + 16] & 0x0F) << 4);
    for (int j = 0; j < 32; ++j) h[j / 4] |= (c[j] >> 4) << (2 * (j % 4));
  }
}

static void quantize_row_fp6(const float* x, block_fp6* y, int64_t n, bool search) {
  for (int64_t i = 0; i < n / QK; ++i) {
    const float d = search_scale(x + i * QK, QK, 28.0f, search, encode_e3m2,
                                 [](uint8_t c) { return E3M2_TABLE[c]; });
    const float id = d ? 1.0f / d : 0.0f;
    y[i].d = fp32_to_fp16(d);
    uint8_t codes[QK];
    for (int j = 0; j < QK; ++j) codes[j] = encode_e3m2(x[i * QK + j] * id);
    pack_fp6(codes, y[i].ql, y[i].qh, 1);
  }
}

static void quantize_row_fp8(const float* x, block_fp8* y, int64_t n, bool e4m3, bool search) {
  for (int64_t i = 0; i < n / QK; ++i) {
    float d;
    if (e4m3)
      d = search_scale(x + i * QK, QK, 448.0f, search, encode_fp8_e4m3, [](uint8_t c) { return E4M3_TABLE[c]; });
    else
      d = search_scale(x + i * QK, QK, 57344.0f, search, encode_fp8_e5m2, [](uint8_t c) { return E5M2_TABLE[c]; });
    // keep the scale a normal fp16 so that value/d stays finite
    if (d != 0.0f && d < 6.2e-5f) d = 6.2e-5f;
    d = fp16_to_fp32(fp32_to_fp16(d));
    const float id = d ? 1.0f / d : 0.0f;
    y[i].d = fp32_to_fp16(d);
    for (int j = 0; j < QK; ++j)
      y[i].qs[j] = e4m3 ? encode_fp8_e4m3(x[i * QK + j] * id) : encode_fp8_e5m2(x[i * QK + j] * id);
  }
}

static void quantize_row_woq4(const float* x, block_woq4* y, int64_t n) {
  for (int64_t i = 0; i < n / QK_WOQ; ++i) {
    const float* xb = x + i * QK_WOQ;
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int j = 0; j < QK_WOQ; ++j) { mn = std::min(mn, xb[j]); mx = std::max(mx, xb[j]); }
    const float d = (mx - mn) / 15.0f;
    const float id = d ? 1.0f / d : 0.0f;
    y[i].d = fp32_to_fp16(d);
    y[i].m = fp32_to_fp16(mn);
    for (int j = 0; j < QK_WOQ / 2; ++j) {
      const uint8_t q0 = (uint8_t)std::min(15, (int)((xb[j] - mn) * id + 0.5f));
      const uint8_t q1 = (uint8_t)std::min(15, (int)((xb[j + QK_WOQ / 2] - mn) * id + 0.5f));
      y[i].qs[j] = q0 | (q1 << 4);
    }
  }
}

// ---------------------------------------------------------------------------
// k-quant helpers: weighted least-squares scale searches
// ---------------------------------------------------------------------------
static inline int nearest_int(float f) { return (int)std::nearbyint(f); }

// symmetric: x ~ scale * L, L in [-nmax, nmax-1]; returns scale, L stored offset by +nmax
static float make_sym_quants(int n, int nmax, const float* x, int8_t* L, const float* w) {
  float amax = 0.0f, mx = 0.0f;
  for (int i = 0; i < n; ++i) {
    if (std::fabs(x[i]) > amax) { amax = std::fabs(x[i]); mx = x[i]; }
  }
  if (amax < 1e-30f) {
    for (int i = 0; i < n; ++i) L[i] = (int8_t)nmax;
    return 0.0f;
  }
  float best_scale = 0.0f, best_err = FLT_MAX;
  for (int is = -9; is <= 9; ++is) {
    const float iscale = -(nmax + 0.1f * is) / mx;
    float sumlx = 0.0f, suml2 = 0.0f;
    for (int i = 0; i < n; ++i) {
      const int l = std::max(-nmax, std::min(nmax - 1, nearest_int(iscale * x[i])));
      const float wi = w ? w[i] : x[i] * x[i] + 1e-12f;
      sumlx += wi * x[i] * l;
      suml2 += wi * l * l;
    }
    if (suml2 <= 0.0f) continue;
    const float scale = sumlx / suml2;
    float err = 0.0f;
    for (int i = 0; i < n; ++i) {
      const int l = std::max(-nmax, std::min(nmax - 1, nearest_int(iscale * x[i])));
      const float wi = w ? w[i] : x[i] * x[i] + 1e-12f;
      const float r = scale * l - x[i];
      err += wi * r * r;
    }
    if (err < best_err) { best_err = err; best_scale = scale; }
  }
  if (best_scale == 0.0f) best_scale = mx / -nmax;
  const float id = 1.0f / best_scale;
  for (int i = 0; i < n; ++i)
    L[i] = (int8_t)(nmax + std::max(-nmax, std::min(nmax - 1, nearest_int(id * x[i]))));
  return best_scale;
}

// asymmetric: x ~ scale * L - min, L in [0, nmax], min >= 0; returns scale, writes *the_min
static float make_asym_quants(int n, int nmax, const float* x, const float* w, uint8_t* L, float* the_min) {
  float mn = x[0], mx = x[0];
  for (int i = 1; i < n; ++i) { mn = std::min(mn, x[i]); mx = std::max(mx, x[i]); }
  if (mn > 0.0f) mn = 0.0f;
  if (mx <= mn) {
    for (int i = 0; i < n; ++i) L[i] = 0;
    *the_min = -mn;
    return 0.0f;
  }
  float best_scale = (mx - mn) / nmax, best_min = mn, best_err = FLT_MAX;
  std::vector<uint8_t> La(n);
  for (int is = 0; is <= 20; ++is) {
    const float iscale = (-1.0f + 0.1f * is + nmax) / (mx - mn);
    float sw = 0, sx = 0, sl = 0, sl2 = 0, sxl = 0;
    for (int i = 0; i < n; ++i) {
      const int l = std::max(0, std::min(nmax, nearest_int(iscale * (x[i] - mn))));
      La[i] = (uint8_t)l;
      const float wi = w ? w[i] : 1.0f;
      sw += wi; sx += wi * x[i]; sl += wi * l; sl2 += wi * l * l; sxl += wi * x[i] * l;
    }
    const float D = sw * sl2 - sl * sl;
    if (D <= 0.0f) continue;
    float sc = (sw * sxl - sx * sl) / D;
    float m = (sl2 * sx - sl * sxl) / D;
    if (m > 0.0f) { m = 0.0f; sc = sl2 > 0 ? sxl / sl2 : sc; }
    if (sc <= 0.0f) continue;
    float err = 0.0f;
    for (int i = 0; i < n; ++i) {
      const float wi = w ? w[i] : 1.0f;
      const float r = sc * La[i] + m - x[i];
      err += wi * r * r;
    }
    if (err < best_err) { best_err = err; best_scale = sc; best_min = m; }
  }
  const float id = 1.0f / best_scale;
  for (int i = 0; i < n; ++i)
    L[i] = (uint8_t)std::max(0, std::min(nmax, nearest_int(id * (x[i] - best_min))));
  *the_min = -best_min;
  return best_scale;
}

static inline void get_scale_min_k4(int j, const uint8_t* q, uint8_t* d, uint8_t* m) {
  if (j < 4) {
    *d = q[j] & 63;
    *m = q[j + 4] & 63;
  } else {
    *d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4);
    *m = (q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4);
  }
}

static void pack_scales_k4(const uint8_t* ls, const uint8_t* lm, uint8_t* scales) {
  std::memset(scales, 0, 12);
  for (int j = 0; j < 8; ++j) {
    if (j < 4) {
      scales[j] |= ls[j];
      scales[j + 4] |= lm[j];
    } else {
      scales[j + 4] = (ls[j] & 0xF) | ((lm[j] & 0xF) << 4);
      scales[j - 4] |= (ls[j] >> 4) << 6;
      scales[j - 0] |= (lm[j] >> 4) << 6;
    }
  }
}

// shared by q4_k (nmax 15) and q5_k (nmax 31): fills L (0..nmax), scales[12], d, dmin
static void quantize_superblock_k45(const float* x, const float* w, int nmax, uint8_t* L, uint8_t* scales12,
                                    fp16_t* d_out, fp16_t* dmin_out) {
  float scales[8], mins[8];
  float max_scale = 0, max_min = 0;
  float wl[32];
  for (int j = 0; j < 8; ++j) {
    const float* xb = x + 32 * j;
    const float* wb = nullptr;
    if (w) { wb = w + 32 * j; }
    else {  // default weighting favours large magnitudes a little, like ggml's av_x + |x|
      float sum2 = 0;
      for (int l = 0; l < 32; ++l) sum2 += xb[l] * xb[l];
      const float av = std::sqrt(sum2 / 32);
      for (int l = 0; l < 32; ++l) wl[l] = av + std::fabs(xb[l]);
      wb = wl;
    }
    scales[j] = make_asym_quants(32, nmax, xb, wb, L + 32 * j, &mins[j]);
    max_scale = std::max(max_scale, scales[j]);
    max_min = std::max(max_min, mins[j]);
  }
  const float inv_scale = max_scale > 0 ? 63.0f / max_scale : 0.0f;
  const float inv_min = max_min > 0 ? 63.0f / max_min : 0.0f;
  uint8_t ls[8], lm[8];
  for (int j = 0; j < 8; ++j) {
    ls[j] = (uint8_t)std::min(63, nearest_int(inv_scale * scales[j]));
    lm[j] = (uint8_t)std::min(63, nearest_int(inv_min * mins[j]));
  }
  pack_scales_k4(ls, lm, scales12);
  *d_out = fp32_to_fp16(max_scale / 63.0f);
  *dmin_out = fp32_to_fp16(max_min / 63.0f);
  const float d = fp16_to_fp32(*d_out), dmin = fp16_to_fp32(*dmin_out);
  for (int j = 0; j < 8; ++j) {
    uint8_t sc, m;
    get_scale_min_k4(j, scales12, &sc, &m);
    const float dl = d * sc;
    if (dl == 0.0f) { for (int l = 0; l < 32; ++l) L[32 * j + l] = 0; continue; }
    const float ml = dmin * m;
    for (int l = 0; l < 32; ++l)
      L[32 * j + l] = (uint8_t)std::max(0, std::min(nmax, nearest_int((x[32 * j + l] + ml) / dl)));
  }
}

static void quantize_row_q4_K(const float* x, block_q4_K* y, int64_t n, const float* w) {
  uint8_t L[QK_K];
  for (int64_t i = 0; i < n / QK_K; ++i) {
    quantize_superblock_k45(x + i * QK_K, w ? w + (i * QK_K) : nullptr, 15, L, y[i].scales, &y[i].d, &y[i].dmin);
    uint8_t* q = y[i].qs;
    for (int j = 0; j < QK_K; j += 64) {
      for (int l = 0; l < 32; ++l) q[l] = L[j + l] | (L[j + l + 32] << 4);
      q += 32;
    }
  }
}

static void quantize_row_q5_K(const float* x, block_q5_K* y, int64_t n, const float* w) {
  uint8_t L[QK_K];
  for (int64_t i = 0; i < n / QK_K; ++i) {
    quantize_superblock_k45(x + i * QK_K, w ? w + (i * QK_K) : nullptr, 31, L, y[i].scales, &y[i].d, &y[i].dmin);
    uint8_t* qh = y[i].qh;
    uint8_t* ql = y[i].qs;
    std::memset(qh, 0, QK_K / 8);
    uint8_t m1 = 1, m2 = 2;
    for (int j = 0; j < QK_K; j += 64) {
      for (int l = 0; l < 32; ++l) {
        int l1 = L[j + l];
        if (l1 > 15) { l1 -= 16; qh[l] |= m1; }
        int l2 = L[j + l + 32];
        if (l2 > 15) { l2 -= 16; qh[l] |= m2; }
        ql[l] = (uint8_t)(l1 | (l2 << 4));
      }
      m1 <<= 2; m2 <<= 2;
      ql += 32;
    }
  }
}

static void quantize_row_q6_K(const float* x, block_q6_K* y, int64_t n, const float* w) {
  int8_t L[QK_K];
  float scales[QK_K / 16];
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const float* xb = x + i * QK_K;
    float max_scale = 0, max_abs = 0;
    for (int ib = 0; ib < QK_K / 16; ++ib) {
      const float s = make_sym_quants(16, 32, xb + 16 * ib, L + 16 * ib, w ? w + i * QK_K + 16 * ib : nullptr);
      scales[ib] = s;
      if (std::fabs(s) > max_abs) { max_abs = std::fabs(s); max_scale = s; }
    }
    if (max_abs < 1e-30f) {
      std::memset(&y[i], 0, sizeof(block_q6_K));
      continue;
    }
    const float iscale = -128.0f / max_scale;
    y[i].d = fp32_to_fp16(1.0f / iscale);
    for (int ib = 0; ib < QK_K / 16; ++ib)
      y[i].scales[ib] = (int8_t)std::min(127, std::max(-128, nearest_int(iscale * scales[ib])));
    const float d = fp16_to_fp32(y[i].d);
    for (int j = 0; j < QK_K / 16; ++j) {
      const float dl = d * y[i].scales[j];
      for (int ii = 0; ii < 16; ++ii) {
        int l = dl != 0.0f ? nearest_int(xb[16 * j + ii] / dl) : 0;
        l = std::max(-32, std::min(31, l));
        L[16 * j + ii] = (int8_t)(l + 32);
      }
    }
    uint8_t* ql = y[i].ql;
    uint8_t* qh = y[i].qh;
    for (int j = 0; j < QK_K; j += 128) {
      for (int l = 0; l < 32; ++l) {
        const uint8_t q1 = L[j + l + 0] & 0xF, q2 = L[j + l + 32] & 0xF;
        const uint8_t q3 = L[j + l + 64] & 0xF, q4 = L[j + l + 96] & 0xF;
        ql[l + 0] = q1 | (q3 << 4);
        ql[l + 32] = q2 | (q4 << 4);
        qh[l] = (L[j + l] >> 4) | ((L[j + l + 32] >> 4) << 2) | ((L[j + l + 64] >> 4) << 4) | ((L[j + l + 96] >> 4) << 6);
      }
      ql += 64;
      qh += 32;
    }
  }
}

static void quantize_row_q2_K(const float* x, block_q2_K* y, int64_t n, const float* w) {
  uint8_t L[QK_K];
  float mins[QK_K / 16], scales[QK_K / 16];
  float wl[16];
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const float* xb = x + i * QK_K;
    float max_scale = 0, max_min = 0;
    for (int j = 0; j < QK_K / 16; ++j) {
      const float* wb;
      if (w) wb = w + i * QK_K + 16 * j;
      else { for (int l = 0; l < 16; ++l) wl[l] = std::fabs(xb[16 * j + l]) + 1e-8f; wb = wl; }
      scales[j] = make_asym_quants(16, 3, xb + 16 * j, wb, L + 16 * j, &mins[j]);
      max_scale = std::max(max_scale, scales[j]);
      max_min = std::max(max_min, mins[j]);
    }
    if (max_scale > 0) {
      const float iscale = 15.0f / max_scale;
      for (int j = 0; j < QK_K / 16; ++j) y[i].scales[j] = (uint8_t)std::min(15, nearest_int(iscale * scales[j]));
      y[i].d = fp32_to_fp16(max_scale / 15.0f);
    } else {
      for (int j = 0; j < QK_K / 16; ++j) y[i].scales[j] = 0;
      y[i].d = fp32_to_fp16(0.0f);
    }
    if (max_min > 0) {
      const float iscale = 15.0f / max_min;
      for (int j = 0; j < QK_K / 16; ++j) y[i].scales[j] |= (uint8_t)(std::min(15, nearest_int(iscale * mins[j])) << 4);
      y[i].dmin = fp32_to_fp16(max_min / 15.0f);
    } else {
      y[i].dmin = fp32_to_fp16(0.0f);
    }
    const float d = fp16_to_fp32(y[i].d), dmin = fp16_to_fp32(y[i].dmin);
    for (int j = 0; j < QK_K / 16; ++j) {
      const float dl = d * (y[i].scales[j] & 0xF);
      if (dl == 0.0f) { for (int l = 0; l < 16; ++l) L[16 * j + l] = 0; continue; }
      const float ml = dmin * (y[i].scales[j] >> 4);
      for (int ii = 0; ii < 16; ++ii)
        L[16 * j + ii] = (uint8_t)std::max(0, std::min(3, nearest_int((xb[16 * j + ii] + ml) / dl)));
    }
    for (int j = 0; j < QK_K; j += 128)
      for (int l = 0; l < 32; ++l)
        y[i].qs[j / 4 + l] = L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6);
  }
}

static void quantize_row_fp6_K(const float* x, block_fp6_K* y, int64_t n, const float* w) {
  (void)w;
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const float* xb = x + i * QK_K;
    float sub[8], smax = 0.0f;
    for (int j = 0; j < 8; ++j) {
      float amax = 0.0f;
      for (int l = 0; l < 32; ++l) amax = std::max(amax, std::fabs(xb[32 * j + l]));
      sub[j] = amax / 28.0f;
      smax = std::max(smax, sub[j]);
    }
    const float d = smax / 127.0f;
    y[i].d = fp32_to_fp16(d);
    const float dq = fp16_to_fp32(y[i].d);
    uint8_t codes[QK_K];
    for (int j = 0; j < 8; ++j) {
      // round the sub-scale up so that the sub-block max still fits the e3m2 range
      int s = dq > 0 ? (int)std::ceil(sub[j] / dq - 1e-6f) : 0;
      s = std::max(0, std::min(127, s));
      y[i].scales[j] = (int8_t)s;
      const float dl = dq * s;
      const float id = dl ? 1.0f / dl : 0.0f;
      for (int l = 0; l < 32; ++l) codes[32 * j + l] = encode_e3m2(xb[32 * j + l] * id);
    }
    pack_fp6(codes, y[i].ql, y[i].qh, 8);
  }
}

void quantize_row(int qtype, const float* x, void* y, int64_t n, const float* imatrix, bool scale_search) {
  init_tables();
  switch (qtype) {
    case Q_SYM_INT4: quantize_row_q4_0(x, (block_q4_0*)y, n); break;
    case Q_ASYM_INT4: quantize_row_q4_1(x, (block_q4_1*)y, n); break;
    case Q_SYM_INT5: quantize_row_q5_0(x, (block_q5_0*)y, n); break;
    case Q_ASYM_INT5: quantize_row_q5_1(x, (block_q5_1*)y, n); break;
    case Q_SYM_INT8: quantize_row_q8_0(x, (block_q8_0*)y, n); break;
    case Q_NF4: quantize_row_nf4(x, (block_nf4*)y, n); break;
    case Q_NF3: quantize_row_nf3(x, (block_nf3*)y, n); break;
    case Q_FP4: quantize_row_fp4(x, (block_fp4*)y, n, scale_search); break;
    case Q_FP6: quantize_row_fp6(x, (block_fp6*)y, n, scale_search); break;
    case Q_FP8_E4M3: quantize_row_fp8(x, (block_fp8*)y, n, true, scale_search); break;
    case Q_FP8_E5M2: quantize_row_fp8(x, (block_fp8*)y, n, false, scale_search); break;
    case Q_WOQ_INT4: quantize_row_woq4(x, (block_woq4*)y, n); break;
    case Q_Q2_K: quantize_row_q2_K(x, (block_q2_K*)y, n, imatrix); break;
    case Q_Q4_K: quantize_row_q4_K(x, (block_q4_K*)y, n, imatrix); break;
    case Q_Q5_K: quantize_row_q5_K(x, (block_q5_K*)y, n, imatrix); break;
    case Q_Q6_K: quantize_row_q6_K(x, (block_q6_K*)y, n, imatrix); break;
    case Q_FP6_K: quantize_row_fp6_K(x, (block_fp6_K*)y, n, imatrix); break;
    case Q_IQ2_XXS: quantize_row_iq2_xxs(x, (block_iq2_xxs*)y, n, imatrix); break;
    case Q_IQ2_XS: quantize_row_iq2_xs(x, (block_iq2_xs*)y, n, imatrix); break;
    case Q_IQ1_S: case Q_IQ1_M: quantize_row_iq1_s(x, (block_iq1_s*)y, n, imatrix); break;
    default: break;
  }
}

}  // namespace bigdl
