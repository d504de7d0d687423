// Host block dequantizers: the oracle every device kernel is tested against.
// Counterpart of `ggml_dequantize` (reference binding:
// python/llm/src/ipex_llm/ggml/model/llama/llama_cpp.py:1068-1099); layouts follow
// the python reader in python/llm/src/ipex_llm/transformers/gguf/gguf.py:273-367.
#include "qformats.h"

namespace bigdl {

static void dq_q4_0(const block_q4_0* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n / QK; ++i) {
    const float d = fp16_to_fp32(x[i].d);
    for (int j = 0; j < QK / 2; ++j) {
      y[i * QK + j] = ((x[i].qs[j] & 0x0F) - 8) * d;
      y[i * QK + j + QK / 2] = ((x[i].qs[j] >> 4) - 8) * d;
    }
  }
}

static void dq_q4_1(const block_q4_1* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n / QK; ++i) {
    const float d = fp16_to_fp32(x[i].d), m = fp16_to_fp32(x[i].m);
    for (int j = 0; j < QK / 2; ++j) {
      y[i * QK + j] = (x[i].qs[j] & 0x0F) * d + m;
      y[i * QK + j + QK / 2] = (x[i].qs[j] >> 4) * d + m;
    }
  }
}

static void dq_q5_0(const block_q5_0* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n / QK; ++i) {
    const float d = fp16_to_fp32(x[i].d);
    uint32_t qh;
    std::memcpy(&qh, x[i].qh, 4);
    for (int j = 0; j < QK / 2; ++j) {
      const uint8_t h0 = ((qh >> (j + 0)) << 4) & 0x10;
      const uint8_t h1 = (qh >> (j + 12)) & 0x10;
      y[i * QK + j] = (((x[i].qs[j] & 0x0F) | h0) - 16) * d;
      y[i * QK + j + QK / 2] = (((x[i].qs[j] >> 4) | h1) - 16) * d;
    }
  }
}

static void dq_q5_1(const block_q5_1* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n / QK; ++i) {
    const float d = fp16_to_fp32(x[i].d), m = fp16_to_fp32(x[i].m);
    uint32_t qh;
    std::memcpy(&qh, x[i].qh, 4);
    for (int j = 0; j < QK / 2; ++j) {
      const uint8_t h0 = ((qh >> (j + 0)) << 4) & 0x10;
      const uint8_t h1 = (qh >> (j + 12)) & 0x10;
      y[i * QK + j] = ((x[i].qs[j] & 0x0F) | h0) * d + m;
      y[i * QK + j + QK / 2] = ((x[i].qs[j] >> 4) | h1) * d + m;
    }
  }
}

static void dq_q8_0(const block_q8_0* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n / QK; ++i) {
    const float d = fp16_to_fp32(x[i].d);
    for (int j = 0; j < QK; ++j) y[i * QK + j] = x[i].qs[j] * d;
  }
}

static void dq_nf4(const block_nf4* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n / QK_NF4; ++i) {
    const float d = fp16_to_fp32(x[i].d);
    for (int j = 0; j < QK_NF4 / 2; ++j) {
      y[i * QK_NF4 + j] = NF4_TABLE[x[i].qs[j] & 0x0F] * d;
      y[i * QK_NF4 + j + QK_NF4 / 2] = NF4_TABLE[x[i].qs[j] >> 4] * d;
    }
  }
}

static void dq_nf3(const block_nf3* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n / QK; ++i) {
    const float d = fp16_to_fp32(x[i].d);
    for (int j = 0; j < QK; ++j) {
      const int q = ((x[i].ql[j / 4] >> (2 * (j % 4))) & 3) | (((x[i].qh[j / 8] >> (j % 8)) & 1) << 2);
      y[i * QK + j] = NF3_TABLE[q] * d;
    }
  }
}

static void dq_fp4(const block_fp4* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n / QK; ++i) {
    const float d = fp16_to_fp32(x[i].d);
    for (int j = 0; j < QK / 2; ++j) {
      y[i * QK + j] = E2M1_TABLE[x[i].qs[j] & 0x0F] * d;
      y[i * QK + j + QK / 2] = E2M1_TABLE[x[i].qs[j] >> 4] * d;
    }
  }
}

static inline void unpack_fp6(const uint8_t* ql, const uint8_t* qh, uint8_t* codes, int nblk32) {
  for (int b = 0; b < nblk32; ++b) {
    const uint8_t* l = ql + b * 16;
    const uint8_t* h = qh + b * 8;
    uint8_t* c = codes + b * 32;
    for (int j = 0; j < 16; ++j) { c[j] = l[j] & 0x0F; c[j + 16] = l[j] >> 4; }
    for (int j = 0; j < 32; ++j) c[j] |= ((h[j / 4] >> (2 * (j % 4))) & 3) << 4;
  }
}

static void dq_fp6(const block_fp6* x, float* y, int64_t n) {
  uint8_t codes[QK];
  for (int64_t i = 0; i < n / QK; ++i) {
    const float d = fp16_to_fp32(x[i].d);
    unpack_fp6(x[i].ql, x[i].qh, codes, 1);
    for (int j = 0; j < QK; ++j) y[i * QK + j] = E3M2_TABLE[codes[j]] * d;
  }
}

static void dq_fp8(const block_fp8* x, float* y, int64_t n, const float* table) {
  for (int64_t i = 0; i < n / QK; ++i) {
    const float d = fp16_to_fp32(x[i].d);
    for (int j = 0; j < QK; ++j) y[i * QK + j] = table[x[i].qs[j]] * d;
  }
}

static void dq_woq4(const block_woq4* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n / QK_WOQ; ++i) {
    const float d = fp16_to_fp32(x[i].d), m = fp16_to_fp32(x[i].m);
    for (int j = 0; j < QK_WOQ / 2; ++j) {
      y[i * QK_WOQ + j] = (x[i].qs[j] & 0x0F) * d + m;
      y[i * QK_WOQ + j + QK_WOQ / 2] = (x[i].qs[j] >> 4) * d + m;
    }
  }
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

static void dq_q2_K(const block_q2_K* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const float d = fp16_to_fp32(x[i].d), mn = fp16_to_fp32(x[i].dmin);
    const uint8_t* q = x[i].qs;
    int is = 0;
    for (int nn = 0; nn < QK_K; nn += 128) {
      int shift = 0;
      for (int j = 0; j < 4; ++j) {
        uint8_t sc = x[i].scales[is++];
        float dl = d * (sc & 0xF), ml = mn * (sc >> 4);
        for (int l = 0; l < 16; ++l) *y++ = dl * ((q[l] >> shift) & 3) - ml;
        sc = x[i].scales[is++];
        dl = d * (sc & 0xF); ml = mn * (sc >> 4);
        for (int l = 0; l < 16; ++l) *y++ = dl * ((q[l + 16] >> shift) & 3) - ml;
        shift += 2;
      }
      q += 32;
    }
  }
}

static void dq_q4_K(const block_q4_K* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const uint8_t* q = x[i].qs;
    const float d = fp16_to_fp32(x[i].d), mn = fp16_to_fp32(x[i].dmin);
    int is = 0;
    uint8_t sc, m;
    for (int j = 0; j < QK_K; j += 64) {
      get_scale_min_k4(is + 0, x[i].scales, &sc, &m);
      const float d1 = d * sc, m1 = mn * m;
      get_scale_min_k4(is + 1, x[i].scales, &sc, &m);
      const float d2 = d * sc, m2 = mn * m;
      for (int l = 0; l < 32; ++l) *y++ = d1 * (q[l] & 0xF) - m1;
      for (int l = 0; l < 32; ++l) *y++ = d2 * (q[l] >> 4) - m2;
      q += 32;
      is += 2;
    }
  }
}

static void dq_q5_K(const block_q5_K* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const uint8_t* ql = x[i].qs;
    const uint8_t* qh = x[i].qh;
    const float d = fp16_to_fp32(x[i].d), mn = fp16_to_fp32(x[i].dmin);
    int is = 0;
    uint8_t sc, m, u1 = 1, u2 = 2;
    for (int j = 0; j < QK_K; j += 64) {
      get_scale_min_k4(is + 0, x[i].scales, &sc, &m);
      const float d1 = d * sc, m1 = mn * m;
      get_scale_min_k4(is + 1, x[i].scales, &sc, &m);
      const float d2 = d * sc, m2 = mn * m;
      for (int l = 0; l < 32; ++l) *y++ = d1 * ((ql[l] & 0xF) + (qh[l] & u1 ? 16 : 0)) - m1;
      for (int l = 0; l < 32; ++l) *y++ = d2 * ((ql[l] >> 4) + (qh[l] & u2 ? 16 : 0)) - m2;
      ql += 32;
      is += 2;
      u1 <<= 2;
      u2 <<= 2;
    }
  }
}

static void dq_q6_K(const block_q6_K* x, float* y, int64_t n) {
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const float d = fp16_to_fp32(x[i].d);
    const uint8_t* ql = x[i].ql;
    const uint8_t* qh = x[i].qh;
    const int8_t* sc = x[i].scales;
    for (int nn = 0; nn < QK_K; nn += 128) {
      for (int l = 0; l < 32; ++l) {
        const int is = l / 16;
        const int8_t q1 = (int8_t)((ql[l + 0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
        const int8_t q2 = (int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
        const int8_t q3 = (int8_t)((ql[l + 0] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
        const int8_t q4 = (int8_t)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
        y[l + 0] = d * sc[is + 0] * q1;
        y[l + 32] = d * sc[is + 2] * q2;
        y[l + 64] = d * sc[is + 4] * q3;
        y[l + 96] = d * sc[is + 6] * q4;
      }
      y += 128;
      ql += 64;
      qh += 32;
      sc += 8;
    }
  }
}

static void dq_fp6_K(const block_fp6_K* x, float* y, int64_t n) {
  uint8_t codes[QK_K];
  for (int64_t i = 0; i < n / QK_K; ++i) {
    const float d = fp16_to_fp32(x[i].d);
    unpack_fp6(x[i].ql, x[i].qh, codes, 8);
    for (int j = 0; j < 8; ++j) {
      const float dl = d * x[i].scales[j];
      for (int l = 0; l < 32; ++l) y[i * QK_K + 32 * j + l] = E3M2_TABLE[codes[32 * j + l]] * dl;
    }
  }
}

void dequantize_row(int qtype, const void* x, float* y, int64_t n) {
  init_tables();
  switch (qtype) {
    case Q_SYM_INT4: dq_q4_0((const block_q4_0*)x, y, n); break;
    case Q_ASYM_INT4: dq_q4_1((const block_q4_1*)x, y, n); break;
    case Q_SYM_INT5: dq_q5_0((const block_q5_0*)x, y, n); break;
    case Q_ASYM_INT5: dq_q5_1((const block_q5_1*)x, y, n); break;
    case Q_SYM_INT8: dq_q8_0((const block_q8_0*)x, y, n); break;
    case Q_NF4: dq_nf4((const block_nf4*)x, y, n); break;
    case Q_NF3: dq_nf3((const block_nf3*)x, y, n); break;
    case Q_FP4: dq_fp4((const block_fp4*)x, y, n); break;
    case Q_FP6: dq_fp6((const block_fp6*)x, y, n); break;
    case Q_FP8_E4M3: dq_fp8((const block_fp8*)x, y, n, E4M3_TABLE); break;
    case Q_FP8_E5M2: dq_fp8((const block_fp8*)x, y, n, E5M2_TABLE); break;
    case Q_WOQ_INT4: dq_woq4((const block_woq4*)x, y, n); break;
    case Q_Q2_K: dq_q2_K((const block_q2_K*)x, y, n); break;
    case Q_Q4_K: dq_q4_K((const block_q4_K*)x, y, n); break;
    case Q_Q5_K: dq_q5_K((const block_q5_K*)x, y, n); break;
    case Q_Q6_K: dq_q6_K((const block_q6_K*)x, y, n); break;
    case Q_FP6_K: dq_fp6_K((const block_fp6_K*)x, y, n); break;
    case Q_IQ2_XXS: dequantize_row_iq2_xxs((const block_iq2_xxs*)x, y, n); break;
    case Q_IQ2_XS: dequantize_row_iq2_xs((const block_iq2_xs*)x, y, n); break;
    case Q_IQ1_S: case Q_IQ1_M: dequantize_row_iq1_s((const block_iq1_s*)x, y, n); break;
    default: break;
  }
}

}  // namespace bigdl
