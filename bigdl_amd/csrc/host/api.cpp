// C ABI of the host library (loaded with ctypes from bigdl_amd/ops/host.py).
//
// Mirrors the entry points the reference binds from its closed CPU library
// (python/llm/src/ipex_llm/ggml/model/llama/llama_cpp.py:946-1185): tensor quantize
// (plain / importance-weighted / per-channel RTN), dequantize, block geometry queries,
// host<->device re-layout, quantized matmul, MoE token bucketing.
#include "qformats.h"
#include <immintrin.h>
#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <vector>
#include <omp.h>

namespace bigdl {
int64_t device_nbytes(int qtype, int64_t N, int64_t K);
int relayout_to_device(int qtype, const uint8_t* src, uint8_t* dst, int64_t N, int64_t K);
int relayout_to_host(int qtype, const uint8_t* src, uint8_t* dst, int64_t N, int64_t K);
}

using namespace bigdl;

extern "C" {

int bigdl_qk_size(int qtype) { return qk_size(qtype); }
int bigdl_type_size(int qtype) { return type_size(qtype); }

// src: [nrow, n_per_row] fp32 ; dst: nrow * n_per_row / qk * type_size bytes
// imatrix: nullptr or [n_per_row] importance weights
int bigdl_quantize(const float* src, void* dst, int qtype, int64_t nrow, int64_t n_per_row,
                   const float* imatrix, int scale_search) {
  const int qk = qk_size(qtype), ts = type_size(qtype);
  if (qk <= 1 || n_per_row % qk != 0) return -1;
  init_tables();
  const int64_t row_bytes = n_per_row / qk * ts;
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t r = 0; r < nrow; ++r)
    quantize_row(qtype, src + r * n_per_row, (uint8_t*)dst + r * row_bytes, n_per_row, imatrix, scale_search != 0);
  return 0;
}

int bigdl_dequantize(const void* src, float* dst, int qtype, int64_t nrow, int64_t n_per_row) {
  const int qk = qk_size(qtype), ts = type_size(qtype);
  if (qk <= 1 || n_per_row % qk != 0) return -1;
  init_tables();
  const int64_t row_bytes = n_per_row / qk * ts;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrow; ++r)
    dequantize_row(qtype, (const uint8_t*)src + r * row_bytes, dst + r * n_per_row, n_per_row);
  return 0;
}

// Per-output-channel round-to-nearest formats (qtypes 31/32/33): codes and scales are
// separate tensors.  sym_int4_rtn / asym_int4_rtn pack two codes per byte
// (element 2j in the low nibble); sym_int8_rtn stores int8.  scale: [nrow] (sym) or
// [2, nrow] = scales then zero-points*scale offsets (asym).
int bigdl_quantize_rtn(const float* src, void* dst, float* scale, int qtype, int64_t nrow, int64_t n_per_row,
                       const float* imatrix) {
  (void)imatrix;
  if (qtype != Q_SYM_INT4_RTN && qtype != Q_SYM_INT8_RTN && qtype != Q_ASYM_INT4_RTN) return -1;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrow; ++r) {
    const float* x = src + r * n_per_row;
    float mn = FLT_MAX, mx = -FLT_MAX, amax = 0.0f;
    for (int64_t k = 0; k < n_per_row; ++k) { mn = std::min(mn, x[k]); mx = std::max(mx, x[k]); amax = std::max(amax, std::fabs(x[k])); }
    if (qtype == Q_SYM_INT8_RTN) {
      const float d = amax / 127.0f, id = d ? 1.0f / d : 0.0f;
      scale[r] = d;
      int8_t* q = (int8_t*)dst + r * n_per_row;
      for (int64_t k = 0; k < n_per_row; ++k) q[k] = (int8_t)std::nearbyint(x[k] * id);
    } else if (qtype == Q_SYM_INT4_RTN) {
      const float d = amax / 7.0f, id = d ? 1.0f / d : 0.0f;
      scale[r] = d;
      uint8_t* q = (uint8_t*)dst + r * n_per_row / 2;
      for (int64_t k = 0; k < n_per_row; k += 2) {
        const int a = std::max(-8, std::min(7, (int)std::nearbyint(x[k] * id))) + 8;
        const int b = std::max(-8, std::min(7, (int)std::nearbyint(x[k + 1] * id))) + 8;
        q[k / 2] = (uint8_t)(a | (b << 4));
      }
    } else {
      const float d = (mx - mn) / 15.0f, id = d ? 1.0f / d : 0.0f;
      scale[r] = d;
      scale[nrow + r] = mn;
      uint8_t* q = (uint8_t*)dst + r * n_per_row / 2;
      for (int64_t k = 0; k < n_per_row; k += 2) {
        const int a = std::max(0, std::min(15, (int)std::nearbyint((x[k] - mn) * id)));
        const int b = std::max(0, std::min(15, (int)std::nearbyint((x[k + 1] - mn) * id)));
        q[k / 2] = (uint8_t)(a | (b << 4));
      }
    }
  }
  return 0;
}

int bigdl_dequantize_rtn(const void* src, const float* scale, float* dst, int qtype, int64_t nrow, int64_t n_per_row) {
  if (qtype != Q_SYM_INT4_RTN && qtype != Q_SYM_INT8_RTN && qtype != Q_ASYM_INT4_RTN) return -1;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrow; ++r) {
    float* y = dst + r * n_per_row;
    if (qtype == Q_SYM_INT8_RTN) {
      const int8_t* q = (const int8_t*)src + r * n_per_row;
      for (int64_t k = 0; k < n_per_row; ++k) y[k] = q[k] * scale[r];
    } else {
      const uint8_t* q = (const uint8_t*)src + r * n_per_row / 2;
      const bool asym = qtype == Q_ASYM_INT4_RTN;
      const float off = asym ? scale[nrow + r] : -8.0f * scale[r];
      for (int64_t k = 0; k < n_per_row; k += 2) {
        y[k] = (q[k / 2] & 15) * scale[r] + off;
        y[k + 1] = (q[k / 2] >> 4) * scale[r] + off;
      }
    }
  }
  return 0;
}

int64_t bigdl_device_nbytes(int qtype, int64_t N, int64_t K) { return device_nbytes(qtype, N, K); }
int bigdl_relayout_to_device(int qtype, const void* src, void* dst, int64_t N, int64_t K) {
  return relayout_to_device(qtype, (const uint8_t*)src, (uint8_t*)dst, N, K);
}
int bigdl_relayout_to_host(int qtype, const void* src, void* dst, int64_t N, int64_t K) {
  return relayout_to_host(qtype, (const uint8_t*)src, (uint8_t*)dst, N, K);
}

// y[M, N] = x[M, K] . dequant(w)[N, K]^T   (fp32 activations, host block layout)
// Each thread owns weight rows; a row is read exactly once per call and reused for all M activation rows.  AVX2 + FMA
// (the library is built for x86-64-v3):
//  * q4_0 / q4_1 / q8_0 (sym_int4, asym_int4, sym_int8: what CPU inference normally loads) never materialise the row:
//    a block's 32 codes are widened to fp32 in registers, multiplied with the matching 32 activations of up to 4 rows
//    at a time, and the block's partial sum is scaled by d (and the activations' block sum by m / -8 d) once per block;
//  * every other format dequantizes the row into a scratch line (its own block decoder) and takes 8-wide dot products.
}  // extern "C"

static inline float hsum256(__m256 v) {
  __m128 lo = _mm_add_ps(_mm256_castps256_ps128(v), _mm256_extractf128_ps(v, 1));
  lo = _mm_add_ps(lo, _mm_movehl_ps(lo, lo));
  lo = _mm_add_ss(lo, _mm_shuffle_ps(lo, lo, 1));
  return _mm_cvtss_f32(lo);
}

static inline float dot_f32(const float* a, const float* b, int64_t K) {
  __m256 s0 = _mm256_setzero_ps(), s1 = s0, s2 = s0, s3 = s0;
  int64_t k = 0;
  for (; k + 32 <= K; k += 32) {
    s0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k), _mm256_loadu_ps(b + k), s0);
    s1 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k + 8), _mm256_loadu_ps(b + k + 8), s1);
    s2 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k + 16), _mm256_loadu_ps(b + k + 16), s2);
    s3 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k + 24), _mm256_loadu_ps(b + k + 24), s3);
  }
  for (; k + 8 <= K; k += 8) s0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k), _mm256_loadu_ps(b + k), s0);
  float r = hsum256(_mm256_add_ps(_mm256_add_ps(s0, s1), _mm256_add_ps(s2, s3)));
  for (; k < K; ++k) r += a[k] * b[k];
  return r;
}

// the 32 codes of one block as four fp32 vectors: elements 0-7, 8-15, 16-23, 24-31
static inline void widen_nibbles(const uint8_t* qs, __m256 (&c)[4]) {
  const __m128i b = _mm_loadu_si128((const __m128i*)qs);
  const __m128i lo = _mm_and_si128(b, _mm_set1_epi8(0x0F));                       // elements 0-15
  const __m128i hi = _mm_and_si128(_mm_srli_epi16(b, 4), _mm_set1_epi8(0x0F));    // elements 16-31
  c[0] = _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(lo));
  c[1] = _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(_mm_srli_si128(lo, 8)));
  c[2] = _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(hi));
  c[3] = _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(_mm_srli_si128(hi, 8)));
}
static inline void widen_bytes(const int8_t* qs, __m256 (&c)[4]) {
  for (int i = 0; i < 4; ++i) c[i] = _mm256_cvtepi32_ps(_mm256_cvtepi8_epi32(_mm_loadl_epi64((const __m128i*)(qs + 8 * i))));
}

// KIND 0: q4_0 ((q - 8) d), 1: q4_1 (q d + m), 2: q8_0 (q d).  R activation rows at once.
template <int KIND, int R>
static inline void block_row_dot(const uint8_t* wrow, const float* x, int64_t ldx, int64_t K, float* out, int64_t ldo) {
  constexpr int BS = KIND == 0 ? sizeof(block_q4_0) : (KIND == 1 ? sizeof(block_q4_1) : sizeof(block_q8_0));
  float acc[R];
  for (int r = 0; r < R; ++r) acc[r] = 0.0f;
  for (int64_t b = 0; b < K / QK; ++b) {
    const uint8_t* blk = wrow + b * BS;
    __m256 c[4];
    float d, m = 0.0f;
    if constexpr (KIND == 0) {
      d = fp16_to_fp32(((const block_q4_0*)blk)->d);
      widen_nibbles(((const block_q4_0*)blk)->qs, c);
      m = -8.0f * d;
    } else if constexpr (KIND == 1) {
      d = fp16_to_fp32(((const block_q4_1*)blk)->d);
      m = fp16_to_fp32(((const block_q4_1*)blk)->m);
      widen_nibbles(((const block_q4_1*)blk)->qs, c);
    } else {
      d = fp16_to_fp32(((const block_q8_0*)blk)->d);
      widen_bytes(((const block_q8_0*)blk)->qs, c);
    }
    for (int r = 0; r < R; ++r) {
      const float* xr = x + r * ldx + b * QK;
      const __m256 x0 = _mm256_loadu_ps(xr), x1 = _mm256_loadu_ps(xr + 8), x2 = _mm256_loadu_ps(xr + 16), x3 = _mm256_loadu_ps(xr + 24);
      const __m256 p = _mm256_fmadd_ps(x0, c[0], _mm256_fmadd_ps(x1, c[1], _mm256_fmadd_ps(x2, c[2], _mm256_mul_ps(x3, c[3]))));
      float v = d * hsum256(p);
      if constexpr (KIND != 2) v += m * hsum256(_mm256_add_ps(_mm256_add_ps(x0, x1), _mm256_add_ps(x2, x3)));
      acc[r] += v;
    }
  }
  for (int r = 0; r < R; ++r) out[r * ldo] = acc[r];
}

template <int KIND>
static void matmul_blocks(const float* x, const uint8_t* w, int64_t M, int64_t N, int64_t K, float* y) {
  constexpr int BS = KIND == 0 ? sizeof(block_q4_0) : (KIND == 1 ? sizeof(block_q4_1) : sizeof(block_q8_0));
  const int64_t row_bytes = K / QK * BS;
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < N; ++n) {
    const uint8_t* wrow = w + n * row_bytes;
    int64_t m = 0;
    for (; m + 4 <= M; m += 4) block_row_dot<KIND, 4>(wrow, x + m * K, K, K, y + m * N + n, N);
    for (; m < M; ++m) block_row_dot<KIND, 1>(wrow, x + m * K, K, K, y + m * N + n, N);
  }
}

extern "C" {

int bigdl_matmul_q(const float* x, const void* w, int qtype, int64_t M, int64_t N, int64_t K, float* y) {
  const int qk = qk_size(qtype), ts = type_size(qtype);
  if (qk <= 1 || K % qk != 0) return -1;
  init_tables();
  if (qtype == Q_SYM_INT4) { matmul_blocks<0>(x, (const uint8_t*)w, M, N, K, y); return 0; }
  if (qtype == Q_ASYM_INT4) { matmul_blocks<1>(x, (const uint8_t*)w, M, N, K, y); return 0; }
  if (qtype == Q_SYM_INT8) { matmul_blocks<2>(x, (const uint8_t*)w, M, N, K, y); return 0; }
  const int64_t row_bytes = K / qk * ts;
#pragma omp parallel
  {
    std::vector<float> line(K);
#pragma omp for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
      dequantize_row(qtype, (const uint8_t*)w + n * row_bytes, line.data(), K);
      for (int64_t m = 0; m < M; ++m) y[m * N + n] = dot_f32(x + m * K, line.data(), K);
    }
  }
  return 0;
}

// Gather + dequantize rows of a quantized embedding table: out[t, :] = dequant(w[ids[t], :])
int bigdl_dequantize_rows(const int64_t* ids, int64_t n_ids, const void* w, int qtype, int64_t dim, int64_t vocab,
                          float* out) {
  const int qk = qk_size(qtype), ts = type_size(qtype);
  if (qk <= 1 || dim % qk != 0) return -1;
  init_tables();
  const int64_t row_bytes = dim / qk * ts;
  int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
  for (int64_t t = 0; t < n_ids; ++t) {
    const int64_t id = ids[t];
    if (id < 0 || id >= vocab) { bad |= 1; continue; }
    dequantize_row(qtype, (const uint8_t*)w + id * row_bytes, out + t * dim, dim);
  }
  return bad ? -2 : 0;
}

// MoE bucketing (host variant): selected[T, top_k] expert ids ->
//   counts[E], offsets[E+1], and perm[T*top_k] listing flat (token*top_k + slot) indices
//   grouped by expert, ascending inside a group.
int bigdl_moe_indexes(const int32_t* selected, int64_t T, int64_t top_k, int64_t E, int64_t* offsets, int64_t* perm) {
  std::vector<int64_t> counts(E, 0);
  for (int64_t i = 0; i < T * top_k; ++i) {
    if (selected[i] < 0 || selected[i] >= E) return -1;
    ++counts[selected[i]];
  }
  offsets[0] = 0;
  for (int64_t e = 0; e < E; ++e) offsets[e + 1] = offsets[e] + counts[e];
  std::vector<int64_t> cur(offsets, offsets + E);
  for (int64_t i = 0; i < T * top_k; ++i) perm[cur[selected[i]]++] = i;
  return 0;
}

int bigdl_num_threads() { return omp_get_max_threads(); }

}  // extern "C"
