// C ABI of the host library (loaded with ctypes from bigdl_amd/ops/host.py).
//
// Mirrors the entry points the reference binds from its closed CPU library
// (python/llm/src/ipex_llm/ggml/model/llama/llama_cpp.py:946-1185): tensor quantize
// (plain / importance-weighted / per-channel RTN), dequantize, block geometry queries,
// host<->device re-layout, quantized matmul, MoE token bucketing.
#include "qformats.h"
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
// Each thread dequantizes one weight row into a scratch line and reuses it for all M
// activation rows, so the weight is read exactly once per call.
int bigdl_matmul_q(const float* x, const void* w, int qtype, int64_t M, int64_t N, int64_t K, float* y) {
  const int qk = qk_size(qtype), ts = type_size(qtype);
  if (qk <= 1 || K % qk != 0) return -1;
  init_tables();
  const int64_t row_bytes = K / qk * ts;
#pragma omp parallel
  {
    std::vector<float> line(K);
#pragma omp for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
      dequantize_row(qtype, (const uint8_t*)w + n * row_bytes, line.data(), K);
      for (int64_t m = 0; m < M; ++m) {
        const float* xr = x + m * K;
        float acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
        int64_t k = 0;
        for (; k + 4 <= K; k += 4) {
          acc0 += xr[k] * line[k];
          acc1 += xr[k + 1] * line[k + 1];
          acc2 += xr[k + 2] * line[k + 2];
          acc3 += xr[k + 3] * line[k + 3];
        }
        for (; k < K; ++k) acc0 += xr[k] * line[k];
        y[m * N + n] = (acc0 + acc1) + (acc2 + acc3);
      }
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
