// Weight dequantisation on gfx950.
//  * T16 (tiled device layout) -> dense row-major [N, K]: the reference's xe_linear.dequant, used by the
//    QLoRA backward and by the prefill path (python/llm/src/ipex_llm/transformers/low_bit_linear.py:530-539).
//  * host-layout block rows (k-quants, nf3, fp6_k, and every simple format) -> dense rows, with an
//    optional row gather: the reference's xe_linear.dequantize_rows for low-bit embeddings
//    (python/llm/src/ipex_llm/transformers/embedding.py:220-222).
#include "common.h"

namespace bigdl {

template <int OUT>
__device__ __forceinline__ void store8_out(void* p, int64_t idx, const float* v) {
  if constexpr (OUT == OUT_F16) {
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
    *(half8*)((_Float16*)p + idx) = h;
  } else if constexpr (OUT == OUT_BF16) {
    uint4v u;
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = (uint32_t)f32_to_bf16_bits(v[2 * j]) | ((uint32_t)f32_to_bf16_bits(v[2 * j + 1]) << 16);
    *(uint4v*)((uint16_t*)p + idx) = u;
  } else {
    float4v a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *(float4v*)((float*)p + idx) = a;
    *(float4v*)((float*)p + idx + 4) = b;
  }
}

// one wavefront per 16x128 tile, 4 wavefronts per workgroup
template <int QT, int OUT>
__global__ __launch_bounds__(256) void dequant_t16_kernel(const void* w, void* out, int N, int K, int64_t ldo, int n_planes,
                                                          int row0) {
  __shared__ _Float16 nf4_lut[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int KT = (K + 127) >> 7, NT = (N + 15) >> 4;
  if constexpr (QT == Q_NF4) {
    if (threadIdx.x < 16) nf4_lut[threadIdx.x] = (_Float16)NF4_VALUES[threadIdx.x];
    __syncthreads();
  }
  // rows [row0, row0 + N) of a [n_planes, K] buffer (row0 % 16 == 0) -> out rows [0, N)
  const int64_t ltile = (int64_t)blockIdx.x * 4 + wave;
  if (ltile >= (int64_t)NT * KT) return;
  const int nt = (int)(ltile / KT), kt = (int)(ltile % KT);
  const int64_t tile = ltile + (int64_t)(row0 >> 4) * KT;
  const T16Planes p = t16_planes(w, QT, n_planes, K);
  const int r = lane & 15, g = lane >> 4;
  const int n = nt * 16 + r;
  constexpr int LO = (QT == Q_SYM_INT8 || QT == Q_FP8_E4M3 || QT == Q_FP8_E5M2) ? 8 : 4;
  uint4v lo0, lo1 = {0, 0, 0, 0};
  if constexpr (LO == 4) lo0 = *(const uint4v*)(p.lo + tile * 1024 + lane * 16);
  else { lo0 = *(const uint4v*)(p.lo + tile * 2048 + lane * 16); lo1 = *(const uint4v*)(p.lo + tile * 2048 + 1024 + lane * 16); }
  uint32_t hi0 = 0, hi1 = 0;
  if constexpr (QT == Q_SYM_INT5 || QT == Q_ASYM_INT5) hi0 = *(const uint32_t*)(p.hi + tile * 256 + lane * 4);
  if constexpr (QT == Q_FP6) { const uint2v h = *(const uint2v*)(p.hi + tile * 512 + lane * 8); hi0 = h.x; hi1 = h.y; }
  const half4 s4 = *(const half4*)(p.sc + tile * 64 + r * 4);
  half4 m4 = {0, 0, 0, 0};
  constexpr bool HASMIN = (QT == Q_ASYM_INT4 || QT == Q_ASYM_INT5 || QT == Q_WOQ_INT4);
  if constexpr (HASMIN) m4 = *(const half4*)(p.mn + tile * 64 + r * 4);
  if (n >= N) return;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k0 = kt * 128 + 32 * s + 8 * g;
    if (k0 >= K) continue;
    half8 f;
    float post = 1.0f;
    if constexpr (QT == Q_SYM_INT4) f = dec_int4(lo0[s], 8.0f);
    else if constexpr (QT == Q_ASYM_INT4 || QT == Q_WOQ_INT4) f = dec_int4(lo0[s], 0.0f);
    else if constexpr (QT == Q_FP4) { f = dec_fp4(lo0[s]); post = FP4_POST; }
    else if constexpr (QT == Q_NF4) f = dec_nf4(lo0[s], nf4_lut);
    else if constexpr (QT == Q_SYM_INT5) f = dec_int5(lo0[s], (hi0 >> (8 * s)) & 0xffu, 16.0f);
    else if constexpr (QT == Q_ASYM_INT5) f = dec_int5(lo0[s], (hi0 >> (8 * s)) & 0xffu, 0.0f);
    else if constexpr (QT == Q_FP6) { f = dec_fp6(lo0[s], ((s < 2 ? hi0 : hi1) >> (16 * (s & 1))) & 0xffffu); post = FP6_POST; }
    else {
      const uint4v q = s < 2 ? lo0 : lo1;
      const uint32_t w0 = q[(s & 1) * 2], w1 = q[(s & 1) * 2 + 1];
      if constexpr (QT == Q_SYM_INT8) f = dec_int8(w0, w1);
      else if constexpr (QT == Q_FP8_E5M2) f = dec_e5m2(w0, w1);
      else { f = dec_e4m3(w0, w1); post = E4M3_POST; }
    }
    const float d = (float)s4[s] * post, mn = (float)m4[s];
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)f[j] * d + mn;
    store8_out<OUT>(out, (int64_t)n * ldo + k0, v);
  }
}

// ---------------------------------------------------------------------------------------------
// host-layout block rows
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float ldh(const uint8_t* p) {
  union { uint16_t u; _Float16 h; } c;
  c.u = (uint16_t)p[0] | ((uint16_t)p[1] << 8);
  return (float)c.h;
}

__constant__ const float NF3_VALUES[8] = {-1.0f, -0.5350227355957031f, -0.2469314038753510f, 0.0f,
                                          0.1833375245332718f, 0.3819939494132996f, 0.6229856610298157f, 1.0f};
__constant__ const float E2M1_VALUES[16] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f,
                                            -0.0f, -0.5f, -1.0f, -1.5f, -2.0f, -3.0f, -4.0f, -6.0f};

__device__ __forceinline__ float e3m2_value(int c) {
  const int s = c >> 5, e = (c >> 2) & 7, m = c & 3;
  const float v = e == 0 ? ldexpf((float)m * 0.25f, -2) : ldexpf(1.0f + m * 0.25f, e - 3);
  return s ? -v : v;
}
__device__ __forceinline__ float e4m3_value(int c) {
  const int s = c >> 7, e = (c >> 3) & 15, m = c & 7;
  float v = e == 0 ? ldexpf((float)m * 0.125f, -6) : ldexpf(1.0f + m * 0.125f, e - 7);
  if ((c & 0x7f) == 0x7f) v = 0.0f;
  return s ? -v : v;
}
__device__ __forceinline__ float e5m2_value(int c) {
  union { uint16_t u; _Float16 h; } cv;
  cv.u = (uint16_t)c << 8;
  return ((c >> 2) & 31) == 31 ? 0.0f : (float)cv.h;
}

__device__ __forceinline__ void scale_min_k4(int j, const uint8_t* q, int* d, int* m) {
  if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
  else { *d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); *m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}

// value of element e (0..qk-1) of one host-layout block
__device__ float block_elem(int qt, const uint8_t* b, int e) {
  switch (qt) {
    case Q_SYM_INT4: { const int q = e < 16 ? (b[2 + e] & 15) : (b[2 + e - 16] >> 4); return (q - 8) * ldh(b); }
    case Q_ASYM_INT4: { const int q = e < 16 ? (b[4 + e] & 15) : (b[4 + e - 16] >> 4); return q * ldh(b) + ldh(b + 2); }
    case Q_FP4: { const int q = e < 16 ? (b[2 + e] & 15) : (b[2 + e - 16] >> 4); return E2M1_VALUES[q] * ldh(b); }
    case Q_SYM_INT5: {
      const uint32_t qh = b[2] | (b[3] << 8) | (b[4] << 16) | ((uint32_t)b[5] << 24);
      const int lo = e < 16 ? (b[6 + e] & 15) : (b[6 + e - 16] >> 4);
      return ((lo | (((qh >> e) & 1) << 4)) - 16) * ldh(b);
    }
    case Q_ASYM_INT5: {
      const uint32_t qh = b[4] | (b[5] << 8) | (b[6] << 16) | ((uint32_t)b[7] << 24);
      const int lo = e < 16 ? (b[8 + e] & 15) : (b[8 + e - 16] >> 4);
      return (lo | (((qh >> e) & 1) << 4)) * ldh(b) + ldh(b + 2);
    }
    case Q_SYM_INT8: return (float)(int8_t)b[2 + e] * ldh(b);
    case Q_FP8_E4M3: return e4m3_value(b[2 + e]) * ldh(b);
    case Q_FP8_E5M2: return e5m2_value(b[2 + e]) * ldh(b);
    case Q_NF4: { const int q = e < 32 ? (b[2 + e] & 15) : (b[2 + e - 32] >> 4); return NF4_VALUES[q] * ldh(b); }
    case Q_NF3: {
      const int q = ((b[2 + e / 4] >> (2 * (e % 4))) & 3) | (((b[10 + e / 8] >> (e % 8)) & 1) << 2);
      return NF3_VALUES[q] * ldh(b);
    }
    case Q_FP6: {
      const int lo = e < 16 ? (b[2 + e] & 15) : (b[2 + e - 16] >> 4);
      const int hi = (b[18 + e / 4] >> (2 * (e % 4))) & 3;
      return e3m2_value(lo | (hi << 4)) * ldh(b);
    }
    case Q_WOQ_INT4: { const int q = e < 64 ? (b[4 + e] & 15) : (b[4 + e - 64] >> 4); return q * ldh(b) + ldh(b + 2); }
    case Q_Q4_K: {
      const float d = ldh(b), mn = ldh(b + 2);
      const int sub = e / 32, l = e % 32;
      int sc, m;
      scale_min_k4(sub, b + 4, &sc, &m);
      const uint8_t qb = b[16 + (sub / 2) * 32 + l];
      return d * sc * ((sub & 1) ? (qb >> 4) : (qb & 15)) - mn * m;
    }
    case Q_Q5_K: {
      const float d = ldh(b), mn = ldh(b + 2);
      const int sub = e / 32, l = e % 32;
      int sc, m;
      scale_min_k4(sub, b + 4, &sc, &m);
      const uint8_t qb = b[48 + (sub / 2) * 32 + l];
      const int hi = (b[16 + l] >> sub) & 1;
      return d * sc * (((sub & 1) ? (qb >> 4) : (qb & 15)) + 16 * hi) - mn * m;
    }
    case Q_Q6_K: {
      const uint8_t* ql = b; const uint8_t* qh = b + 128; const int8_t* sc = (const int8_t*)(b + 192);
      const float d = ldh(b + 208);
      const int half = e / 128, ee = e % 128, quad = ee / 32, l = ee % 32;
      const uint8_t lb = ql[half * 64 + (quad & 1) * 32 + l];
      const int lo = (quad >= 2) ? (lb >> 4) : (lb & 15);
      const int hi = (qh[half * 32 + l] >> (2 * quad)) & 3;
      const int q = (lo | (hi << 4)) - 32;
      return d * sc[half * 8 + (l / 16) + 2 * quad] * q;
    }
    case Q_Q2_K: {
      const uint8_t* scales = b; const uint8_t* qs = b + 16;
      const float d = ldh(b + 80), mn = ldh(b + 82);
      const int half = e / 128, ee = e % 128, j = ee / 32, l = ee % 32;
      const uint8_t sc = scales[half * 8 + 2 * j + l / 16];
      const int q = (qs[half * 32 + l] >> (2 * j)) & 3;
      return d * (sc & 15) * q - mn * (sc >> 4);
    }
    case Q_FP6_K: {
      const float d = ldh(b);
      const int sub = e / 32, l = e % 32;
      const int8_t s = (int8_t)b[2 + sub];
      const uint8_t* ql = b + 10 + sub * 16;
      const uint8_t* qh = b + 10 + 128 + sub * 8;
      const int lo = l < 16 ? (ql[l] & 15) : (ql[l - 16] >> 4);
      const int hi = (qh[l / 4] >> (2 * (l % 4))) & 3;
      return e3m2_value(lo | (hi << 4)) * d * s;
    }
    default: return 0.0f;
  }
}

// out[t, :] = dequant(row ids[t]) (ids == null: identity).  One workgroup per output row.
template <int OUT>
__global__ __launch_bounds__(256) void dequant_rows_kernel(const uint8_t* w, const int64_t* ids, void* out, int64_t n_out,
                                                           int64_t n_rows, int K, int qt, int qk, int ts, int64_t ldo) {
  const int64_t t = blockIdx.x;
  int64_t row = ids ? ids[t] : t;
  if (row < 0 || row >= n_rows) row = 0;  // the host validates ids for eager calls; stay in bounds regardless
  const uint8_t* rp = w + row * (int64_t)(K / qk) * ts;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float v = block_elem(qt, rp + (int64_t)(k / qk) * ts, k % qk);
    store_out<OUT>(out, t * ldo + k, v);
  }
}

}  // namespace bigdl

using namespace bigdl;

template <int QT>
static int launch_dq(const void* w, void* out, int N, int K, int64_t ldo, int out_type, int n_planes, int row0, hipStream_t st) {
  const int64_t tiles = (int64_t)((N + 15) / 16) * ((K + 127) / 128);
  const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
  switch (out_type) {
    case OUT_F16: hipLaunchKernelGGL((dequant_t16_kernel<QT, OUT_F16>), grid, block, 0, st, w, out, N, K, ldo, n_planes, row0); break;
    case OUT_BF16: hipLaunchKernelGGL((dequant_t16_kernel<QT, OUT_BF16>), grid, block, 0, st, w, out, N, K, ldo, n_planes, row0); break;
    case OUT_F32: hipLaunchKernelGGL((dequant_t16_kernel<QT, OUT_F32>), grid, block, 0, st, w, out, N, K, ldo, n_planes, row0); break;
    default: return -3;
  }
  return HIP_CHECK_LAUNCH();
}

// rows [row0, row0 + N) of the T16 buffer of an [n_total, K] matrix -> dense out [N, ldo]
extern "C" int bigdl_dequant_t16(const void* w, void* out, int N, int K, int64_t ldo, int qtype, int out_type, int n_total,
                                 int row0, hipStream_t st) {
  qtype = t16_kernel_qtype(qtype);
  if (n_total <= 0) { n_total = N; row0 = 0; }
  if ((row0 & 15) || row0 < 0 || row0 + N > (n_total + 15) / 16 * 16) return -1;
  const int n_planes = n_total;
  if (N <= 0 || K <= 0 || (K & 31) || (ldo & 7) || ((uintptr_t)w & 15) || ((uintptr_t)out & 15)) return -1;
  switch (qtype) {
    case Q_SYM_INT4: return launch_dq<Q_SYM_INT4>(w, out, N, K, ldo, out_type, n_planes, row0, st);
    case Q_ASYM_INT4: return launch_dq<Q_ASYM_INT4>(w, out, N, K, ldo, out_type, n_planes, row0, st);
    case Q_WOQ_INT4: return launch_dq<Q_WOQ_INT4>(w, out, N, K, ldo, out_type, n_planes, row0, st);
    case Q_FP4: return launch_dq<Q_FP4>(w, out, N, K, ldo, out_type, n_planes, row0, st);
    case Q_NF4: return launch_dq<Q_NF4>(w, out, N, K, ldo, out_type, n_planes, row0, st);
    case Q_SYM_INT5: return launch_dq<Q_SYM_INT5>(w, out, N, K, ldo, out_type, n_planes, row0, st);
    case Q_ASYM_INT5: return launch_dq<Q_ASYM_INT5>(w, out, N, K, ldo, out_type, n_planes, row0, st);
    case Q_FP6: return launch_dq<Q_FP6>(w, out, N, K, ldo, out_type, n_planes, row0, st);
    case Q_SYM_INT8: return launch_dq<Q_SYM_INT8>(w, out, N, K, ldo, out_type, n_planes, row0, st);
    case Q_FP8_E4M3: return launch_dq<Q_FP8_E4M3>(w, out, N, K, ldo, out_type, n_planes, row0, st);
    case Q_FP8_E5M2: return launch_dq<Q_FP8_E5M2>(w, out, N, K, ldo, out_type, n_planes, row0, st);
    default: return -2;
  }
}

static int host_block_geometry(int qt, int* qk, int* ts) {
  switch (qt) {
    case Q_SYM_INT4: case Q_FP4: *qk = 32; *ts = 18; return 0;
    case Q_ASYM_INT4: *qk = 32; *ts = 20; return 0;
    case Q_SYM_INT5: *qk = 32; *ts = 22; return 0;
    case Q_ASYM_INT5: *qk = 32; *ts = 24; return 0;
    case Q_SYM_INT8: case Q_FP8_E4M3: case Q_FP8_E5M2: *qk = 32; *ts = 34; return 0;
    case Q_NF4: *qk = 64; *ts = 34; return 0;
    case Q_NF3: *qk = 32; *ts = 14; return 0;
    case Q_FP6: *qk = 32; *ts = 26; return 0;
    case Q_WOQ_INT4: *qk = 128; *ts = 68; return 0;
    case Q_Q2_K: *qk = 256; *ts = 84; return 0;
    case Q_Q4_K: *qk = 256; *ts = 144; return 0;
    case Q_Q5_K: *qk = 256; *ts = 176; return 0;
    case Q_Q6_K: *qk = 256; *ts = 210; return 0;
    case Q_FP6_K: *qk = 256; *ts = 202; return 0;
    default: return -2;
  }
}

// w: host-layout block rows [n_rows, K]; ids: optional gather indices [n_out] (device int64)
extern "C" int bigdl_dequant_rows(const void* w, const int64_t* ids, void* out, int64_t n_out, int64_t n_rows, int K,
                                  int qtype, int64_t ldo, int out_type, hipStream_t st) {
  int qk, ts;
  if (host_block_geometry(qtype, &qk, &ts) != 0) return -2;
  if (n_out <= 0 || K <= 0 || K % qk) return -1;
  const dim3 grid((unsigned)n_out), block(256);
  switch (out_type) {
    case OUT_F16: hipLaunchKernelGGL((dequant_rows_kernel<OUT_F16>), grid, block, 0, st, (const uint8_t*)w, ids, out, n_out, n_rows, K, qtype, qk, ts, ldo); break;
    case OUT_BF16: hipLaunchKernelGGL((dequant_rows_kernel<OUT_BF16>), grid, block, 0, st, (const uint8_t*)w, ids, out, n_out, n_rows, K, qtype, qk, ts, ldo); break;
    case OUT_F32: hipLaunchKernelGGL((dequant_rows_kernel<OUT_F32>), grid, block, 0, st, (const uint8_t*)w, ids, out, n_out, n_rows, K, qtype, qk, ts, ldo); break;
    default: return -3;
  }
  return HIP_CHECK_LAUNCH();
}
