// RWKV recurrences on gfx950 (reference: the closed xe_linear.rwkv_time_shift / rwkv_linear_attention_v4 / _v5,
// signatures python/llm/src/ipex_llm/transformers/xpu_ops.py:40-56, call sites models/rwkv4.py:60-96 and
// models/rwkv5.py:62-102).
//  * time_shift: out[m] = hidden * mix[m] + shifted * (1 - mix[m]) for the m stacked mixing vectors, one pass.
//  * v4 wkv: per (batch, channel) scalar recurrence with a running maximum for stability; one thread per channel,
//    consecutive threads = consecutive channels, so every step is one coalesced row read of k and v.
//  * v5 wkv: per (batch, head) matrix state [S, S] kept in registers: thread j owns column j of the state; r, k and
//    the decay / bonus vectors of the step are staged in LDS.
#include "common.h"

namespace bigdl {

enum RDType : int { RD_F16 = 0, RD_BF16 = 1, RD_F32 = 2 };

__device__ __forceinline__ float rd_load(const void* p, int64_t i, int dt) {
  if (dt == RD_F16) return (float)((const _Float16*)p)[i];
  if (dt == RD_BF16) return bf16_bits_to_f32(((const uint16_t*)p)[i]);
  return ((const float*)p)[i];
}
__device__ __forceinline__ void rd_store(void* p, int64_t i, float v, int dt) {
  if (dt == RD_F16) ((_Float16*)p)[i] = (_Float16)v;
  else if (dt == RD_BF16) ((uint16_t*)p)[i] = f32_to_bf16_bits(v);
  else ((float*)p)[i] = v;
}

__global__ __launch_bounds__(256) void rwkv_time_shift_kernel(const void* hidden, const void* shifted, const void* mix,
                                                              void* out, int64_t rows, int C, int n_mix, int dt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * C) return;
  const int c = (int)(i % C);
  const float h = rd_load(hidden, i, dt), s = rd_load(shifted, i, dt);
  for (int m = 0; m < n_mix; ++m) {
    const float w = rd_load(mix, (int64_t)m * C + c, dt);
    rd_store(out, (int64_t)m * rows * C + i, h * w + s * (1.0f - w), dt);
  }
}

// key / value / out: [B, T, C]; time_decay (already -exp(w)) and time_first: [C] fp32; states [B, C] fp32 in/out
__global__ __launch_bounds__(128) void rwkv4_wkv_kernel(const float* time_decay, const float* time_first, const void* key,
                                                        const void* value, void* out, float* num_state, float* den_state,
                                                        float* max_state, int B, int T, int C, int dt, int has_state) {
  const int c = blockIdx.x * 128 + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  const float w = time_decay[c], u = time_first[c];
  float aa = 0.0f, bb = 0.0f, pp = -1e38f;
  if (has_state) {
    aa = num_state[(int64_t)b * C + c];
    bb = den_state[(int64_t)b * C + c];
    pp = max_state[(int64_t)b * C + c];
  }
  for (int t = 0; t < T; ++t) {
    const int64_t i = ((int64_t)b * T + t) * C + c;
    const float kk = rd_load(key, i, dt), vv = rd_load(value, i, dt);
    float ww = u + kk;
    float p = fmaxf(pp, ww);
    float e1 = __expf(pp - p), e2 = __expf(ww - p);
    rd_store(out, i, (e1 * aa + e2 * vv) / (e1 * bb + e2), dt);
    ww = pp + w;
    p = fmaxf(ww, kk);
    e1 = __expf(ww - p);
    e2 = __expf(kk - p);
    aa = e1 * aa + e2 * vv;
    bb = e1 * bb + e2;
    pp = p;
  }
  if (num_state != nullptr) {
    num_state[(int64_t)b * C + c] = aa;
    den_state[(int64_t)b * C + c] = bb;
    max_state[(int64_t)b * C + c] = pp;
  }
}

// r / k / v: [B, H, T, S] fp32; time_decay, time_first: [H, S] fp32; state [B, H, S, S] fp32 (in/out, state[i][j]);
// out [B, T, H, S] fp32.   out_t[j] = sum_i r_t[i] * (u[i] k_t[i] v_t[j] + state[i][j]);  state = state * w[i] + k v^T
template <int S>
__global__ __launch_bounds__(S) void rwkv5_wkv_kernel(const float* time_decay, const float* time_first, const float* r,
                                                      const float* k, const float* v, float* state, float* out, int B, int H,
                                                      int T) {
  __shared__ float rs[S], ks[S], ws[S], us[S];
  const int j = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
  float st[S];
  float* sp = state + (((int64_t)b * H + h) * S) * S;
#pragma unroll
  for (int i = 0; i < S; ++i) st[i] = sp[(int64_t)i * S + j];
  ws[j] = time_decay[h * S + j];
  us[j] = time_first[h * S + j];
  const int64_t base = ((int64_t)b * H + h) * T * S;
  for (int t = 0; t < T; ++t) {
    __syncthreads();
    rs[j] = r[base + (int64_t)t * S + j];
    ks[j] = k[base + (int64_t)t * S + j];
    const float vj = v[base + (int64_t)t * S + j];
    __syncthreads();
    float y = 0.0f;
#pragma unroll
    for (int i = 0; i < S; ++i) {
      const float kv = ks[i] * vj;
      y += rs[i] * (us[i] * kv + st[i]);
      st[i] = st[i] * ws[i] + kv;
    }
    out[(((int64_t)b * T + t) * H + h) * S + j] = y;
  }
#pragma unroll
  for (int i = 0; i < S; ++i) sp[(int64_t)i * S + j] = st[i];
}

}  // namespace bigdl

using namespace bigdl;

extern "C" int bigdl_rwkv_time_shift(const void* hidden, const void* shifted, const void* mix, void* out, int64_t rows, int C,
                                     int n_mix, int dtype, hipStream_t st) {
  if (rows <= 0 || C <= 0 || n_mix <= 0 || dtype < 0 || dtype > 2) return -1;
  const int64_t n = rows * C;
  hipLaunchKernelGGL(rwkv_time_shift_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hidden, shifted, mix, out,
                     rows, C, n_mix, dtype);
  return HIP_CHECK_LAUNCH();
}

extern "C" int bigdl_rwkv4_wkv(const float* time_decay, const float* time_first, const void* key, const void* value, void* out,
                               float* num_state, float* den_state, float* max_state, int B, int T, int C, int dtype,
                               int has_state, hipStream_t st) {
  if (B <= 0 || T <= 0 || C <= 0 || B > 65535 || dtype < 0 || dtype > 2) return -1;
  hipLaunchKernelGGL(rwkv4_wkv_kernel, dim3((C + 127) / 128, B), dim3(128), 0, st, time_decay, time_first, key, value, out,
                     num_state, den_state, max_state, B, T, C, dtype, has_state);
  return HIP_CHECK_LAUNCH();
}

extern "C" int bigdl_rwkv5_wkv(const float* time_decay, const float* time_first, const float* r, const float* k, const float* v,
                               float* state, float* out, int B, int H, int T, int S, hipStream_t st) {
  if (B <= 0 || H <= 0 || T <= 0 || B > 65535) return -1;
  const dim3 grid(H, B);
  switch (S) {
    case 64: hipLaunchKernelGGL((rwkv5_wkv_kernel<64>), grid, dim3(64), 0, st, time_decay, time_first, r, k, v, state, out, B, H, T); break;
    case 32: hipLaunchKernelGGL((rwkv5_wkv_kernel<32>), grid, dim3(32), 0, st, time_decay, time_first, r, k, v, state, out, B, H, T); break;
    default: return -2;
  }
  return HIP_CHECK_LAUNCH();
}
