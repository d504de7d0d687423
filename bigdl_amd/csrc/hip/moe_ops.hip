// Mixture-of-experts routing helpers for the decode step on gfx950.
//  * moe_route: softmax over the router logits + top-k (+ renormalisation) per token, one wavefront per token;
//    writes int32 expert ids and fp32 weights that the routed GEMV (gemv_kernel.h, slot mode) reads on the device.
//    Optional expert range [lo, hi): ids outside are written as -1 (expert-parallel ranks skip them), ids inside
//    are rebased to the local index.
//  * moe_combine: out[t] = residual[t] + sum_k y[t*k + kk] + sigmoid(shared_gate[t]) * shared[t]
// They replace the reference's xe_addons moe_softmax_topk / moe_group_topk routing + the python-side
// index_add loops (python/llm/src/ipex_llm/transformers/models/common.py:339-354, mixtral.py:60-130).
#include "common.h"

namespace bigdl {

constexpr int ROUTE_MAX_PER_LANE = 16;   // experts <= 64 * 16

template <bool F32>
__global__ __launch_bounds__(256) void moe_route_kernel(const void* logits, int* ids, float* weights, int T, int E, int top_k,
                                                        int normalize, int lo, int hi, int64_t ld, int sigmoid_scores) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  float v[ROUTE_MAX_PER_LANE];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < ROUTE_MAX_PER_LANE; ++i) {
    const int e = i * 64 + lane;
    float x = -3.0e38f;
    if (e < E) x = F32 ? ((const float*)logits)[t * ld + e] : (float)((const _Float16*)logits)[t * ld + e];
    v[i] = x;
    mx = fmaxf(mx, x);
  }
  mx = wave_max(mx);
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < ROUTE_MAX_PER_LANE; ++i) {
    const int e = i * 64 + lane;
    if (e < E) {
      v[i] = sigmoid_scores ? 1.0f / (1.0f + __expf(-v[i])) : __expf(v[i] - mx);
      sum += v[i];
    } else {
      v[i] = -1.0f;
    }
  }
  sum = wave_sum(sum);
  const float inv = sigmoid_scores ? 1.0f : 1.0f / sum;
  float picked = 0.0f;
  float my_w = 0.0f;
  int my_id = -1;
  for (int k = 0; k < top_k; ++k) {
    // wave argmax (ties: lowest expert index, as torch.topk on the sorted order would give for equal values)
    float best = -1.0f;
    int best_e = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < ROUTE_MAX_PER_LANE; ++i) {
      const int e = i * 64 + lane;
      if (v[i] > best) { best = v[i]; best_e = e; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off);
      const int oe = __shfl_xor(best_e, off);
      if (ob > best || (ob == best && oe < best_e)) { best = ob; best_e = oe; }
    }
#pragma unroll
    for (int i = 0; i < ROUTE_MAX_PER_LANE; ++i)
      if (i * 64 + lane == best_e) v[i] = -2.0f;   // taken
    picked += best * inv;
    if (lane == k) { my_w = best * inv; my_id = best_e; }
  }
  if (lane < top_k) {
    float w = my_w;
    if (normalize) w /= picked;
    int id = my_id;
    if (hi > lo) id = (id >= lo && id < hi) ? id - lo : -1;
    ids[t * top_k + lane] = id;
    weights[t * top_k + lane] = w;
  }
}

__global__ __launch_bounds__(256) void moe_combine_kernel(const _Float16* y, const _Float16* shared, const _Float16* shared_gate,
                                                          const _Float16* residual, _Float16* out, int T, int top_k, int H,
                                                          int64_t ld_gate) {
  const int t = blockIdx.y;
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= H) return;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < top_k; ++k) {
    const half8 h = *(const half8*)(y + ((int64_t)t * top_k + k) * H + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += (float)h[j];
  }
  if (shared != nullptr) {
    float gs = 1.0f;
    if (shared_gate != nullptr) gs = 1.0f / (1.0f + __expf(-(float)shared_gate[t * ld_gate]));
    const half8 h = *(const half8*)(shared + (int64_t)t * H + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += gs * (float)h[j];
  }
  if (residual != nullptr) {
    const half8 h = *(const half8*)(residual + (int64_t)t * H + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += (float)h[j];
  }
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (_Float16)acc[j];
  *(half8*)(out + (int64_t)t * H + c) = o;
}

}  // namespace bigdl

using namespace bigdl;

// logits [T, E] (fp16 or fp32, row stride ld) -> ids int32 [T, top_k], weights fp32 [T, top_k]
extern "C" int bigdl_moe_route(const void* logits, int* ids, float* weights, int T, int E, int top_k, int normalize, int lo,
                               int hi, int64_t ld, int is_f32, int sigmoid_scores, hipStream_t st) {
  if (T <= 0 || E <= 0 || E > 64 * ROUTE_MAX_PER_LANE || top_k <= 0 || top_k > 64 || top_k > E) return -1;
  const dim3 grid((T + 3) / 4), block(256);
  if (is_f32) hipLaunchKernelGGL((moe_route_kernel<true>), grid, block, 0, st, logits, ids, weights, T, E, top_k, normalize, lo, hi, ld, sigmoid_scores);
  else hipLaunchKernelGGL((moe_route_kernel<false>), grid, block, 0, st, logits, ids, weights, T, E, top_k, normalize, lo, hi, ld, sigmoid_scores);
  return HIP_CHECK_LAUNCH();
}

// all fp16, contiguous rows of H (H % 8 == 0); shared / shared_gate / residual may be null; out may alias residual
extern "C" int bigdl_moe_combine(const void* y, const void* shared, const void* shared_gate, const void* residual, void* out,
                                 int T, int top_k, int H, int64_t ld_gate, hipStream_t st) {
  if (T <= 0 || T > 65535 || top_k <= 0 || H <= 0 || (H & 7)) return -1;
  const dim3 grid((H / 8 + 255) / 256, T), block(256);
  hipLaunchKernelGGL(moe_combine_kernel, grid, block, 0, st, (const _Float16*)y, (const _Float16*)shared,
                     (const _Float16*)shared_gate, (const _Float16*)residual, (_Float16*)out, T, top_k, H, ld_gate);
  return HIP_CHECK_LAUNCH();
}
