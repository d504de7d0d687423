// Mixture-of-experts routing helpers for the decode step on gfx950.
//  * moe_route: softmax over the router logits + top-k (+ renormalisation) per token, one wavefront per token;
//    writes int32 expert ids and fp32 weights that the routed GEMV (gemv_kernel.h, slot mode) reads on the device.
//    Optional expert range [lo, hi): ids outside are written as -1 (expert-parallel ranks skip them), ids inside
//    are rebased to the local index.
//  * moe_combine: out[t] = residual[t] + sum_k y[t*k + kk] + sigmoid(shared_gate[t]) * shared[t]
//  * moe_norm_route: RMSNorm of the residual row, the router product (E x H fp16) and the top-k of moe_route in ONE
//    launch per decode step: one workgroup per token keeps the normed row in LDS, every wavefront takes experts
//    wave, wave + 4, ... (all 16-byte requests of a router row issued before the first product), wavefront 0 picks.
//    Replaces three launches (norm, library GEMM for 8 .. 64 logits, top-k) whose fixed cost is most of their time.
// They replace the reference's xe_addons moe_softmax_topk / moe_group_topk routing + the python-side
// index_add loops (python/llm/src/ipex_llm/transformers/models/common.py:339-354, mixtral.py:60-130).
#include "common.h"

namespace bigdl {

constexpr int ROUTE_MAX_PER_LANE = 16;   // experts <= 64 * 16

// softmax (or sigmoid) scores + top-k of one token by one wavefront: lane l holds the logits of experts i * 64 + l in v[i]
// (-3e38 beyond E), mx = this lane's maximum
__device__ __forceinline__ void route_pick(float (&v)[ROUTE_MAX_PER_LANE], float mx, int t, int E, int top_k, int normalize,
                                           int lo, int hi, int sigmoid_scores, int* ids, float* weights) {
  const int lane = threadIdx.x & 63;
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
  route_pick(v, mx, t, E, top_k, normalize, lo, hi, sigmoid_scores, ids, weights);
}

// h [T, ldh] fp16 -> x = rmsnorm(h) * (gamma + offset) [T, H] fp16, router logits of x (rounded to fp16 like the dense
// product they replace) -> ids / weights.  H % 8 == 0, H <= 8192; E <= 64 * ROUTE_MAX_PER_LANE.
__global__ __launch_bounds__(256) void moe_norm_route_kernel(const _Float16* h, const _Float16* gamma, const _Float16* router,
                                                             _Float16* x_out, int* ids, float* weights, int H, int64_t ldh,
                                                             float eps, float offset, int E, int top_k, int normalize, int lo,
                                                             int hi, int sigmoid_scores) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  _Float16* xs = (_Float16*)smem;                       // [H]
  float* lg = (float*)(xs + H);                         // [E]
  float* sh = lg + ((E + 3) & ~3);                      // [4]
  const int t = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float v[4][8];
  half8 gm[4];
  float ss = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (threadIdx.x + i * 256) * 8;
    if (c < H) {
      const half8 hv = *(const half8*)(h + (int64_t)t * ldh + c);
      gm[i] = *(const half8*)(gamma + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[i][j] = (float)hv[j]; ss += v[i][j] * v[i][j]; }
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) sh[wave] = ss;
  __syncthreads();
  const float rs = rsqrtf((sh[0] + sh[1] + sh[2] + sh[3]) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (threadIdx.x + i * 256) * 8;
    if (c < H) {
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (_Float16)(v[i][j] * rs * ((float)gm[i][j] + offset));
      *(half8*)(xs + c) = o;
      *(half8*)(x_out + (int64_t)t * H + c) = o;
    }
  }
  __syncthreads();
  // router rows: lane l of the owning wavefront takes the 16-byte chunks l, l + 64, ... of the row
  const int nch = H >> 3;
  for (int e = wave; e < E; e += 4) {
    const _Float16* row = router + (int64_t)e * H;
    float acc = 0.0f;
    for (int c0 = 0; c0 < nch; c0 += 64 * 8) {
      half8 w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int ch = c0 + u * 64 + lane;
        if (ch < nch) w[u] = *(const half8*)(row + ch * 8);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int ch = c0 + u * 64 + lane;
        if (ch < nch) {
          const half8 xv = *(const half8*)(xs + ch * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc += (float)w[u][j] * (float)xv[j];
        }
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) lg[e] = (float)(_Float16)acc;
  }
  __syncthreads();
  if (wave != 0) return;
  float lv[ROUTE_MAX_PER_LANE];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < ROUTE_MAX_PER_LANE; ++i) {
    const int e = i * 64 + lane;
    lv[i] = e < E ? lg[e] : -3.0e38f;
    mx = fmaxf(mx, lv[i]);
  }
  route_pick(lv, mx, t, E, top_k, normalize, lo, hi, sigmoid_scores, ids, weights);
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

// h [T, ldh] fp16, gamma [H] fp16, router [E, H] fp16 (contiguous rows) -> x [T, H] fp16, ids int32 / weights fp32 [T, top_k]
extern "C" int bigdl_moe_norm_route(const void* h, const void* gamma, const void* router, void* x, int* ids, float* weights,
                                    int T, int H, int64_t ldh, float eps, float offset, int E, int top_k, int normalize,
                                    int lo, int hi, int sigmoid_scores, hipStream_t st) {
  if (T <= 0 || T > 65535 || H <= 0 || (H & 7) || H > 8192 || ldh < H || (ldh & 7) || E <= 0 || E > 64 * ROUTE_MAX_PER_LANE ||
      top_k <= 0 || top_k > 64 || top_k > E)
    return -1;
  const size_t lds = (size_t)H * 2 + (size_t)((E + 3) & ~3) * 4 + 16;
  hipLaunchKernelGGL(moe_norm_route_kernel, dim3(T), dim3(256), lds, st, (const _Float16*)h, (const _Float16*)gamma,
                     (const _Float16*)router, (_Float16*)x, ids, weights, H, ldh, eps, offset, E, top_k, normalize, lo, hi,
                     sigmoid_scores);
  return HIP_CHECK_LAUNCH();
}
