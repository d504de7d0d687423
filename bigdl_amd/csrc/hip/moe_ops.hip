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
// They replace the reference's python-side routing (softmax + torch.topk), its host-side bucketing
// (xe_linear.get_moe_indexes) and index_add loops: python/llm/src/ipex_llm/transformers/models/qwen2_moe.py:376-452,
// models/phixtral.py:44-104.
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

// Group-limited / biased selection (DeepSeek-V2 and V3 routers) by one wavefront: p = softmax or sigmoid of the logits,
// choice = p + bias; with n_group > 1 the groups are ranked by their best (or the sum of their two best) choice values,
// experts outside the topk_group best groups get choice 0 (not -inf: that is what those routers do), the top_k best
// choices are picked and weighted by their p (renormalized with the routers' 1e-20 guard when asked).  ``ch`` is LDS
// scratch of E + n_group floats; lane l holds experts i * 64 + l.
__device__ __forceinline__ void route_pick_grouped(float (&v)[ROUTE_MAX_PER_LANE], float mx, int t, int E, int top_k,
                                                   int normalize, int lo, int hi, int sigmoid_scores, const float* bias,
                                                   int n_group, int topk_group, int group_top2, float* ch, int* ids,
                                                   float* weights) {
  const int lane = threadIdx.x & 63;
  const float NEG = -3.0e38f;
  float pr[ROUTE_MAX_PER_LANE];
  mx = wave_max(mx);
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < ROUTE_MAX_PER_LANE; ++i) {
    const int e = i * 64 + lane;
    pr[i] = 0.0f;
    if (e < E) {
      pr[i] = sigmoid_scores ? 1.0f / (1.0f + __expf(-v[i])) : __expf(v[i] - mx);
      sum += pr[i];
    }
  }
  sum = wave_sum(sum);
  const float inv = sigmoid_scores ? 1.0f : 1.0f / sum;
#pragma unroll
  for (int i = 0; i < ROUTE_MAX_PER_LANE; ++i) {
    const int e = i * 64 + lane;
    pr[i] *= inv;
    v[i] = e < E ? pr[i] + (bias != nullptr ? bias[e] : 0.0f) : NEG;
  }
  if (n_group > 1) {
    const int gs = E / n_group;
#pragma unroll
    for (int i = 0; i < ROUTE_MAX_PER_LANE; ++i) {
      const int e = i * 64 + lane;
      if (e < E) ch[e] = v[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float score = NEG;                       // lane g < n_group: score of group g
    if (lane < n_group) {
      float b1 = NEG, b2 = NEG;
      for (int j = 0; j < gs; ++j) {
        const float x = ch[lane * gs + j];
        if (x > b1) { b2 = b1; b1 = x; } else if (x > b2) { b2 = x; }
      }
      score = group_top2 ? b1 + (gs > 1 ? b2 : 0.0f) : b1;
    }
    bool keep = false;
    for (int k = 0; k < topk_group; ++k) {
      float best = score;
      int best_g = lane < n_group ? lane : 0x7fffffff;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int og = __shfl_xor(best_g, off);
        if (ob > best || (ob == best && og < best_g)) { best = ob; best_g = og; }
      }
      if (lane == best_g) { keep = true; score = NEG; }
    }
    if (lane < n_group) ch[E + lane] = keep ? 1.0f : 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < ROUTE_MAX_PER_LANE; ++i) {
      const int e = i * 64 + lane;
      if (e < E && ch[E + e / gs] == 0.0f) v[i] = 0.0f;
    }
  }
  float picked = 0.0f, my_w = 0.0f;
  int my_id = -1;
  for (int k = 0; k < top_k; ++k) {
    float best = NEG, best_p = 0.0f;
    int best_e = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < ROUTE_MAX_PER_LANE; ++i) {
      const int e = i * 64 + lane;
      if (e < E && v[i] > best) { best = v[i]; best_e = e; best_p = pr[i]; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off), op = __shfl_xor(best_p, off);
      const int oe = __shfl_xor(best_e, off);
      if (ob > best || (ob == best && oe < best_e)) { best = ob; best_e = oe; best_p = op; }
    }
#pragma unroll
    for (int i = 0; i < ROUTE_MAX_PER_LANE; ++i)
      if (i * 64 + lane == best_e) v[i] = NEG;   // taken
    picked += best_p;
    if (lane == k) { my_w = best_p; my_id = best_e; }
  }
  if (lane < top_k) {
    float w = my_w;
    if (normalize) w /= picked + 1e-20f;
    int id = my_id;
    if (hi > lo) id = (id >= lo && id < hi) ? id - lo : -1;
    ids[t * top_k + lane] = id;
    weights[t * top_k + lane] = w;
  }
}

template <bool F32>
__global__ __launch_bounds__(256) void moe_route_kernel(const void* logits, int* ids, float* weights, int T, int E, int top_k,
                                                        int normalize, int lo, int hi, int64_t ld, int sigmoid_scores,
                                                        const float* bias, int n_group, int topk_group, int group_top2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // grouped selection: [4][E + n_group] floats
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
  if (bias != nullptr || n_group > 1)
    route_pick_grouped(v, mx, t, E, top_k, normalize, lo, hi, sigmoid_scores, bias, n_group, topk_group, group_top2,
                       (float*)smem + (threadIdx.x >> 6) * (E + n_group), ids, weights);
  else
    route_pick(v, mx, t, E, top_k, normalize, lo, hi, sigmoid_scores, ids, weights);
}

// Selection by rank counting, one thread per expert (E <= NR_WAVES * 64), every cross-thread step through LDS broadcasts
// instead of dependent cross-lane permutes: p = softmax / sigmoid of the logits, choice = p + bias, optional group limit
// (groups ranked by their best or their two best choices, choices outside the topk_group best groups become 0), rank of
// each choice (ties: lower expert first), the top_k ranks write their id and p (renormalized on request).
// lg[E]: logits; ch[E + 64]: scratch.  All threads of the workgroup must call it.
__device__ __forceinline__ void route_pick_ranked(const float* lg, float* ch, int t, int E, int top_k, int normalize, int lo,
                                                  int hi, int sigmoid_scores, const float* bias, int n_group, int topk_group,
                                                  int group_top2, int* ids, float* weights) {
  const int e = threadIdx.x;
  const bool mine = e < E;
  float* gsc = ch + E;                    // [n_group] group scores, later the picked p by rank [top_k]
  float p = 0.0f;
  if (mine) {
    if (sigmoid_scores) {
      p = 1.0f / (1.0f + __expf(-lg[e]));
    } else {
      float mx = -3.0e38f, sum = 0.0f;
      for (int j = 0; j < E; ++j) mx = fmaxf(mx, lg[j]);
      for (int j = 0; j < E; ++j) sum += __expf(lg[j] - mx);
      p = __expf(lg[e] - mx) / sum;
    }
  }
  float c = mine ? p + (bias != nullptr ? bias[e] : 0.0f) : -3.0e38f;
  if (n_group > 1) {
    const int gs = E / n_group;
    if (mine) ch[e] = c;
    __syncthreads();
    if (e < n_group) {
      float b1 = -3.0e38f, b2 = -3.0e38f;
      for (int j = 0; j < gs; ++j) {
        const float x = ch[e * gs + j];
        if (x > b1) { b2 = b1; b1 = x; } else if (x > b2) { b2 = x; }
      }
      gsc[e] = group_top2 ? b1 + (gs > 1 ? b2 : 0.0f) : b1;
    }
    __syncthreads();
    if (mine) {
      const int g = e / gs;
      const float mine_s = gsc[g];
      int rank = 0;
      for (int j = 0; j < n_group; ++j) {
        const float o = gsc[j];
        rank += (o > mine_s || (o == mine_s && j < g)) ? 1 : 0;
      }
      if (rank >= topk_group) c = 0.0f;
    }
    __syncthreads();
  }
  if (mine) ch[e] = c;
  __syncthreads();
  int rank = 0x7fffffff;
  if (mine) {
    rank = 0;
    for (int j = 0; j < E; ++j) {
      const float o = ch[j];
      rank += (o > c || (o == c && j < e)) ? 1 : 0;
    }
    if (rank < top_k) gsc[rank] = p;
  }
  __syncthreads();
  if (rank < top_k) {
    float w = p;
    if (normalize) {
      float picked = 0.0f;
      for (int j = 0; j < top_k; ++j) picked += gsc[j];
      w /= picked + ((bias != nullptr || n_group > 1) ? 1e-20f : 0.0f);
    }
    int id = e;
    if (hi > lo) id = (id >= lo && id < hi) ? id - lo : -1;
    ids[t * top_k + rank] = id;
    weights[t * top_k + rank] = w;
  }
}

// h [T, ldh] fp16 -> x = rmsnorm(h) * (gamma + offset) [T, H] fp16, router logits of x (rounded to fp16 like the dense
// product they replace) -> ids / weights.  One workgroup of 8 wavefronts per token (256 VGPRs per lane: 1024 threads
// spilled 800 registers); H % 8 == 0, H <= 8192 (two 16-byte chunks per thread); E <= 512 (one thread per expert in
// the selection).  LPE = 16-byte requests per lane and router row (H <= 512 * LPE): a wavefront takes 32 / LPE experts
// at a time with all their requests in flight together, so a 64-expert, 2048-wide router (DeepSeek-V2-Lite, Qwen-MoE)
// and an 8-expert, 4096-wide one (Mixtral) are both one memory round trip.
constexpr int NR_WAVES = 8;
template <int LPE>
__global__ __launch_bounds__(NR_WAVES * 64) void moe_norm_route_kernel(const _Float16* h, const _Float16* gamma, const _Float16* router,
                                                              _Float16* x_out, int* ids, float* weights, int H, int64_t ldh,
                                                              float eps, float offset, int E, int top_k, int normalize, int lo,
                                                              int hi, int sigmoid_scores, const float* bias, int n_group,
                                                              int topk_group, int group_top2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  _Float16* xs = (_Float16*)smem;                       // [H]
  float* lg = (float*)(xs + H);                         // [E]
  float* sh = lg + ((E + 3) & ~3);                      // [16]
  float* ch = sh + 16;                                  // [E + 64]
  constexpr int EB = 32 / LPE;
  const int t = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = H >> 3;
  // the first batch of router rows is requested before anything else: it does not depend on the norm
  half8 w[EB][LPE];
  auto request = [&](int e0) {
#pragma unroll
    for (int b = 0; b < EB; ++b)
#pragma unroll
      for (int u = 0; u < LPE; ++u) {
        const int e = min(e0 + b, E - 1), chk = min(u * 64 + lane, nch - 1);     // clamped: unconditional loads
        w[b][u] = *(const half8*)(router + (int64_t)e * H + chk * 8);
      }
  };
  request(wave * EB);
  float v[2][8];
  half8 gm[2];
  float ss = 0.0f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (threadIdx.x + i * NR_WAVES * 64) * 8;
    if (c < H) {
      const half8 hv = *(const half8*)(h + (int64_t)t * ldh + c);
      gm[i] = *(const half8*)(gamma + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[i][j] = (float)hv[j]; ss += v[i][j] * v[i][j]; }
    }
  }
  ss = wave_sum_dpp(ss);
  if (lane == 0) sh[wave] = ss;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int q = 0; q < NR_WAVES; ++q) tot += sh[q];
  const float rs = rsqrtf(tot / (float)H + eps);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (threadIdx.x + i * NR_WAVES * 64) * 8;
    if (c < H) {
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (_Float16)(v[i][j] * rs * ((float)gm[i][j] + offset));
      *(half8*)(xs + c) = o;
      *(half8*)(x_out + (int64_t)t * H + c) = o;
    }
  }
  __syncthreads();
  for (int e0 = wave * EB; e0 < E; e0 += NR_WAVES * EB) {
    if (e0 != wave * EB) request(e0);
    float acc[EB];
#pragma unroll
    for (int b = 0; b < EB; ++b) {
      acc[b] = 0.0f;
#pragma unroll
      for (int u = 0; u < LPE; ++u) {
        const int chk = u * 64 + lane;
        if (chk < nch) {
          const half8 xv = *(const half8*)(xs + chk * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[b] += (float)w[b][u][j] * (float)xv[j];
        }
      }
    }
#pragma unroll
    for (int b = 0; b < EB; ++b) acc[b] = wave_sum_dpp(acc[b]);      // EB independent chains, interleaved by the scheduler
#pragma unroll
    for (int b = 0; b < EB; ++b)
      if (lane == 0 && e0 + b < E) lg[e0 + b] = (float)(_Float16)acc[b];
  }
  __syncthreads();
  route_pick_ranked(lg, ch, t, E, top_k, normalize, lo, hi, sigmoid_scores, bias, n_group, topk_group, group_top2, ids, weights);
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
static int moe_route_launch(const void* logits, int* ids, float* weights, int T, int E, int top_k, int normalize, int lo,
                            int hi, int64_t ld, int is_f32, int sigmoid_scores, const float* bias, int n_group,
                            int topk_group, int group_top2, hipStream_t st) {
  if (T <= 0 || E <= 0 || E > 64 * ROUTE_MAX_PER_LANE || top_k <= 0 || top_k > 64 || top_k > E) return -1;
  if (n_group < 1 || n_group > 64 || E % n_group || topk_group < 1 || topk_group > n_group) return -1;
  const dim3 grid((T + 3) / 4), block(256);
  const size_t lds = (bias != nullptr || n_group > 1) ? (size_t)4 * (E + n_group) * 4 : 0;
  if (is_f32) hipLaunchKernelGGL((moe_route_kernel<true>), grid, block, lds, st, logits, ids, weights, T, E, top_k, normalize, lo, hi, ld, sigmoid_scores, bias, n_group, topk_group, group_top2);
  else hipLaunchKernelGGL((moe_route_kernel<false>), grid, block, lds, st, logits, ids, weights, T, E, top_k, normalize, lo, hi, ld, sigmoid_scores, bias, n_group, topk_group, group_top2);
  return HIP_CHECK_LAUNCH();
}

extern "C" int bigdl_moe_route(const void* logits, int* ids, float* weights, int T, int E, int top_k, int normalize, int lo,
                               int hi, int64_t ld, int is_f32, int sigmoid_scores, hipStream_t st) {
  return moe_route_launch(logits, ids, weights, T, E, top_k, normalize, lo, hi, ld, is_f32, sigmoid_scores, nullptr, 1, 1, 0, st);
}

// ... with the group-limited / biased selection of route_pick_grouped (bias fp32 [E] or null)
extern "C" int bigdl_moe_route_grouped(const void* logits, int* ids, float* weights, int T, int E, int top_k, int normalize,
                                       int lo, int hi, int64_t ld, int is_f32, int sigmoid_scores, const float* bias,
                                       int n_group, int topk_group, int group_top2, hipStream_t st) {
  return moe_route_launch(logits, ids, weights, T, E, top_k, normalize, lo, hi, ld, is_f32, sigmoid_scores, bias, n_group,
                          topk_group, group_top2, st);
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

// h [T, ldh] fp16, gamma [H] fp16, router [E, H] fp16 (contiguous rows) -> x [T, H] fp16, ids int32 / weights fp32 [T, top_k].
// bias (fp32 [E], may be null), n_group / topk_group / group_top2: the group-limited selection of route_pick_grouped.
extern "C" int bigdl_moe_norm_route(const void* h, const void* gamma, const void* router, void* x, int* ids, float* weights,
                                    int T, int H, int64_t ldh, float eps, float offset, int E, int top_k, int normalize,
                                    int lo, int hi, int sigmoid_scores, const float* bias, int n_group, int topk_group,
                                    int group_top2, hipStream_t st) {
  if (T <= 0 || T > 65535 || H <= 0 || (H & 7) || H > 8192 || ldh < H || (ldh & 7) || E <= 0 || E > 64 * ROUTE_MAX_PER_LANE ||
      top_k <= 0 || top_k > 64 || top_k > E)
    return -1;
  if (n_group < 1 || n_group > 64 || E % n_group || topk_group < 1 || topk_group > n_group) return -1;
  const size_t lds = (size_t)H * 2 + (size_t)((E + 3) & ~3) * 4 + 64 + (size_t)(E + 64) * 4;
  if (E > NR_WAVES * 64) return -1;
#define BIGDL_NR_LAUNCH(LPE)                                                                                                 \
  hipLaunchKernelGGL((moe_norm_route_kernel<LPE>), dim3(T), dim3(NR_WAVES * 64), lds, st, (const _Float16*)h, (const _Float16*)gamma, \
                     (const _Float16*)router, (_Float16*)x, ids, weights, H, ldh, eps, offset, E, top_k, normalize, lo, hi,  \
                     sigmoid_scores, bias, n_group, topk_group, group_top2)
  if (H <= 512) BIGDL_NR_LAUNCH(1);
  else if (H <= 1024) BIGDL_NR_LAUNCH(2);
  else if (H <= 2048) BIGDL_NR_LAUNCH(4);
  else if (H <= 4096) BIGDL_NR_LAUNCH(8);
  else BIGDL_NR_LAUNCH(16);
#undef BIGDL_NR_LAUNCH
  return HIP_CHECK_LAUNCH();
}
