// Host entry points of the dequant-fused MFMA GEMM (kernel: gemm_kernel.h; one instantiation unit per weight
// format: gemm_inst_*.hip), its split-K reduce/epilogue kernel and the device-side block table of the grouped
// (mixture-of-experts) form.
#include "gemm_kernel.h"
#include <stdlib.h>

namespace bigdl {
int gemm_launch_sym_int4(const GemmArgs& a, int bm, int bn, int mblocks, hipStream_t st);
int gemm_launch_asym_int4(const GemmArgs& a, int bm, int bn, int mblocks, hipStream_t st);
int gemm_launch_woq_int4(const GemmArgs& a, int bm, int bn, int mblocks, hipStream_t st);
int gemm_launch_fp4(const GemmArgs& a, int bm, int bn, int mblocks, hipStream_t st);
int gemm_launch_nf4(const GemmArgs& a, int bm, int bn, int mblocks, hipStream_t st);
int gemm_launch_sym_int5(const GemmArgs& a, int bm, int bn, int mblocks, hipStream_t st);
int gemm_launch_asym_int5(const GemmArgs& a, int bm, int bn, int mblocks, hipStream_t st);
int gemm_launch_fp6(const GemmArgs& a, int bm, int bn, int mblocks, hipStream_t st);
int gemm_launch_sym_int8(const GemmArgs& a, int bm, int bn, int mblocks, hipStream_t st);
int gemm_launch_fp8_e4m3(const GemmArgs& a, int bm, int bn, int mblocks, hipStream_t st);
int gemm_launch_fp8_e5m2(const GemmArgs& a, int bm, int bn, int mblocks, hipStream_t st);

// out[m][n] = epilogue(sum_s partial[s][m][n]); pair form: act(first) * second; optional folded RMSNorm factor of the
// row (row_ssq) and sum of squares of the rows written (ssq_out, for the norm folded into the next product).
// V = elements per thread (4: N % 4 == 0, rows of out / residual 8-byte aligned; else 1).  A workgroup covers 256 * V
// consecutive elements; with ssq_out it adds ONE value per row it touches (LDS reduction first: same-address atomics
// from every wavefront of a row cost more than the rest of the kernel).
template <int V>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs a) {
  const int64_t total = (int64_t)a.M * a.N;
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  const bool in = i < total;
  const int64_t ic = in ? i : total - V;
  const int m = (int)(ic / a.N);
  const int n = (int)(ic - (int64_t)m * a.N);
  typedef float fvec __attribute__((ext_vector_type(V)));
  float v[V], u[V];
#pragma unroll
  for (int e = 0; e < V; ++e) v[e] = u[e] = 0.0f;
  // the slabs of up to four splits are requested together: one at a time, every split costs its own memory round
  // trip (8 splits: 8 dependent trips, most of this kernel's time)
  auto gather = [&](float (&acc)[V], const float* base) {
    int s = 0;
    for (; s + 4 <= a.splitk; s += 4) {
      const fvec p0 = *(const fvec*)(base + (int64_t)s * total + ic);
      const fvec p1 = *(const fvec*)(base + (int64_t)(s + 1) * total + ic);
      const fvec p2 = *(const fvec*)(base + (int64_t)(s + 2) * total + ic);
      const fvec p3 = *(const fvec*)(base + (int64_t)(s + 3) * total + ic);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += (p0[e] + p1[e]) + (p2[e] + p3[e]);
    }
    for (; s < a.splitk; ++s) {
      const fvec p = *(const fvec*)(base + (int64_t)s * total + ic);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += p[e];
    }
  };
  gather(v, a.partial);
  if (a.w2 != nullptr) gather(u, a.partial + a.pair_stride);
  const float rs = a.row_ssq != nullptr ? rsqrtf(a.row_ssq[m] / (float)a.K + a.eps) : 1.0f;
  const float rsc = a.row_scale != nullptr ? a.row_scale[m] : 1.0f;
  const int64_t oi = (int64_t)m * a.ldo + n;
  float q = 0.0f;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    float x = v[e] * rs;
    if (a.bias != nullptr) x += (float)((const _Float16*)a.bias)[n + e];
    x = act_apply(a.act, x);
    if (a.w2 != nullptr) {
      float y = u[e] * rs;
      if (a.bias2 != nullptr) y += (float)((const _Float16*)a.bias2)[n + e];
      x *= y;
    }
    x *= rsc;
    if (a.residual) {
      if (a.out_type == OUT_F16) x += load_out<OUT_F16>(a.residual, oi + e);
      else if (a.out_type == OUT_BF16) x += load_out<OUT_BF16>(a.residual, oi + e);
      else x += load_out<OUT_F32>(a.residual, oi + e);
    }
    // what is stored, rounded (the next norm sees the rounded values)
    if (a.out_type == OUT_F16) x = (float)(_Float16)x;
    else if (a.out_type == OUT_BF16) x = bf16_bits_to_f32(f32_to_bf16_bits(x));
    v[e] = x;
    q += x * x;
  }
  if (in) {
    if (a.out_type == OUT_F16) {
      _Float16 h[V];
#pragma unroll
      for (int e = 0; e < V; ++e) h[e] = (_Float16)v[e];
      if constexpr (V == 4) *(uint2v*)((_Float16*)a.out + oi) = *(const uint2v*)h;
      else ((_Float16*)a.out)[oi] = h[0];
    } else if (a.out_type == OUT_BF16) {
#pragma unroll
      for (int e = 0; e < V; ++e) store_out<OUT_BF16>(a.out, oi + e, v[e]);
    } else {
#pragma unroll
      for (int e = 0; e < V; ++e) store_out<OUT_F32>(a.out, oi + e, v[e]);
    }
  }
  if (a.ssq_out != nullptr) {
    __shared__ float part[4];
    __shared__ int rowof[4];
    if (!in) q = 0.0f;
    const int wave = threadIdx.x >> 6;
    const int m_first = __shfl(m, 0), m_last = __shfl(m, 63);
    const bool uniform = m_first == m_last;            // (N % (64 V) == 0 makes every wavefront single-row)
    if (uniform) {
      q = wave_sum(q);
      if ((threadIdx.x & 63) == 0) { part[wave] = q; rowof[wave] = m; }
    } else {
      if (in) unsafeAtomicAdd(a.ssq_out + m, q);
      if ((threadIdx.x & 63) == 0) { part[wave] = 0.0f; rowof[wave] = -1; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      // merge the wavefronts that share a row: at most one add per row and workgroup
      for (int w = 0; w < 4; ++w) {
        if (rowof[w] < 0) continue;
        float t = part[w];
        for (int y = w + 1; y < 4; ++y)
          if (rowof[y] == rowof[w]) { t += part[y]; rowof[y] = -1; }
        // hardware fp32 add (plain atomicAdd compiles to a compare-and-swap loop)
        unsafeAtomicAdd(a.ssq_out + rowof[w], t);
      }
    }
  }
}

static int launch_reduce(const GemmArgs& a, hipStream_t stream) {
  const int64_t total = (int64_t)a.M * a.N;
  const bool vec = (a.N % 4 == 0) && (a.ldo % 4 == 0) && (((uintptr_t)a.out & 7) == 0) &&
                   (a.residual == nullptr || ((uintptr_t)a.residual & 7) == 0) && (a.pair_stride % 4 == 0) &&
                   (((uintptr_t)a.partial & 15) == 0);
  if (vec) hipLaunchKernelGGL(gemm_splitk_reduce_kernel<4>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(gemm_splitk_reduce_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
  return HIP_CHECK_LAUNCH();
}

int gemm_launch_reduce(const GemmArgs& a, hipStream_t stream) { return launch_reduce(a, stream); }   // gemm_big.hip

// counts[e] rows of expert e, rows sorted by expert -> table of m-blocks of at most bm rows; unused entries get
// expert -1.  One wavefront: E is at most a few hundred.
__global__ void moe_block_table_kernel(const int64_t* counts, int n_experts, int bm, int max_blocks, int* blk_expert,
                                       int* blk_row0, int* blk_rows) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int b = 0;
  int64_t row = 0;
  for (int e = 0; e < n_experts; ++e) {
    int64_t c = counts[e];
    for (int64_t o = 0; o < c && b < max_blocks; o += bm, ++b) {
      blk_expert[b] = e;
      blk_row0[b] = (int)(row + o);
      blk_rows[b] = (int)min((int64_t)bm, c - o);
    }
    row += c;
  }
  for (; b < max_blocks; ++b) { blk_expert[b] = -1; blk_row0[b] = 0; blk_rows[b] = 0; }
}
}  // namespace bigdl

using namespace bigdl;

static int dispatch(const GemmArgs& a, int qtype, int bm, int bn, int mblocks, hipStream_t st) {
  switch (qtype) {
    case Q_SYM_INT4: return gemm_launch_sym_int4(a, bm, bn, mblocks, st);
    case Q_ASYM_INT4: return gemm_launch_asym_int4(a, bm, bn, mblocks, st);
    case Q_WOQ_INT4: return gemm_launch_woq_int4(a, bm, bn, mblocks, st);
    case Q_FP4: return gemm_launch_fp4(a, bm, bn, mblocks, st);
    case Q_NF4: return gemm_launch_nf4(a, bm, bn, mblocks, st);
    case Q_SYM_INT5: return gemm_launch_sym_int5(a, bm, bn, mblocks, st);
    case Q_ASYM_INT5: return gemm_launch_asym_int5(a, bm, bn, mblocks, st);
    case Q_FP6: return gemm_launch_fp6(a, bm, bn, mblocks, st);
    case Q_SYM_INT8: return gemm_launch_sym_int8(a, bm, bn, mblocks, st);
    case Q_FP8_E4M3: return gemm_launch_fp8_e4m3(a, bm, bn, mblocks, st);
    case Q_FP8_E5M2: return gemm_launch_fp8_e5m2(a, bm, bn, mblocks, st);
    default: return -2;
  }
}

static void plan(int mblocks, int bn, int N, int K, int* splitk, int* kt_per_split) {
  const int KT = (K + 127) / 128;
  const int blocks = ((N + bn - 1) / bn) * mblocks;
  int sk = 1;
  static const int target = getenv("BIGDL_GEMM_TARGET") ? atoi(getenv("BIGDL_GEMM_TARGET")) : 256;
  if (blocks < target * 3 / 4) sk = (target + blocks - 1) / blocks;
  sk = sk < 1 ? 1 : sk;
  if (sk > KT / 2) sk = KT / 2 > 0 ? KT / 2 : 1;   // at least two k tiles per split
  const int per = (KT + sk - 1) / sk;
  *kt_per_split = per;
  *splitk = (KT + per - 1) / per;
}

// Number of K splits the launch below will use for this shape (the caller provides splitk * M * N floats of
// workspace when it is > 1).  mblocks: number of row blocks (dense: ceil(M / bm)).
extern "C" int bigdl_gemm_lowbit_bm(int M);
static int pick_bn(int M) {
  (void)M;
  return 128;
}

extern "C" int bigdl_gemm_lowbit_splitk(int M, int N, int K) {
  int sk, per;
  const int bm = bigdl_gemm_lowbit_bm(M);
  plan((M + bm - 1) / bm, pick_bn(M), N, K, &sk, &per);
  return sk;
}

extern "C" int bigdl_gemm_lowbit_bm(int M) {
  static const int forced = getenv("BIGDL_GEMM_BM") ? atoi(getenv("BIGDL_GEMM_BM")) : 0;
  if (forced == 16 || forced == 32 || forced == 64 || forced == 128) return forced;
  // 64-row blocks keep two or more wavefronts per SIMD (170 VGPRs): measured 10-40 % faster than 128-row blocks at
  // every M between 64 and 1024 on the Llama-2-7B projection shapes; 16 / 32-row blocks for row counts that fit them
  // (a quarter / half of the x tile traffic of a 64-row block)
  static const int small = getenv("BIGDL_GEMM_SMALL_BM") ? atoi(getenv("BIGDL_GEMM_SMALL_BM")) : 1;
  if (small && M <= 16) return 16;
  if (small && M <= 32) return 32;
  return 64;
}

// Dense form.  x fp16 [M, ldx] (ldx % 8 == 0, 16-byte aligned), K % 32 == 0.  workspace: fp32, >= splitk*M*N.
extern "C" int bigdl_gemm_lowbit(const void* x, const void* w, void* out, const void* bias, const void* residual,
                                 float* workspace, int64_t workspace_floats, int M, int N, int K, int ldx, int ldo,
                                 int qtype, int act, int out_type, hipStream_t stream) {
  qtype = t16_kernel_qtype(qtype);
  if (M <= 0 || N <= 0 || K <= 0 || (K & 31) || (ldx & 7) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return -1;
  if (out_type < 0 || out_type > 2) return -3;
  if ((int64_t)M * ldx >= (1ll << 31)) return -6;
  GemmArgs a = {};
  a.x = (const _Float16*)x; a.w = w; a.out = out; a.bias = bias; a.residual = residual; a.partial = workspace;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.act = act; a.out_type = out_type;
  const int bm = bigdl_gemm_lowbit_bm(M);
  const int bn = pick_bn(M);
  const int mblocks = (M + bm - 1) / bm;
  plan(mblocks, bn, N, K, &a.splitk, &a.kt_per_split);
  if (a.splitk > 1 && (workspace == nullptr || workspace_floats < (int64_t)a.splitk * M * N)) return -5;
  const int rc = dispatch(a, qtype, bm, bn, mblocks, stream);
  if (rc != 0 || a.splitk == 1) return rc;
  return launch_reduce(a, stream);
}

// Extended dense form for batched decode: RMSNorm folded in (gamma + row_ssq of x, see GemmArgs), gate / up pair in one
// launch (w2: out = act(x W^T + bias) * (x W2^T + bias2)), and the sums of squares of the output rows (ssq_out, after
// the residual add) for the norm folded into the next product.  workspace: fp32, >= (w2 ? 2 : 1) * splitk * M * N.
extern "C" int bigdl_gemm_lowbit_ex(const void* x, const void* w, const void* w2, void* out, const void* bias, const void* bias2,
                                    const void* residual, const void* gamma, float gamma_offset, float eps,
                                    const float* row_ssq, float* ssq_out, float* workspace, int64_t workspace_floats, int M,
                                    int N, int K, int ldx, int ldo, int qtype, int act, int out_type, hipStream_t stream) {
  qtype = t16_kernel_qtype(qtype);
  if (M <= 0 || N <= 0 || K <= 0 || (K & 31) || (ldx & 7) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return -1;
  if (w2 != nullptr && ((uintptr_t)w2 & 15)) return -1;
  if (gamma != nullptr && (((uintptr_t)gamma & 15) || row_ssq == nullptr)) return -1;
  if (out_type < 0 || out_type > 2) return -3;
  if ((int64_t)M * ldx >= (1ll << 31)) return -6;
  GemmArgs a = {};
  a.x = (const _Float16*)x; a.w = w; a.out = out; a.bias = bias; a.residual = residual; a.partial = workspace;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.act = act; a.out_type = out_type;
  a.gamma = (const _Float16*)gamma; a.gamma_offset = gamma_offset; a.eps = eps; a.row_ssq = gamma != nullptr ? row_ssq : nullptr;
  a.ssq_out = ssq_out; a.w2 = w2; a.bias2 = bias2;
  const int bm = gamma != nullptr ? (bigdl_gemm_lowbit_bm(M) < 64 ? 64 : bigdl_gemm_lowbit_bm(M)) : bigdl_gemm_lowbit_bm(M);
  const int bn = pick_bn(M);
  const int mblocks = (M + bm - 1) / bm;
  plan(mblocks * (w2 != nullptr ? 2 : 1), bn, N, K, &a.splitk, &a.kt_per_split);
  a.force_partial = (w2 != nullptr || ssq_out != nullptr) ? 1 : 0;
  a.pair_stride = (int64_t)a.splitk * M * N;
  const bool partial_out = a.splitk > 1 || a.force_partial;
  if (partial_out && (workspace == nullptr || workspace_floats < (w2 != nullptr ? 2 : 1) * (int64_t)a.splitk * M * N)) return -5;
  const int rc = dispatch(a, qtype, bm, bn, mblocks, stream);
  if (rc != 0 || !partial_out) return rc;
  return launch_reduce(a, stream);
}

// splits the extended form will use (workspace sizing): pair = 1 when w2 is given
extern "C" int bigdl_gemm_lowbit_ex_splitk(int M, int N, int K, int pair) {
  int sk, per;
  const int bm = bigdl_gemm_lowbit_bm(M);
  plan(((M + bm - 1) / bm) * (pair ? 2 : 1), pick_bn(M), N, K, &sk, &per);
  return sk;
}

// Grouped form: `rows` sorted rows in total; row i uses the weights of the expert that the block table assigns
// to its block, reads x row x_rows[i] (null: i) and writes out row i (scaled by row_scale[i] if given).
// The experts' [N, K] matrices are stacked by rows in w (N % 16 == 0).  bm is 64 or 128 and must be the value the
// block table was built with; max_blocks >= ceil(rows / bm) + n_experts.
extern "C" int bigdl_gemm_lowbit_grouped(const void* x, const void* w, void* out, const int* blk_expert,
                                         const int* blk_row0, const int* blk_rows, const int* x_rows,
                                         const float* row_scale, int rows, int max_blocks, int bm, int n_experts,
                                         int N, int K, int ldx, int ldo, int qtype, int act, int out_type,
                                         int x_row_count, hipStream_t stream) {
  qtype = t16_kernel_qtype(qtype);
  if (x_row_count <= 0 || (int64_t)x_row_count * ldx >= (1ll << 31)) return -6;
  if (rows <= 0 || N <= 0 || K <= 0 || (K & 31) || (N & 15) || (ldx & 7) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15) ||
      blk_expert == nullptr || blk_row0 == nullptr || blk_rows == nullptr || (bm != 64 && bm != 128) || max_blocks <= 0 ||
      max_blocks > 65535 || n_experts <= 0)
    return -1;
  if (out_type < 0 || out_type > 2) return -3;
  GemmArgs a = {};
  a.x = (const _Float16*)x; a.w = w; a.out = out; a.M = rows; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.act = act;
  a.out_type = out_type; a.splitk = 1; a.kt_per_split = (K + 127) / 128;
  a.blk_expert = blk_expert; a.blk_row0 = blk_row0; a.blk_rows = blk_rows; a.x_rows = x_rows; a.row_scale = row_scale;
  a.n_total = n_experts * N;
  return dispatch(a, qtype, bm, 128, max_blocks, stream);
}

extern "C" int bigdl_moe_block_table(const int64_t* counts, int n_experts, int bm, int max_blocks, int* blk_expert,
                                     int* blk_row0, int* blk_rows, hipStream_t stream) {
  if (counts == nullptr || n_experts <= 0 || bm <= 0 || max_blocks <= 0) return -1;
  hipLaunchKernelGGL(moe_block_table_kernel, dim3(1), dim3(64), 0, stream, counts, n_experts, bm, max_blocks, blk_expert,
                     blk_row0, blk_rows);
  return HIP_CHECK_LAUNCH();
}
