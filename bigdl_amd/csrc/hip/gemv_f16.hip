// Skinny GEMM for plain fp16 weights, [N, K] row-major exactly as the checkpoint stores them (no re-layout, no second
// copy): the decode and speculative-verification path of fp16 models ("bigdl_fp16_gpu" in the reference's benchmark,
// python/llm/dev/benchmark/all-in-one/run.py; the reference's fp16 linear: transformers/low_bit_linear.py:745-840).
//
//   out[m, n] = f( rs[m] * sum_k x[m, k] * gamma'[k] * W[n, k] + bias[n] )      m < M <= 16
//
// with the same fusions as the low-bit GEMV: RMSNorm prologue (rs = rsqrt(mean(x^2) + eps), gamma' = gamma + offset),
// gate/up pair with activation (f = act(gate) * up), bias, residual add.
//
// Mapping (wave64, 8 waves per workgroup, one workgroup per 16 output columns):
//   * K is cut into 64-wide chunks; wave w takes chunks w, w + 8, ...  A lane (r = lane % 16, g = lane / 16) loads 32
//     contiguous bytes of weight row n0 + r at k = 64 c + 16 g: the four lane groups of a row cover one full 128-byte
//     line per request, so a row-major matrix streams at line granularity without a tiled copy.
//   * The MFMA (v_mfma_f32_16x16x32_f16, x as the A operand, 16 weight rows as B) does not care which k an operand
//     slot holds as long as A and B agree: slot (g, i) of the first issue is k = 64 c + 16 g + i, of the second
//     k = 64 c + 16 g + 8 + i.  x is read from global memory in exactly that order (it is a few KB and stays in L2).
//   * Four chunks (128 B per lane, 8 requests) are in flight per wave and stream before the first MFMA of a round.
//   * Partial sums of the 8 waves meet in LDS; wave 0 applies the epilogue.
#include "common.h"

namespace bigdl {

struct GemvF16Args {
  const _Float16* x;
  const _Float16* w;
  const _Float16* w2;
  void* out;
  const _Float16* bias;
  const _Float16* bias2;
  const void* residual;
  const _Float16* gamma;
  float eps, gamma_offset;
  int M, N, K, ldx, ldo, act, out_type;
};

__device__ __forceinline__ float frag_sumsq_f16(half8 a) {
  H8 u = {a};
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_fdot2(u.h2[i], u.h2[i], s, false);
  return s;
}

constexpr int F16_NW = 8;      // waves per workgroup
constexpr int F16_U = 4;       // chunks in flight per wave

template <bool GATE_UP, bool NORM>
__global__ __launch_bounds__(F16_NW * 64) void gemv_f16_kernel(GemvF16Args a) {
  constexpr int NACC = GATE_UP ? 2 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int nchunks = a.K >> 6;
  const _Float16* wrow = a.w + (int64_t)(n0 + r) * a.K + 16 * g;
  const _Float16* w2row = GATE_UP ? a.w2 + (int64_t)(n0 + r) * a.K + 16 * g : nullptr;
  const _Float16* xrow = a.x + (int64_t)min(r, a.M - 1) * a.ldx + 16 * g;
  const _Float16* grow = NORM ? a.gamma + 16 * g : nullptr;

  __shared__ float4v red[F16_NW][NACC][64];
  __shared__ float ssq_sh[F16_NW][16];

  float4v acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = float4v{0, 0, 0, 0};
  float ssq = 0.0f;

  for (int c0 = wave; c0 < nchunks; c0 += F16_NW * F16_U) {
    uint4v wv[NACC][F16_U][2], xv[F16_U][2], gv[F16_U][2];
    // x (and gamma) first: they are L2 hits the MFMAs wait for, the weight stream follows
#pragma unroll
    for (int u = 0; u < F16_U; ++u) {
      const int c = min(c0 + u * F16_NW, nchunks - 1);
      xv[u][0] = *(const uint4v*)(xrow + c * 64);
      xv[u][1] = *(const uint4v*)(xrow + c * 64 + 8);
      if constexpr (NORM) {
        gv[u][0] = *(const uint4v*)(grow + c * 64);
        gv[u][1] = *(const uint4v*)(grow + c * 64 + 8);
      }
    }
#pragma unroll
    for (int u = 0; u < F16_U; ++u) {
      const int c = min(c0 + u * F16_NW, nchunks - 1);
      wv[0][u][0] = __builtin_nontemporal_load((const uint4v*)(wrow + c * 64));
      wv[0][u][1] = __builtin_nontemporal_load((const uint4v*)(wrow + c * 64 + 8));
      if constexpr (GATE_UP) {
        wv[1][u][0] = __builtin_nontemporal_load((const uint4v*)(w2row + c * 64));
        wv[1][u][1] = __builtin_nontemporal_load((const uint4v*)(w2row + c * 64 + 8));
      }
    }
#pragma unroll
    for (int u = 0; u < F16_U; ++u) {
      const bool live = c0 + u * F16_NW < nchunks;     // (wave-uniform) a chunk past the end contributes zeros
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        H8 xa;
        xa.q = xv[u][h];
        if (!live) xa.q = uint4v{0, 0, 0, 0};
        if constexpr (NORM) {
          ssq += frag_sumsq_f16(xa.v);
          H8 gm;
          gm.q = gv[u][h];
          half8 gmv = gm.v;
          if (a.gamma_offset != 0.0f) gmv = gmv + (_Float16)a.gamma_offset;
          xa.v = xa.v * gmv;
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          H8 b;
          b.q = wv[i][u][h];
          acc[i] = mfma16(xa.v, b.v, acc[i]);
        }
      }
    }
  }

  // ---- cross-wave reduction: lane holds C[m = 4g + q][n = r], q = 0..3
#pragma unroll
  for (int i = 0; i < NACC; ++i) red[wave][i][lane] = acc[i];
  if constexpr (NORM) {
    ssq += __shfl_xor(ssq, 16);
    ssq += __shfl_xor(ssq, 32);
    if (lane < 16) ssq_sh[wave][lane] = ssq;
  }
  __syncthreads();
  if (wave != 0) return;
  float4v tot[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    tot[i] = red[0][i][lane];
#pragma unroll
    for (int w = 1; w < F16_NW; ++w) tot[i] += red[w][i][lane];
  }
  const int n = n0 + r;
  const float b1 = a.bias ? (float)a.bias[n] : 0.0f;
  const float b2 = (GATE_UP && a.bias2) ? (float)a.bias2[n] : 0.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = 4 * g + q;
    if (m >= a.M) continue;
    float rs = 1.0f;
    if constexpr (NORM) {
      float s = 0.0f;
#pragma unroll
      for (int w = 0; w < F16_NW; ++w) s += ssq_sh[w][m];
      rs = rsqrtf(s / (float)a.K + a.eps);
    }
    float v = tot[0][q] * rs + b1;
    if constexpr (GATE_UP) v = act_apply(a.act, v) * (tot[1][q] * rs + b2);
    else v = act_apply(a.act, v);
    const int64_t o = (int64_t)m * a.ldo + n;
    if (a.out_type == OUT_F16) {
      if (a.residual) v += (float)((const _Float16*)a.residual)[o];
      ((_Float16*)a.out)[o] = (_Float16)v;
    } else if (a.out_type == OUT_BF16) {
      if (a.residual) v += bf16_bits_to_f32(((const uint16_t*)a.residual)[o]);
      ((uint16_t*)a.out)[o] = f32_to_bf16_bits(v);
    } else {
      if (a.residual) v += ((const float*)a.residual)[o];
      ((float*)a.out)[o] = v;
    }
  }
}

}  // namespace bigdl

using namespace bigdl;

// Returns 0 on success; negative = rejected on the host before any launch (the caller falls back to a library GEMM).
// Requirements: M <= 16, N % 16 == 0, K % 64 == 0, x rows 16-byte aligned (ldx % 8 == 0), weights 16-byte aligned.
extern "C" int bigdl_gemv_f16(const void* x, const void* w, const void* w2, void* out, const void* bias,
                              const void* bias2, const void* residual, const void* gamma, float eps, float gamma_offset,
                              int M, int N, int K, int ldx, int ldo, int act, int out_type, hipStream_t stream) {
  if (M <= 0 || M > 16 || N <= 0 || (N & 15) || K <= 0 || (K & 63) || (ldx & 7) || ldx < K || ldo < N) return -1;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || (w2 && ((uintptr_t)w2 & 15)) || (gamma && ((uintptr_t)gamma & 15)))
    return -1;
  if (out_type < 0 || out_type > 2) return -3;
  GemvF16Args a;
  a.x = (const _Float16*)x; a.w = (const _Float16*)w; a.w2 = (const _Float16*)w2; a.out = out;
  a.bias = (const _Float16*)bias; a.bias2 = (const _Float16*)bias2; a.residual = residual;
  a.gamma = (const _Float16*)gamma; a.eps = eps; a.gamma_offset = gamma_offset;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.act = act; a.out_type = out_type;
  const dim3 grid(N / 16), block(F16_NW * 64);
  if (w2) {
    if (gamma) hipLaunchKernelGGL((gemv_f16_kernel<true, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((gemv_f16_kernel<true, false>), grid, block, 0, stream, a);
  } else {
    if (gamma) hipLaunchKernelGGL((gemv_f16_kernel<false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((gemv_f16_kernel<false, false>), grid, block, 0, stream, a);
  }
  return HIP_CHECK_LAUNCH();
}
