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
//     line, so a row-major matrix streams at line granularity without a tiled copy.
//   * The MFMA (v_mfma_f32_16x16x32_f16, x as the A operand, 16 weight rows as B) does not care which k an operand
//     slot holds as long as A and B agree: slot (g, i) of issue h is k = 64 c + 16 g + 8 h + i.  x is read in exactly
//     that order.
//     When M * K fits 64 KB, x * gamma is staged once in LDS instead (the first weight requests are already in flight).
//   * Two register sets of four chunks each (128 B per lane and set): the next round streams while this one is
//     multiplied.
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

#ifndef F16_H
// 16-byte pieces a lane loads per chunk (contiguous): a chunk is 32 * F16_H halves of K.  Measured on MI355X at the
// Llama-2-7B shapes (M = 1): 2 pieces -> 4.6 / 3.4 / 4.3 / 3.6 TB/s for qkv / o / gate+up / down, 4 pieces -> 3.4 / 2.6 /
// 3.0 / 2.9 TB/s: what counts is how close together the 64 addresses of ONE request are (pieces of a row 32 B apart
// with 2, 64 B apart with 4), not how much of a row a lane covers over several requests.
#define F16_H 2
#endif
constexpr int F16_NW = 8;      // waves per workgroup
constexpr int F16_U = 8 / F16_H;   // chunks per register set and wave
constexpr int F16_CK = 32 * F16_H;     // halves of K per chunk
constexpr int F16_LK = 8 * F16_H;      // contiguous halves per lane and chunk

// Weight fragments of one round (F16_U chunks) of one wave
template <int NACC>
struct F16Round {
  uint4v w[NACC][F16_U][F16_H];
};

template <int NACC>
__device__ __forceinline__ void f16_fetch(F16Round<NACC>& t, const _Float16* wrow, const _Float16* w2row, int c0, int nchunks) {
#pragma unroll
  for (int u = 0; u < F16_U; ++u) {
    const int c = min(c0 + u * F16_NW, nchunks - 1);
#pragma unroll
    for (int h = 0; h < F16_H; ++h) {
      t.w[0][u][h] = __builtin_nontemporal_load((const uint4v*)(wrow + c * F16_CK + 8 * h));
      if constexpr (NACC == 2) t.w[1][u][h] = __builtin_nontemporal_load((const uint4v*)(w2row + c * F16_CK + 8 * h));
    }
  }
}

// XLDS: x (already multiplied by gamma) is staged once in LDS, rows of K + 8 halves; otherwise every wave reads its
// x fragments from global memory (row counts whose x does not fit the staging budget).
template <bool GATE_UP, bool NORM, bool XLDS>
__global__ __launch_bounds__(F16_NW * 64) void gemv_f16_kernel(GemvF16Args a) {
  constexpr int NACC = GATE_UP ? 2 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int nchunks = a.K / F16_CK;
  const _Float16* wrow = a.w + (int64_t)(n0 + r) * a.K + F16_LK * g;
  const _Float16* w2row = GATE_UP ? a.w2 + (int64_t)(n0 + r) * a.K + F16_LK * g : nullptr;
  const _Float16* xrow = a.x + (int64_t)min(r, a.M - 1) * a.ldx + F16_LK * g;
  const _Float16* grow = NORM ? a.gamma + F16_LK * g : nullptr;

  __shared__ float4v red[F16_NW][NACC][64];
  __shared__ float ssq_sh[F16_NW][16];
  extern __shared__ _Float16 xs[];
  const int XS = a.K + 8;

  float4v acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = float4v{0, 0, 0, 0};
  float ssq = 0.0f;

  F16Round<NACC> ta, tb;
  f16_fetch<NACC>(ta, wrow, w2row, wave, nchunks);          // the weight stream starts before x is staged

  if constexpr (XLDS) {
    const int per_row = a.K >> 3;                            // 16-byte pieces per row
    for (int m = 0; m < a.M; ++m) {
      float s = 0.0f;
      for (int i = tid; i < per_row; i += F16_NW * 64) {
        H8 xv;
        xv.q = *(const uint4v*)(a.x + (int64_t)m * a.ldx + i * 8);
        if constexpr (NORM) {
          s += frag_sumsq_f16(xv.v);
          H8 gm;
          gm.q = *(const uint4v*)(a.gamma + i * 8);
          half8 gmv = gm.v;
          if (a.gamma_offset != 0.0f) gmv = gmv + (_Float16)a.gamma_offset;
          xv.v = xv.v * gmv;
        }
        *(uint4v*)(xs + m * XS + i * 8) = xv.q;
      }
      if constexpr (NORM) {
        s = wave_sum_dpp(s);
        if (lane == 0) ssq_sh[wave][m] = s;
      }
    }
    __syncthreads();
  }
  const _Float16* xl = xs + min(r, a.M - 1) * XS + F16_LK * g;

  auto consume = [&](const F16Round<NACC>& t, int c0) {
#pragma unroll
    for (int u = 0; u < F16_U; ++u) {
      const int cu = c0 + u * F16_NW;
      const bool live = cu < nchunks;                        // (wave-uniform) a chunk past the end contributes zeros
      const int c = min(cu, nchunks - 1);
#pragma unroll
      for (int h = 0; h < F16_H; ++h) {
        H8 xa;
        if constexpr (XLDS) {
          xa.q = *(const uint4v*)(xl + c * F16_CK + 8 * h);
        } else {
          xa.q = *(const uint4v*)(xrow + c * F16_CK + 8 * h);
        }
        if (!live) xa.q = uint4v{0, 0, 0, 0};
        if constexpr (NORM && !XLDS) {
          ssq += frag_sumsq_f16(xa.v);
          H8 gm;
          gm.q = *(const uint4v*)(grow + c * F16_CK + 8 * h);
          half8 gmv = gm.v;
          if (a.gamma_offset != 0.0f) gmv = gmv + (_Float16)a.gamma_offset;
          xa.v = xa.v * gmv;
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          H8 b;
          b.q = t.w[i][u][h];
          acc[i] = mfma16(xa.v, b.v, acc[i]);
        }
      }
    }
  };

  // two register sets: the next round's weights are in flight while this one is multiplied
  constexpr int STEP = F16_NW * F16_U;
  for (int c0 = wave; c0 < nchunks; c0 += 2 * STEP) {
    if (c0 + STEP < nchunks) f16_fetch<NACC>(tb, wrow, w2row, c0 + STEP, nchunks);
    consume(ta, c0);
    if (c0 + STEP < nchunks) {
      if (c0 + 2 * STEP < nchunks) f16_fetch<NACC>(ta, wrow, w2row, c0 + 2 * STEP, nchunks);
      consume(tb, c0 + STEP);
    }
  }

  // ---- cross-wave reduction: lane holds C[m = 4g + q][n = r], q = 0..3
#pragma unroll
  for (int i = 0; i < NACC; ++i) red[wave][i][lane] = acc[i];
  if constexpr (NORM && !XLDS) {
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

// ---- row-stream form (M <= 2) ------------------------------------------------------------------------------------
// With one or two rows of x the MFMA has nothing to amortise, and what the weight stream wants is requests whose 64
// addresses are one contiguous KiB.  Here a wavefront owns ONE weight row (the gate and the up row of one column in the
// pair form): lane l of request j loads the 16 bytes at k = 512 j + 8 l, all TC requests of the row are issued at the top
// of the kernel with no branch around them (a branch around a load makes every later wait a full one), then x is staged
// once per workgroup in LDS (already scaled by gamma, zeros from K on), and each request meets its 8 halves of x in four
// v_dot2_f32_f16 per row of x.  A cross-lane sum ends the row.  Eight rows per workgroup, ceil(N / 8) workgroups: the
// shape in which a plain streaming read is fastest on this chip (benchmark/hbm_stream_probe.hip: 512 threads, 8 requests
// of 16 bytes per lane in flight).  Same prologue / epilogue fusions as the MFMA form above.
constexpr int FR_NW = 8;

__device__ __forceinline__ float dot8_f16(uint4v w, uint4v x, float acc) {
  H8 a, b;
  a.q = w;
  b.q = x;
#pragma unroll
  for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2(a.h2[i], b.h2[i], acc, false);
  return acc;
}

// TC: 1 KiB requests per weight row (K <= 512 TC)
template <int MR, int TC, bool GATE_UP, bool NORM>
__global__ __launch_bounds__(FR_NW * 64) void gemv_f16_rows_kernel(GemvF16Args a) {
  constexpr int NACC = GATE_UP ? 2 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  extern __shared__ _Float16 xs[];                         // [MR][XS]: x * gamma, zeros from K on
  const int K512 = (a.K + 511) & ~511, XS = K512 + 8;
  __shared__ float ssq_sh[FR_NW][MR];

  const int nraw = blockIdx.x * FR_NW + wave;
  const int n = min(nraw, a.N - 1);                        // (a wavefront past the last row repeats it and stores nothing)

  // ---- requests, in the order in which their data is wanted (vector memory returns in order): x and gamma (out of L2),
  // the bias, then the weight row.  All unconditional: absent operands point at x, indices past the end are clamped.
  constexpr int XJ = (TC + 7) / 8;                         // 16-byte pieces of an x row per thread
  const int per_row = a.K >> 3;
  H8 xv[MR][XJ], gm[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int i = min(tid + j * (FR_NW * 64), per_row - 1);
#pragma unroll
    for (int m = 0; m < MR; ++m) xv[m][j].q = *(const uint4v*)(a.x + (int64_t)m * a.ldx + i * 8);
    if constexpr (NORM) gm[j].q = *(const uint4v*)(a.gamma + i * 8);
  }
  const _Float16 b1h = *(a.bias ? a.bias + n : a.x);
  const _Float16 b2h = *((GATE_UP && a.bias2) ? a.bias2 + n : a.x);
  uint4v w[NACC][TC];
  {
    const int64_t base = (int64_t)n * a.K;
#pragma unroll
    for (int u = 0; u < TC; ++u) {
      // past the end of a row (K % 512 != 0, or K <= 512 (TC - 1)) the address stays on the row's last 16 bytes and x
      // reads as zero there
      const int k = min(u * 512 + lane * 8, a.K - 8);
      w[0][u] = __builtin_nontemporal_load((const uint4v*)(a.w + base + k));
      if constexpr (GATE_UP) w[1][u] = __builtin_nontemporal_load((const uint4v*)(a.w2 + base + k));
    }
  }

  // ---- x (times gamma) into LDS, sums of squares per row.  A piece index past the row was clamped for the load and is
  // clamped for the store too (it rewrites the row's last piece with the same value): every loaded value is used
  // unconditionally, so the compiler cannot sink a load into a branch and wait for the whole weight row there.
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int i = tid + j * (FR_NW * 64), ic = min(i, per_row - 1);
      H8 v = xv[m][j];
      if constexpr (NORM) {
        s += i < per_row ? frag_sumsq_f16(v.v) : 0.0f;
        half8 gmv = gm[j].v;
        if (a.gamma_offset != 0.0f) gmv = gmv + (_Float16)a.gamma_offset;
        v.v = v.v * gmv;
      }
      *(uint4v*)(xs + m * XS + ic * 8) = v.q;
    }
    for (int z = per_row + tid; z * 8 < XS; z += FR_NW * 64) *(uint4v*)(xs + m * XS + z * 8) = uint4v{0, 0, 0, 0};
    if constexpr (NORM) {
      s = wave_sum_dpp(s);
      if (lane == 0) ssq_sh[wave][m] = s;
    }
  }
  __syncthreads();

  float acc[NACC][MR];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[i][m] = 0.0f;
#pragma unroll
  for (int u = 0; u < TC; ++u) {
    const int k = min(u * 512 + lane * 8, K512);           // zeros from K on
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const uint4v xa = *(const uint4v*)(xs + m * XS + k);
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i][m] = dot8_f16(w[i][u], xa, acc[i][m]);
    }
  }

  // ---- cross-lane sums, epilogue by lane m for row m of x
  const float b1 = a.bias ? (float)b1h : 0.0f;
  const float b2 = (GATE_UP && a.bias2) ? (float)b2h : 0.0f;
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    float rs = 1.0f;
    if constexpr (NORM) {
      float s = 0.0f;
#pragma unroll
      for (int ww = 0; ww < FR_NW; ++ww) s += ssq_sh[ww][m];
      rs = rsqrtf(s / (float)a.K + a.eps);
    }
    float v = wave_sum_dpp(acc[0][m]) * rs + b1;
    if constexpr (GATE_UP) v = act_apply(a.act, v) * (wave_sum_dpp(acc[1][m]) * rs + b2);
    else v = act_apply(a.act, v);
    if (lane == m && nraw < a.N) {
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
}

static int g_rows_form = 1;        // bigdl_gemv_f16_tune: 0 = MFMA form for every row count

static size_t f16_rows_lds(int M, int K) { return (size_t)M * (((K + 511) & ~511) + 8) * 2; }
// The row-stream form takes one or two rows of x (measured at the Llama-2-7B shapes, us per launch, row-stream / MFMA form:
// 1 row 16.9 / 22.9 qkv, 30.3 / 41.4 gate + up, 17.7 / 25.1 down; 2 rows 20.7 / 25.8, 39.3 / 43.1, 20.5 / 26.1; 3 rows
// 22.9 / 29.1, 48.0 / 45.3, 35.4 / 35.5; 4 rows 34.5 / 31.1, 54.9 / 46.6, 36.0 / 36.2) with x inside 64 KB of LDS and
// weight rows of at most 32 requests (16 in the pair form).
static bool f16_rows_ok(int M, int K, bool pair) {
  return g_rows_form && M <= 2 && f16_rows_lds(M, K) <= 64 * 1024 && K <= 512 * (pair ? 16 : 32);
}

template <int MR, int TC, bool GATE_UP, bool NORM>
static int launch_f16_rows(const GemvF16Args& a, hipStream_t stream) {
  const int grid = (a.N + FR_NW - 1) / FR_NW;
  const size_t xbytes = f16_rows_lds(MR, a.K);
  auto k = gemv_f16_rows_kernel<MR, TC, GATE_UP, NORM>;
  static bool raised = false;                   // one opt-in per instantiation: dynamic LDS above the default
  if (!raised && xbytes > 32 * 1024) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) return -5;
    raised = true;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(FR_NW * 64), xbytes, stream, a);
  return HIP_CHECK_LAUNCH();
}

template <int MR, bool GATE_UP, bool NORM>
static int launch_f16_rows_tc(const GemvF16Args& a, hipStream_t stream) {
  const int t = (a.K + 511) / 512;
  if (t <= 8) return launch_f16_rows<MR, 8, GATE_UP, NORM>(a, stream);
  if (t <= 16) return launch_f16_rows<MR, 16, GATE_UP, NORM>(a, stream);
  if constexpr (!GATE_UP) {
    if (t <= 22) return launch_f16_rows<MR, 22, GATE_UP, NORM>(a, stream);
    if (t <= 28) return launch_f16_rows<MR, 28, GATE_UP, NORM>(a, stream);
    if (t <= 32) return launch_f16_rows<MR, 32, GATE_UP, NORM>(a, stream);
  }
  return -1;
}

template <bool GATE_UP, bool NORM>
static int launch_f16_rows_m(const GemvF16Args& a, hipStream_t stream) {
  if (a.M == 1) return launch_f16_rows_tc<1, GATE_UP, NORM>(a, stream);
  return launch_f16_rows_tc<2, GATE_UP, NORM>(a, stream);
}

template <bool GATE_UP, bool NORM>
static int launch_f16(const GemvF16Args& a, hipStream_t stream) {
  const dim3 grid(a.N / 16), block(F16_NW * 64);
  const size_t xbytes = (size_t)a.M * (a.K + 8) * 2;
  if (xbytes <= 64 * 1024) {
    auto k = gemv_f16_kernel<GATE_UP, NORM, true>;
    static bool raised = false;                 // one opt-in per instantiation: dynamic LDS above the default 48 KB
    if (!raised && xbytes > 32 * 1024) {
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) return -5;
      raised = true;
    }
    hipLaunchKernelGGL(k, grid, block, xbytes, stream, a);
  } else {
    hipLaunchKernelGGL((gemv_f16_kernel<GATE_UP, NORM, false>), grid, block, 0, stream, a);
  }
  return HIP_CHECK_LAUNCH();
}

}  // namespace bigdl

using namespace bigdl;

// Returns 0 on success; negative = rejected on the host before any launch (the caller falls back to a library GEMM).
// Requirements: M <= 16, N % 16 == 0, K % 64 == 0, x rows 16-byte aligned (ldx % 8 == 0), weights 16-byte aligned.
extern "C" int bigdl_gemv_f16(const void* x, const void* w, const void* w2, void* out, const void* bias,
                              const void* bias2, const void* residual, const void* gamma, float eps, float gamma_offset,
                              int M, int N, int K, int ldx, int ldo, int act, int out_type, hipStream_t stream) {
  if (M <= 0 || M > 16 || N <= 0 || (N & 15) || K <= 0 || (K % F16_CK) || (ldx & 7) || ldx < K || ldo < N) return -1;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || (w2 && ((uintptr_t)w2 & 15)) || (gamma && ((uintptr_t)gamma & 15)))
    return -1;
  if (out_type < 0 || out_type > 2) return -3;
  GemvF16Args a;
  a.x = (const _Float16*)x; a.w = (const _Float16*)w; a.w2 = (const _Float16*)w2; a.out = out;
  a.bias = (const _Float16*)bias; a.bias2 = (const _Float16*)bias2; a.residual = residual;
  a.gamma = (const _Float16*)gamma; a.eps = eps; a.gamma_offset = gamma_offset;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.act = act; a.out_type = out_type;
  if (f16_rows_ok(M, K, w2 != nullptr)) {
    if (w2) return gamma ? launch_f16_rows_m<true, true>(a, stream) : launch_f16_rows_m<true, false>(a, stream);
    return gamma ? launch_f16_rows_m<false, true>(a, stream) : launch_f16_rows_m<false, false>(a, stream);
  }
  if (w2) return gamma ? launch_f16<true, true>(a, stream) : launch_f16<true, false>(a, stream);
  return gamma ? launch_f16<false, true>(a, stream) : launch_f16<false, false>(a, stream);
}

// Measurement hook (benchmark/gemv_f16_probe.py): rows_form 0 sends every row count to the MFMA form.
extern "C" void bigdl_gemv_f16_tune(int rows_form) { g_rows_form = rows_form; }
