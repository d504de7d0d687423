// Rank-r adapter products of LoRA / QLoRA training on gfx950, bound by the bytes of the one big operand:
//  * lora_down: T[M, R] = alpha * X[M, K] . W^T     (t = dropout(x) A^T in the forward pass, gt = s * gy B in the backward pass)
//  * lora_up:   OUT[M, C] = beta * OUT + alpha * mask * (T[M, R] . W)   (y += s * t B^T; gx += (gt A) * dropout mask / (1 - p))
// R = 8 or 16.  A library GEMM runs these shapes (a reduction or an output dimension of 8) as 16 x 16 MFMA tiles at a
// few percent of the memory rate, and the dropout mask of the backward pass costs two more passes over [M, K] (cast of the
// mask, multiply-add); here each product is one pass over X / OUT with the small operand in registers.
// W is [R, K] row-major (WT = false: an adapter's A matrix) or [K, R] row-major (WT = true: its B matrix); no transposed
// copies are made.  Math of the reference's LoraLowBitLinear.forward (python/llm/src/ipex_llm/transformers/qlora.py:57-105).
#include "common.h"

namespace bigdl {

template <int DT>
__device__ __forceinline__ void ld8(const void* p, int64_t i, float (&v)[8]) {
  const uint4v u = *(const uint4v*)((const uint16_t*)p + i);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if constexpr (DT == OUT_BF16) {
      v[2 * q] = __uint_as_float(u[q] << 16);
      v[2 * q + 1] = __uint_as_float(u[q] & 0xffff0000u);
    } else {
      union { uint32_t w; _Float16 h[2]; } c;
      c.w = u[q];
      v[2 * q] = (float)c.h[0];
      v[2 * q + 1] = (float)c.h[1];
    }
  }
}
template <int DT>
__device__ __forceinline__ uint16_t to16(float f) {
  if constexpr (DT == OUT_BF16) {
    return f32_to_bf16_bits(f);
  } else {
    union { _Float16 h; uint16_t b; } c;
    c.h = (_Float16)f;
    return c.b;
  }
}
template <int DT>
__device__ __forceinline__ void st8(void* p, int64_t i, const float (&v)[8]) {
  uint4v u;
#pragma unroll
  for (int q = 0; q < 4; ++q) u[q] = (uint32_t)to16<DT>(v[2 * q]) | ((uint32_t)to16<DT>(v[2 * q + 1]) << 16);
  *(uint4v*)((uint16_t*)p + i) = u;
}

// the lane's 8 columns c .. c + 7 of all R rows of the small operand -> w[j][i]
template <int R, bool WT, int DT>
__device__ __forceinline__ void load_small(const void* W, int64_t ldw, int c, float (&w)[R][8]) {
  if constexpr (!WT) {
#pragma unroll
    for (int j = 0; j < R; ++j) ld8<DT>(W, j * ldw + c, w[j]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j0 = 0; j0 < R; j0 += 8) {
        float t[8];
        ld8<DT>(W, (int64_t)(c + i) * ldw + j0, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j0 + j][i] = t[j];
      }
  }
}

constexpr int LORA_RB = 4;   // rows per workgroup
constexpr int LORA_DW = 8;   // wavefronts of a lora_down workgroup (they split K)

template <int DT>
__device__ __forceinline__ void unpack8(const uint4v u, float (&v)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if constexpr (DT == OUT_BF16) {
      v[2 * q] = __uint_as_float(u[q] << 16);
      v[2 * q + 1] = __uint_as_float(u[q] & 0xffff0000u);
    } else {
      union { uint32_t w; _Float16 h[2]; } c;
      c.w = u[q];
      v[2 * q] = (float)c.h[0];
      v[2 * q + 1] = (float)c.h[1];
    }
  }
}

// one workgroup = LORA_RB rows; its eight wavefronts split K (one 512-column chunk per wavefront and pass: K = 4096 is a
// single memory round trip), the requests of the next pass are issued before the products of the current one; every
// lane keeps LORA_RB x R partial sums
template <int R, bool WT, int DT>
__global__ __launch_bounds__(64 * LORA_DW) void lora_down_kernel(const void* X, const void* W, void* T, int M, int K, int64_t ldx,
                                                               int64_t ldw, float alpha) {
  __shared__ float red[LORA_DW][LORA_RB][R];
  constexpr int NWV = WT ? 8 * (R / 8) : R;        // 16-byte requests of the small operand per chunk
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * LORA_RB;
  const uint16_t* Xp = (const uint16_t*)X;
  const uint16_t* Wp = (const uint16_t*)W;
  float acc[LORA_RB][R];
#pragma unroll
  for (int rb = 0; rb < LORA_RB; ++rb)
#pragma unroll
    for (int j = 0; j < R; ++j) acc[rb][j] = 0.0f;
  uint4v xr[2][LORA_RB], wr[2][NWV];
  auto request = [&](int buf, int c) {
    const int cc = min(c, K - 8);                  // past the end: a valid address, the products are skipped
#pragma unroll
    for (int rb = 0; rb < LORA_RB; ++rb) xr[buf][rb] = *(const uint4v*)(Xp + (int64_t)min(row0 + rb, M - 1) * ldx + cc);
    if constexpr (!WT) {
#pragma unroll
      for (int j = 0; j < R; ++j) wr[buf][j] = *(const uint4v*)(Wp + j * ldw + cc);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j0 = 0; j0 < R / 8; ++j0) wr[buf][i * (R / 8) + j0] = *(const uint4v*)(Wp + (int64_t)(cc + i) * ldw + 8 * j0);
    }
  };
  auto consume = [&](int buf) {
    float w[R][8];
    if constexpr (!WT) {
#pragma unroll
      for (int j = 0; j < R; ++j) unpack8<DT>(wr[buf][j], w[j]);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j0 = 0; j0 < R / 8; ++j0) {
          float t[8];
          unpack8<DT>(wr[buf][i * (R / 8) + j0], t);
#pragma unroll
          for (int j = 0; j < 8; ++j) w[8 * j0 + j][i] = t[j];
        }
    }
#pragma unroll
    for (int rb = 0; rb < LORA_RB; ++rb) {
      float x[8];
      unpack8<DT>(xr[buf][rb], x);
#pragma unroll
      for (int j = 0; j < R; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[rb][j] += x[i] * w[j][i];
    }
  };
  constexpr int STEP = LORA_DW * 64 * 8;
  int c = (wave * 64 + lane) * 8;
  request(0, c);
  for (; c < K; c += 2 * STEP) {
    request(1, c + STEP);
    consume(0);
    if (c + STEP < K) {
      request(0, c + 2 * STEP);
      consume(1);
    }
  }
#pragma unroll
  for (int rb = 0; rb < LORA_RB; ++rb)
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const float s = wave_sum(acc[rb][j]);
      if (lane == 0) red[wave][rb][j] = s;
    }
  __syncthreads();
  if (threadIdx.x < LORA_RB * R) {
    const int rb = threadIdx.x / R, j = threadIdx.x % R;
    if (row0 + rb < M) {
      float s = 0.0f;
#pragma unroll
      for (int w = 0; w < LORA_DW; ++w) s += red[w][rb][j];
      ((uint16_t*)T)[(int64_t)(row0 + rb) * R + j] = to16<DT>(alpha * s);
    }
  }
}

// one workgroup = LORA_RB rows x 2048 columns; a lane keeps the small operand's 8 columns for all R
template <int R, bool WT, bool MASK, int DT>
__global__ __launch_bounds__(256) void lora_up_kernel(const void* T, const void* W, void* OUT, const uint8_t* mask, int M, int C,
                                                      int64_t ldo, int64_t ldw, int64_t ldm, float alpha, int accumulate) {
  __shared__ float ts[LORA_RB][R];
  const int row0 = blockIdx.y * LORA_RB;
  if (threadIdx.x < LORA_RB * R) {
    const int rb = threadIdx.x / R, j = threadIdx.x % R;
    const uint16_t b = ((const uint16_t*)T)[(int64_t)min(row0 + rb, M - 1) * R + j];
    float v;
    if constexpr (DT == OUT_BF16) {
      v = bf16_bits_to_f32(b);
    } else {
      union { uint16_t b; _Float16 h; } cv;
      cv.b = b;
      v = (float)cv.h;
    }
    ts[rb][j] = v * alpha;
  }
  __syncthreads();
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= C) return;
  float w[R][8];
  load_small<R, WT, DT>(W, ldw, c, w);
#pragma unroll
  for (int rb = 0; rb < LORA_RB; ++rb) {
    const int row = row0 + rb;
    if (row >= M) break;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const float t = ts[rb][j];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] += t * w[j][i];
    }
    if constexpr (MASK) {
      const uint2v mk = *(const uint2v*)(mask + (int64_t)row * ldm + c);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (((i < 4 ? mk.x : mk.y) >> (8 * (i & 3)) & 0xffu) == 0) o[i] = 0.0f;
    }
    if (accumulate) {
      float prev[8];
      ld8<DT>(OUT, (int64_t)row * ldo + c, prev);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] += prev[i];
    }
    st8<DT>(OUT, (int64_t)row * ldo + c, o);
  }
}

}  // namespace bigdl

using namespace bigdl;

// T[M, R] = alpha * X[M, K] . W^T;  W [R, K] (wt = 0, row stride ldw) or [K, R] (wt = 1).  dt: 0 fp16, 1 bf16 (all operands).
// K % 8 == 0, rows of X 16-byte aligned, R in {8, 16}.
extern "C" int bigdl_lora_down(const void* X, const void* W, void* T, int M, int K, int R, int64_t ldx, int64_t ldw, int wt,
                               float alpha, int dt, hipStream_t st) {
  if (M <= 0 || K <= 0 || (K & 7) || (ldx & 7) || (R != 8 && R != 16) || (dt != OUT_F16 && dt != OUT_BF16) ||
      ((uintptr_t)X & 15) || ((uintptr_t)W & 15) || (wt ? ldw < R || (ldw & 7) : ldw < K || (ldw & 7)))
    return -1;
  const dim3 grid((M + LORA_RB - 1) / LORA_RB), block(64 * LORA_DW);
#define BIGDL_LD(R_, WT_, DT_) hipLaunchKernelGGL((lora_down_kernel<R_, WT_, DT_>), grid, block, 0, st, X, W, T, M, K, ldx, ldw, alpha)
  if (R == 8) {
    if (wt) { if (dt == OUT_BF16) BIGDL_LD(8, true, OUT_BF16); else BIGDL_LD(8, true, OUT_F16); }
    else { if (dt == OUT_BF16) BIGDL_LD(8, false, OUT_BF16); else BIGDL_LD(8, false, OUT_F16); }
  } else {
    if (wt) { if (dt == OUT_BF16) BIGDL_LD(16, true, OUT_BF16); else BIGDL_LD(16, true, OUT_F16); }
    else { if (dt == OUT_BF16) BIGDL_LD(16, false, OUT_BF16); else BIGDL_LD(16, false, OUT_F16); }
  }
#undef BIGDL_LD
  return HIP_CHECK_LAUNCH();
}

// OUT[M, C] = (accumulate ? OUT : 0) + alpha * mask * (T[M, R] . W);  W [R, C] (wt = 0) or [C, R] (wt = 1); mask: bytes
// [M, C] (row stride ldm, 0 = dropped) or null.  C % 8 == 0, rows of OUT 16-byte aligned, rows of mask 8-byte aligned.
extern "C" int bigdl_lora_up(const void* T, const void* W, void* OUT, const void* mask, int M, int C, int R, int64_t ldo,
                             int64_t ldw, int64_t ldm, int wt, float alpha, int accumulate, int dt, hipStream_t st) {
  if (M <= 0 || M > 65535 * LORA_RB || C <= 0 || (C & 7) || (ldo & 7) || (R != 8 && R != 16) || (dt != OUT_F16 && dt != OUT_BF16) ||
      ((uintptr_t)OUT & 15) || ((uintptr_t)W & 15) || (wt ? ldw < R || (ldw & 7) : ldw < C || (ldw & 7)) ||
      (mask && ((ldm & 7) || ldm < C || ((uintptr_t)mask & 7))))
    return -1;
  const dim3 grid((C / 8 + 255) / 256, (M + LORA_RB - 1) / LORA_RB), block(256);
#define BIGDL_LU(R_, WT_, MK_, DT_)                                                                                       \
  hipLaunchKernelGGL((lora_up_kernel<R_, WT_, MK_, DT_>), grid, block, 0, st, T, W, OUT, (const uint8_t*)mask, M, C, ldo, ldw, \
                     ldm, alpha, accumulate)
#define BIGDL_LU_DT(R_, WT_, MK_) do { if (dt == OUT_BF16) BIGDL_LU(R_, WT_, MK_, OUT_BF16); else BIGDL_LU(R_, WT_, MK_, OUT_F16); } while (0)
#define BIGDL_LU_MK(R_, WT_) do { if (mask) BIGDL_LU_DT(R_, WT_, true); else BIGDL_LU_DT(R_, WT_, false); } while (0)
  if (R == 8) { if (wt) BIGDL_LU_MK(8, true); else BIGDL_LU_MK(8, false); }
  else { if (wt) BIGDL_LU_MK(16, true); else BIGDL_LU_MK(16, false); }
#undef BIGDL_LU_MK
#undef BIGDL_LU_DT
#undef BIGDL_LU
  return HIP_CHECK_LAUNCH();
}
