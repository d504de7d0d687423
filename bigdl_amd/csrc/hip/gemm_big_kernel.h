// Dequant-fused MFMA GEMM for prompt-sized row counts (M >= 256) on gfx950:  y[M,N] = f(x[M,K] . dequant(W)[N,K]^T).
//
// The prompt-side counterpart of the reference's closed xe_linear.forward_new
// (python/llm/src/ipex_llm/transformers/low_bit_linear.py:657-668): the weights stay in their low-bit T16 tiles and are
// expanded in registers, so a prompt pass neither writes nor re-reads an fp16 copy of the model.
//
// Why a third kernel next to gemm_kernel.h: with 128 x 128 output tiles and every wavefront walking the whole x tile,
// that kernel reads 8 KiB of A fragments from LDS per 16 MFMAs per wavefront -- 128 bytes per clock per CU at the MFMA
// rate, which is all the LDS delivers, so it levels off near 650 TFLOP/s.  Here a workgroup owns a 128 x 256 tile:
//  * each of the 4 wavefronts owns 64 output columns (4 weight tiles), so one A fragment feeds 4 MFMAs: 32 MFMAs per
//    8 KiB of LDS reads, half the LDS traffic per flop;
//  * the x tile [128][128 k] goes global -> LDS directly (global_load_lds, 16 bytes per lane, no staging registers:
//    the 128 accumulator registers leave no room for them); the XOR swizzle of the 16-byte slots (slot ^= row & 15)
//    is applied on the source address, the LDS image itself is lane-linear as the instruction requires;
//  * weight tiles stream HBM -> VGPR in MFMA B-operand order as in the other kernels, one k-tile ahead, and are
//    dequantised once per workgroup column (no redundancy: every weight is expanded by exactly one wavefront per
//    128-row block);
//  * small grids split K (fp32 slabs + the shared reduce / epilogue kernel).
#pragma once
#include "gemm_kernel.h"

// BIG_ABL (measurement builds, results wrong): 1 = no weight loads in the k loop, 2 = no x tile loads in the k loop,
// 4 = no dequantization, 8 = no barrier in the k loop, 16 = no MFMA, 32 = no LDS fragment reads
#ifndef BIG_ABL
#define BIG_ABL 0
#endif

namespace bigdl {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_cvoid_t;

template <int QT, int NT>
__global__ __launch_bounds__(256, 2) void gemm_t16_big_kernel(GemmArgs a) {
  constexpr int MT = 8, BM = 128, BN = 64 * NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                                               // [2][BM][256 B]
  _Float16* nf4_lut = (_Float16*)(smem + 2 * BM * 256);           // [64][16] (nf4 only)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int KT = a.K >> 7;                                       // host: K % 128 == 0
  const int nblocks = (a.N + BN - 1) / BN;
  const int nb = blockIdx.x % nblocks, sp = blockIdx.x / nblocks;
  const int mb = blockIdx.y;
  const int row0 = mb * BM, rows = min(BM, a.M - row0);
  const int kt0 = sp * a.kt_per_split, kt1 = min(KT, kt0 + a.kt_per_split);
  const T16Planes pl = t16_planes(a.w, QT, a.N, a.K);

  const _Float16* lut = nullptr;
  if constexpr (QT == Q_NF4) {
    fill_nf4_banked((uint32_t*)nf4_lut, threadIdx.x, 256);
    lut = (const _Float16*)((const uint32_t*)nf4_lut + (lane & 31));
  }

  const int ntiles = (a.N + 15) >> 4;
  int nt_idx[NT];
  int64_t tbase[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    nt_idx[n] = nb * (BN / 16) + wave * NT + n;
    tbase[n] = (int64_t)min(nt_idx[n], ntiles - 1) * KT;
  }

  // x tile: pass p moves rows 16 p .. 16 p + 15; thread t brings the 16 bytes that belong in LDS slot t % 16 of row
  // 16 p + t / 16, i.e. k slot (t % 16) ^ (t / 16) of that row (rows past the block re-read its last row: never stored)
  const int srow = threadIdx.x >> 4, sslot = threadIdx.x & 15;
  uint32_t xoff[MT];
#pragma unroll
  for (int p = 0; p < MT; ++p)
    xoff[p] = (uint32_t)((int64_t)(row0 + min(16 * p + srow, rows - 1)) * a.ldx) + (uint32_t)(((sslot ^ srow) & 15) << 3);
  auto stage_a = [&](int kt, int stage) {
    const _Float16* src = a.x + kt * 128;
    char* dst = As + stage * (BM * 256) + wave * 1024;            // wave-uniform; the instruction adds lane * 16
#pragma unroll
    for (int p = 0; p < MT; ++p)
      __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(src + xoff[p]), (lds_void_t*)(dst + p * 4096), 16, 0, 0);
  };

  float4v acc[MT][NT];
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[j][n] = float4v{0, 0, 0, 0};

  Tile<QT> tt[3][NT];                     // ring: the tile in use, and the next two on their way
  auto fetch_b = [&](Tile<QT> (&t)[NT], int kt) {
#pragma unroll
    for (int n = 0; n < NT; ++n) load_tile<QT>(t[n], pl, tbase[n] + kt, lane);
  };
  auto dequant = [&](Tile<QT> (&t)[NT], int s, half8 (&bf)[NT]) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      if constexpr (BIG_ABL & 4) {
        bf[n] = __builtin_bit_cast(half8, uint4v{t[n].lo0[s], t[n].lo0[(s + 1) & 3], 0x3c003c00u, 0x3c003c00u});
        continue;
      }
      const half8 code = decode_step<QT>(t[n], s, lut);
      _Float16 d = t[n].d[s];
      if constexpr (post_scale<QT>() != 1.0f) d = (_Float16)fminf((float)d * post_scale<QT>(), 65504.0f);
      if constexpr (has_min<QT>()) bf[n] = code * d + t[n].m[s];
      else bf[n] = code * d;
    }
  };
  // One k-tile = four 32-deep steps.  Per step: the 8 A fragments are requested first, the NEXT step's weight fragments
  // are expanded while those LDS reads are in flight, then the step's MT * NT MFMAs run back to back.  (Left to itself the
  // compiler sinks every pair of LDS reads to just before its MFMAs and waits lgkmcnt(0) each time: six MFMAs per
  // exposed LDS round trip.)
  auto compute = [&](Tile<QT> (&t)[NT], int stage) {
    const char* abase = As + stage * (BM * 256) + r * 256;
    half8 bf[2][NT];
    dequant(t, 0, bf[0]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int aoff = (((4 * s + g) ^ r) & 15) << 4;
      half8 af[MT];
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        if constexpr (BIG_ABL & 32) af[j] = __builtin_bit_cast(half8, uint4v{(uint32_t)(aoff + j), 0x3c003c00u, 0x3c003c00u, 0x3c003c00u});
        else af[j] = *(const half8*)(abase + j * 16 * 256 + aoff);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s < 3) dequant(t, s + 1, bf[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < MT; ++j) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          if constexpr (BIG_ABL & 16) {
            const uint4v x = __builtin_bit_cast(uint4v, af[j]) ^ __builtin_bit_cast(uint4v, bf[s & 1][n]);
            acc[j][n][0] += __uint_as_float((x[0] ^ x[1] ^ x[2] ^ x[3]) & 0x3f800000u);
          } else {
            acc[j][n] = mfma16(bf[s & 1][n], af[j], acc[j][n]);     // D[n = 4 g + q][m = r]
          }
        }
      }
    }
  };

  // Pipeline.  Iteration k requests x tile k + 1 (LDS stage (k + 1) & 1) and weight tiles k + 2 (ring slot (k + 2) % 3),
  // then works on tile k.  Vector memory completes in issue order, so waiting until only the NB youngest operations
  // (the weight loads just issued) are outstanding retires x tile k + 1 and weight tiles k + 1; the weight tiles get two
  // iterations to arrive (they come from HBM, cold, exactly once per 128-row block).  The waits are
  // __builtin_amdgcn_s_waitcnt, which the compiler's own wait insertion sees: a __syncthreads() here would drain
  // everything (vmcnt(0)) because an LDS-DMA is in flight.
  constexpr int NB = NT * (has_min<QT>() ? 3 : 2);       // per tile: codes, scales (+ mins); 4-bit formats only
  static_assert(NB < 16, "vmcnt immediate below assumes a 4-bit count");
  constexpr int WAIT_NB = NB | (0x7 << 4) | (0xF << 8);      // vmcnt(NB), expcnt / lgkmcnt untouched
  const int klast = max(kt1 - 1, kt0);
  stage_a(kt0, 0);
  fetch_b(tt[0], kt0);
  fetch_b(tt[1], min(kt0 + 1, klast));
  __builtin_amdgcn_s_waitcnt(WAIT_NB);
  __builtin_amdgcn_s_barrier();
  for (int kt = kt0; kt < kt1; kt += 6) {
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      const int k = kt + d;                 // block-uniform; (k - kt0) % 6 == d
      if (k >= kt1) break;
      if constexpr (!(BIG_ABL & 2)) stage_a(min(k + 1, klast), (d + 1) & 1);
      if constexpr (!(BIG_ABL & 1)) fetch_b(tt[(d + 2) % 3], min(k + 2, klast));
      // (without this the scheduler sinks the requests to the end of the iteration: nothing uses them before the barrier)
      __builtin_amdgcn_sched_barrier(0);
      compute(tt[(BIG_ABL & 1) ? 0 : d % 3], d & 1);
      __builtin_amdgcn_s_waitcnt(WAIT_NB);
      if constexpr (!(BIG_ABL & 8)) __builtin_amdgcn_s_barrier();
    }
  }
  __builtin_amdgcn_s_waitcnt(0 | (0x7 << 4) | (0xF << 8));     // the clamped look-ahead loads of the last iterations

  // epilogue (same contract as gemm_t16_kernel's dense mode).  The weight tile is the MFMA's A operand, so a lane holds
  // 4 consecutive output columns (n = 16 tile + 4 g + q) of one row (m = 16 j + r): 8-byte fp16 / 16-byte fp32 stores
  // instead of 128 scattered 2-byte ones per lane (which made the store tail longer than the k loop).
  const int64_t mtot = a.M;
  const bool partial_out = a.splitk > 1;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int ncol = nt_idx[n] * 16 + 4 * g;
    if (nt_idx[n] >= ntiles || ncol >= a.N) continue;
    const bool vec = ncol + 4 <= a.N && (a.ldo & 3) == 0;
    float bias[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (a.bias != nullptr && !partial_out) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ncol + q < a.N) bias[q] = (float)((const _Float16*)a.bias)[ncol + q];
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = 16 * j + r;
      if (m >= rows) continue;
      const int64_t orow = row0 + m;
      if (partial_out) {
        float* dst = a.partial + ((int64_t)sp * mtot + orow) * a.N + ncol;
        if (ncol + 4 <= a.N && (a.N & 3) == 0) *(float4v*)dst = acc[j][n];
        else {
#pragma unroll
          for (int q = 0; q < 4; ++q)           // (static register indices: a run-time index would put acc in scratch)
            if (ncol + q < a.N) dst[q] = acc[j][n][q];
        }
        continue;
      }
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = act_apply(a.act, acc[j][n][q] + bias[q]);
      const int64_t oi = orow * a.ldo + ncol;
      if (a.out_type == OUT_F16 && vec) {
        if (a.residual) {
          const half4 rr = *(const half4*)((const _Float16*)a.residual + oi);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += (float)rr[q];
        }
        *(half4*)((_Float16*)a.out + oi) = half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        continue;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (ncol + q >= a.N) continue;
        float w = v[q];
        if (a.out_type == OUT_F16) {
          if (a.residual) w += load_out<OUT_F16>(a.residual, oi + q);
          store_out<OUT_F16>(a.out, oi + q, w);
        } else if (a.out_type == OUT_BF16) {
          if (a.residual) w += load_out<OUT_BF16>(a.residual, oi + q);
          store_out<OUT_BF16>(a.out, oi + q, w);
        } else {
          if (a.residual) w += load_out<OUT_F32>(a.residual, oi + q);
          store_out<OUT_F32>(a.out, oi + q, w);
        }
      }
    }
  }
}

template <int QT, int NT>
static int gemm_big_launch(const GemmArgs& a, int mblocks, hipStream_t st) {
  const size_t lds = 2 * 128 * 256 + (QT == Q_NF4 ? 64 * 16 * 2 : 0);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_t16_big_kernel<QT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return -101;
    attr_set = true;
  }
  const int nblocks = (a.N + 64 * NT - 1) / (64 * NT);
  hipLaunchKernelGGL((gemm_t16_big_kernel<QT, NT>), dim3(nblocks * a.splitk, mblocks), dim3(256), lds, st, a);
  return HIP_CHECK_LAUNCH();
}

}  // namespace bigdl
