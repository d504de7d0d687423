// Dequant-fused skinny GEMM ("GEMV") for decode on gfx950:  y[M,N] = f(x[M,K] . dequant(W)[N,K]^T)
//
// Replaces the reference's closed xe_linear.forward_new / xe_batch.batch_forward /
// xe_linear.mlp_forward_xpu for small row counts (signatures: python/llm/src/ipex_llm/
// transformers/xpu_ops.py:23-37; call sites low_bit_linear.py:657-668, models/common.py:118-133).
//
// Design (MI355X-first, not a translation):
//  * one workgroup = one 16-row tile of W (all of K); its NW wavefronts split K tile-wise and are
//    combined through LDS.  N/16 workgroups spread the weight stream evenly over the 256 CUs
//    (decode GEMVs are a few tens of KB per CU: what matters is that every CU has its share of
//    16-byte loads in flight at once).
//  * weights go HBM -> VGPR directly (no LDS: each byte is used once); every lane owns one
//    32-element scale block per MFMA step, so dequantisation is register bit-tricks and the block
//    scale is applied in fp32 to the 16x16 MFMA result.
//  * the product runs on v_mfma_f32_16x16x32_f16 with x as the A operand (rows m) and W^T as the B
//    operand (columns n): M = 1..16 costs the same, which serves batch-1 decode, small batches,
//    speculative verification and prompt-lookup candidates with one code path.
//  * optional fusions: RMSNorm of x in the prologue (gamma folded into the A fragments, the
//    1/rms factor applied to the result), gate/up pair with act(g)*u epilogue, bias, residual add.
#pragma once
#include "common.h"

namespace bigdl {

struct GemvArgs {
  const _Float16* x;      // [M, ldx]
  const void* w;          // T16 buffer (gate weight in GATE_UP mode)
  const void* w2;         // up weight (GATE_UP) or null
  void* out;              // [M, ldo]
  const void* bias;       // [N] fp16 or null
  const void* bias2;
  const void* residual;   // [M, ldo] same dtype as out, or null
  const _Float16* gamma;  // [K] RMSNorm weight or null
  float eps;
  float gamma_offset;     // gemma-style (1 + w): 1.0f, else 0.0f
  int M, N, K;
  int ldx, ldo;
  int act;
  int out_type;
};

template <int QT>
struct Tile {
  // this lane's share of one 16x128 weight tile
  uint4v lo0, lo1;     // 4-bit: lo0 only; 8-bit: lo0 = steps 0,1  lo1 = steps 2,3
  uint32_t hi0, hi1;   // 1-bit plane: hi0 (4 bytes); 2-bit plane: hi0,hi1 (8 bytes)
  half4 d, m;          // block scales / mins of the lane's row for the 4 steps
};

template <typename T>
__device__ __forceinline__ T ldg_nt(const void* p) {
  // weights are streamed exactly once per call: non-temporal loads keep them from displacing x/L2 state
  return __builtin_nontemporal_load((const T*)p);
}

template <int QT>
__device__ __forceinline__ void load_tile(Tile<QT>& t, const T16Planes& p, int64_t tile, int lane, bool valid) {
  constexpr int LO = (QT == Q_SYM_INT8 || QT == Q_FP8_E4M3 || QT == Q_FP8_E5M2) ? 8 : 4;
  t.lo0 = t.lo1 = uint4v{0, 0, 0, 0};
  t.hi0 = t.hi1 = 0;
  t.d = t.m = half4{0, 0, 0, 0};
  if (!valid) return;
  if constexpr (LO == 4) {
    t.lo0 = ldg_nt<uint4v>(p.lo + tile * 1024 + lane * 16);
  } else {
    t.lo0 = ldg_nt<uint4v>(p.lo + tile * 2048 + lane * 16);
    t.lo1 = ldg_nt<uint4v>(p.lo + tile * 2048 + 1024 + lane * 16);
  }
  if constexpr (QT == Q_SYM_INT5 || QT == Q_ASYM_INT5) {
    t.hi0 = ldg_nt<uint32_t>(p.hi + tile * 256 + lane * 4);
  } else if constexpr (QT == Q_FP6) {
    const uint2v h = ldg_nt<uint2v>(p.hi + tile * 512 + lane * 8);
    t.hi0 = h.x;
    t.hi1 = h.y;
  }
  const int r = lane & 15;
  t.d = ldg_nt<half4>(p.sc + tile * 64 + r * 4);
  if constexpr (QT == Q_ASYM_INT4 || QT == Q_ASYM_INT5 || QT == Q_WOQ_INT4) t.m = ldg_nt<half4>(p.mn + tile * 64 + r * 4);
}

template <int QT>
__device__ __forceinline__ half8 decode_step(const Tile<QT>& t, int s, const _Float16* lut) {
  if constexpr (QT == Q_SYM_INT4) return dec_int4(t.lo0[s], 8.0f);
  else if constexpr (QT == Q_ASYM_INT4 || QT == Q_WOQ_INT4) return dec_int4(t.lo0[s], 0.0f);
  else if constexpr (QT == Q_FP4) return dec_fp4(t.lo0[s]);
  else if constexpr (QT == Q_NF4) return dec_nf4(t.lo0[s], lut);
  else if constexpr (QT == Q_SYM_INT5) return dec_int5(t.lo0[s], (t.hi0 >> (8 * s)) & 0xffu, 16.0f);
  else if constexpr (QT == Q_ASYM_INT5) return dec_int5(t.lo0[s], (t.hi0 >> (8 * s)) & 0xffu, 0.0f);
  else if constexpr (QT == Q_FP6) return dec_fp6(t.lo0[s], ((s < 2 ? t.hi0 : t.hi1) >> (16 * (s & 1))) & 0xffffu);
  else {
    const uint4v& q = s < 2 ? t.lo0 : t.lo1;
    const uint32_t w0 = q[(s & 1) * 2], w1 = q[(s & 1) * 2 + 1];
    if constexpr (QT == Q_SYM_INT8) return dec_int8(w0, w1);
    else if constexpr (QT == Q_FP8_E5M2) return dec_e5m2(w0, w1);
    else return dec_e4m3(w0, w1);
  }
}

template <int QT>
__device__ __forceinline__ constexpr float post_scale() {
  if (QT == Q_FP4) return FP4_POST;
  if (QT == Q_FP6) return FP6_POST;
  if (QT == Q_FP8_E4M3) return E4M3_POST;
  return 1.0f;
}

template <int QT>
__device__ __forceinline__ constexpr bool has_min() {
  return QT == Q_ASYM_INT4 || QT == Q_ASYM_INT5 || QT == Q_WOQ_INT4;
}

// sum of the 8 halves of a fragment, fp32
__device__ __forceinline__ float frag_sum(half8 a) {
  H8 u = {a};
  const half2v one = {(_Float16)1.0f, (_Float16)1.0f};
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_fdot2(u.h2[i], one, s, false);
  return s;
}
__device__ __forceinline__ float frag_sumsq(half8 a) {
  H8 u = {a};
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_fdot2(u.h2[i], u.h2[i], s, false);
  return s;
}

constexpr int GEMV_NW = 8;   // wavefronts per workgroup
constexpr int XPAD = 8;      // halves of padding per LDS x row (16 B: rows land on different banks)

// MT = number of 16-row x tiles (M <= 16*MT).  GATE_UP: second weight stream + act(g)*u.
// XLDS: x (after the optional RMSNorm scaling by gamma) is staged once per workgroup in LDS and the
// A fragments are read from there; otherwise they are fetched from global memory (large M*K).
template <int QT, int MT, bool GATE_UP, bool XLDS>
__global__ __launch_bounds__(GEMV_NW * 64) void gemv_t16_kernel(GemvArgs a) {
  constexpr int NACC = GATE_UP ? 2 : 1;
  constexpr int NSTREAM = GATE_UP ? 2 : 1;
  // weight tiles fetched per wavefront per round trip (per stream): 16 B x 64 lanes x 4 = 4 KiB in flight
  constexpr int GEMV_U = GATE_UP ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // carve: red | rs | lut | x
  float* red = (float*)smem;                                  // [NW][NACC*MT][64][4]
  float* rs_sh = red + GEMV_NW * NACC * MT * 256;             // [MT*16] 1/rms per row
  _Float16* nf4_lut = (_Float16*)(rs_sh + MT * 16);           // [64*16] (nf4 only)
  _Float16* xs = nf4_lut + (QT == Q_NF4 ? 64 * 16 : 0);       // [M][Kp + XPAD]

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int KT = (a.K + 127) >> 7;
  const int Kp = KT * 128;
  const int XS = Kp + XPAD;
  const int64_t nt = blockIdx.x;
  T16Planes pl[NSTREAM];
  pl[0] = t16_planes(a.w, QT, a.N, a.K);
  if constexpr (GATE_UP) pl[1] = t16_planes(a.w2, QT, a.N, a.K);

  // ---- 1. put the first round of weight loads in flight before anything else ----
  const int ntw = wave < KT ? (KT - wave + GEMV_NW - 1) / GEMV_NW : 0;   // tiles owned by this wavefront
  Tile<QT> cur[NSTREAM][GEMV_U], nxt[NSTREAM][GEMV_U];
#pragma unroll
  for (int u = 0; u < GEMV_U; ++u)
#pragma unroll
    for (int st = 0; st < NSTREAM; ++st)
      load_tile<QT>(cur[st][u], pl[st], nt * KT + wave + GEMV_NW * u, lane, u < ntw);

  // ---- 2. stage x (optionally RMS-normalised) in LDS: wavefront w handles rows w, w+NW, ... ----
  const bool norm = a.gamma != nullptr;
  const _Float16* lut = nullptr;
  if constexpr (QT == Q_NF4) {
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) nf4_lut[i] = (_Float16)NF4_VALUES[i & 15];
    lut = nf4_lut + lane * 16;
  }
  if constexpr (XLDS) {
    for (int m = wave; m < a.M; m += GEMV_NW) {
      float ssq = 0.0f;
      for (int c = lane * 8; c < Kp; c += 512) {
        half8 xv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c < a.K) xv = *(const half8*)(a.x + (int64_t)m * a.ldx + c);
        if (norm) {
          ssq += frag_sumsq(xv);
          half8 gm = c < a.K ? *(const half8*)(a.gamma + c) : xv;
          if (a.gamma_offset != 0.0f) gm = gm + (_Float16)a.gamma_offset;
          xv = xv * gm;
        }
        *(half8*)(xs + m * XS + c) = xv;
      }
      if (norm) {
        ssq = wave_sum(ssq);
        if (lane == 0) rs_sh[m] = rsqrtf(ssq / (float)a.K + a.eps);
      }
    }
  }
  __syncthreads();

  float4v acc[NACC][MT];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = float4v{0, 0, 0, 0};
  float ssq_g[MT];   // !XLDS path only
#pragma unroll
  for (int j = 0; j < MT; ++j) ssq_g[j] = 0.0f;
  const float4v zero4 = {0, 0, 0, 0};

  // ---- 3. main loop: prefetch round c+1, consume round c ----
  for (int base = 0; base < ntw; base += GEMV_U) {
    const bool more = base + GEMV_U < ntw;
    if (more) {
#pragma unroll
      for (int u = 0; u < GEMV_U; ++u)
#pragma unroll
        for (int st = 0; st < NSTREAM; ++st)
          load_tile<QT>(nxt[st][u], pl[st], nt * KT + wave + GEMV_NW * (base + GEMV_U + u), lane, base + GEMV_U + u < ntw);
    }
#pragma unroll
    for (int u = 0; u < GEMV_U; ++u) {
      if (base + u >= ntw) break;
      const int kt = wave + GEMV_NW * (base + u);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k0 = kt * 128 + 32 * s + 8 * g;
        half8 xa[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const int m = 16 * j + r;
          half8 xv = {0, 0, 0, 0, 0, 0, 0, 0};
          if constexpr (XLDS) {
            if (m < a.M) xv = *(const half8*)(xs + m * XS + k0);
          } else {
            if (m < a.M && k0 < a.K) xv = *(const half8*)(a.x + (int64_t)m * a.ldx + k0);
            if (norm) {
              ssq_g[j] += frag_sumsq(xv);
              half8 gm = k0 < a.K ? *(const half8*)(a.gamma + k0) : xv;
              if (a.gamma_offset != 0.0f) gm = gm + (_Float16)a.gamma_offset;
              xv = xv * gm;
            }
          }
          xa[j] = xv;
        }
#pragma unroll
        for (int st = 0; st < NSTREAM; ++st) {
          const Tile<QT>& t = cur[st][u];
          const half8 bf = decode_step<QT>(t, s, lut);
          const float d = (float)t.d[s] * post_scale<QT>();
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            const float4v c1 = mfma16(xa[j], bf, zero4);
            acc[st][j] += c1 * d;
            if constexpr (has_min<QT>()) {
              // + min * sum_k x[m][k] over the block: per-row sums live on lanes (row m = lane&15),
              // the result needs them on (col n = lane&15, rows 4g+q): exchange through the wave
              float sx = frag_sum(xa[j]);
              sx += __shfl_xor(sx, 16);
              sx += __shfl_xor(sx, 32);
              const float mn = (float)t.m[s];
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[st][j][q] += mn * __shfl(sx, 4 * g + q);
            }
          }
        }
      }
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < GEMV_U; ++u)
#pragma unroll
        for (int st = 0; st < NSTREAM; ++st) cur[st][u] = nxt[st][u];
    }
  }

  // ---- 4. combine the K-split across wavefronts ----
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) *(float4v*)(red + ((wave * NACC * MT + i * MT + j) * 64 + lane) * 4) = acc[i][j];
  if constexpr (!XLDS) {
    if (norm) {
      // per-wave partial sums of squares -> rs_sh via atomics on LDS (rows are few)
      if (threadIdx.x < MT * 16) rs_sh[threadIdx.x] = 0.0f;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        float s = ssq_g[j];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (g == 0) atomicAdd(&rs_sh[16 * j + r], s);
      }
    }
  }
  __syncthreads();
  if (wave != 0) return;

  const int n = (int)nt * 16 + r;
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    float4v o[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      o[i] = *(const float4v*)(red + ((i * MT + j) * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < GEMV_NW; ++w) o[i] += *(const float4v*)(red + ((w * NACC * MT + i * MT + j) * 64 + lane) * 4);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = 16 * j + 4 * g + q;
      if (m >= a.M || n >= a.N) continue;
      float rs = 1.0f;
      if (norm) rs = XLDS ? rs_sh[m] : rsqrtf(rs_sh[m] / (float)a.K + a.eps);
      float v = o[0][q] * rs;
      if (a.bias) v += (float)((const _Float16*)a.bias)[n];
      if constexpr (GATE_UP) {
        float u = o[1][q] * rs;
        if (a.bias2) u += (float)((const _Float16*)a.bias2)[n];
        v = act_apply(a.act, v) * u;
      } else {
        v = act_apply(a.act, v);
      }
      const int64_t oi = (int64_t)m * a.ldo + n;
      if (a.out_type == OUT_F16) {
        if (a.residual) v += load_out<OUT_F16>(a.residual, oi);
        store_out<OUT_F16>(a.out, oi, v);
      } else if (a.out_type == OUT_BF16) {
        if (a.residual) v += load_out<OUT_BF16>(a.residual, oi);
        store_out<OUT_BF16>(a.out, oi, v);
      } else {
        if (a.residual) v += load_out<OUT_F32>(a.residual, oi);
        store_out<OUT_F32>(a.out, oi, v);
      }
    }
  }
}

constexpr size_t GEMV_XLDS_LIMIT = 48 * 1024;   // stage x in LDS when it fits this budget

template <int QT, int MT, bool GATE_UP>
static int launch_x(const GemvArgs& a, hipStream_t st) {
  const dim3 grid((a.N + 15) / 16), block(GEMV_NW * 64);
  const int Kp = (a.K + 127) / 128 * 128;
  const size_t fixed = (size_t)GEMV_NW * (GATE_UP ? 2 : 1) * MT * 256 * 4 + MT * 16 * 4 + (QT == Q_NF4 ? 64 * 16 * 2 : 0);
  const size_t xbytes = (size_t)a.M * (Kp + XPAD) * 2;
  if (xbytes <= GEMV_XLDS_LIMIT) {
    hipLaunchKernelGGL((gemv_t16_kernel<QT, MT, GATE_UP, true>), grid, block, fixed + xbytes, st, a);
  } else {
    hipLaunchKernelGGL((gemv_t16_kernel<QT, MT, GATE_UP, false>), grid, block, fixed, st, a);
  }
  return HIP_CHECK_LAUNCH();
}


template <int QT>
static int launch_qt(const GemvArgs& a, bool gate_up, hipStream_t st) {
  const int mt = (a.M + 15) / 16;
  if (gate_up) {
    switch (mt) {
      case 1: return launch_x<QT, 1, true>(a, st);
      case 2: return launch_x<QT, 2, true>(a, st);
      default: return -4;
    }
  }
  switch (mt) {
    case 1: return launch_x<QT, 1, false>(a, st);
    case 2: return launch_x<QT, 2, false>(a, st);
    case 3: case 4: return launch_x<QT, 4, false>(a, st);
    default: return -4;
  }
}

}  // namespace bigdl
