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
#include <type_traits>

// Compile-time probe used only by benchmark/gemv_probe.hip to attribute kernel time (0 = production kernel):
// 1 = return after the workgroup set-up, 2 = first round of loads + x staging + reduce/epilogue (no main loop),
// 3 = all loads, dequantization and MFMA replaced by an xor of the loaded words.
#ifndef GEMV_PROBE
#define GEMV_PROBE 0
#endif

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
  // optional LayerNorm prologue (needs gamma): with c1 = W . gamma and c2 = W . beta precomputed per output column,
  //   W . ((x - mean) * rs * gamma + beta) = rs * (W . (x * gamma)) - rs * mean * c1 + c2,
  // so the kernel stages x * gamma exactly as for RMSNorm and also accumulates the row sums.  ln2_* belong to w2.
  const float* ln_c1;
  const float* ln_c2;
  const float* ln2_c1;
  const float* ln2_c2;
  int M, N, K;
  int ldx, ldo;
  int act;
  int out_type;
  // optional: x is the not-yet-combined output of the split-KV attention kernel.  Workspace layout
  // [B*Hkv][splits][rows_alloc][D + 4] fp32 (O row, running max m, normaliser l); x[m][hq*D + d] is
  // combined on the fly while x is staged into LDS (saves the attention kernel's cross-split hand-off).
  const float* att_ws;
  int att_splits, att_D, att_G, att_Hkv, att_qlen, att_rows_alloc;
  // optional mixture-of-experts mode: blockIdx.y = routed slot (token-major, top_k slots per token).  Slot s is
  // a one-row product with the weights of expert slot_expert[s] (the experts' [N, K] matrices are stacked by rows
  // into one [n_total, K] T16 buffer, N % 16 == 0, so expert e starts at tile row e * N/16), reads x row s / slot_xdiv, writes out row s scaled by slot_scale[s].  A negative expert id
  // (expert owned by another rank) writes a zero row.  The expert ids are read on the device, so the step stays
  // graph-replayable.
  const int* slot_expert;
  const float* slot_scale;
  int n_total;
  int slot_xdiv, n_slots;
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

// unconditional loads: the caller passes an in-range tile index
template <int QT>
__device__ __forceinline__ void load_tile(Tile<QT>& t, const T16Planes& p, int64_t tile, int lane) {
  constexpr int LO = (QT == Q_SYM_INT8 || QT == Q_FP8_E4M3 || QT == Q_FP8_E5M2) ? 8 : 4;
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

// 8 int4 codes -> 8 fp16 values (code - bias): 1 shift, 4 and/or, 4 packed fp16 ops.
// Codes in nibbles 0 and 4 of a dword become 1024+q in place; codes in nibbles 1 and 5 become
// 1024+16q in place and are brought back with one fused multiply-add (x/16 - 64 - bias).
__device__ __forceinline__ half8 dec_int4_fast(uint32_t w, float bias) {
  H8 r;
  const half2v offA = {(_Float16)(1024.0f + bias), (_Float16)(1024.0f + bias)};
  const half2v mulB = {(_Float16)0.0625f, (_Float16)0.0625f};
  const half2v offB = {(_Float16)(-64.0f - bias), (_Float16)(-64.0f - bias)};
  const uint32_t w8 = w >> 8;
  // v_and_or_b32 takes at most one scalar/literal operand on gfx950: keep the magic exponent in a VGPR
  uint32_t magic;
  asm("v_mov_b32 %0, 0x64006400" : "=v"(magic));
  uint32_t a0, a1, a2, a3;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a0) : "v"(w), "s"(0x000F000Fu), "v"(magic));
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a1) : "v"(w), "s"(0x00F000F0u), "v"(magic));
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a2) : "v"(w8), "s"(0x000F000Fu), "v"(magic));
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a3) : "v"(w8), "s"(0x00F000F0u), "v"(magic));
  r.h2[0] = u2h(a0) - offA;
  r.h2[1] = u2h(a1) * mulB + offB;
  r.h2[2] = u2h(a2) - offA;
  r.h2[3] = u2h(a3) * mulB + offB;
  return r.v;
}

template <int QT>
__device__ __forceinline__ half8 decode_step(const Tile<QT>& t, int s, const _Float16* lut) {
  if constexpr (QT == Q_SYM_INT4) return dec_int4_fast(t.lo0[s], 8.0f);
  else if constexpr (QT == Q_ASYM_INT4 || QT == Q_WOQ_INT4) return dec_int4_fast(t.lo0[s], 0.0f);
  else if constexpr (QT == Q_FP4) return dec_fp4(t.lo0[s]);
  else if constexpr (QT == Q_NF4) return dec_nf4_banked(t.lo0[s], (const uint32_t*)lut);
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

typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int XPAD = 8;      // halves of padding per LDS x row (16 B: rows land on different banks)
constexpr int XTAIL = 128;   // zero halves after each LDS x row: the k-tile "one past the end" reads zeros

// MT = number of 16-row x tiles (M <= 16*MT).  GATE_UP: second weight stream + act(g)*u.
// XLDS: x (after the optional RMSNorm scaling by gamma) is staged once per workgroup in LDS and the
// A fragments are read from there; otherwise they are fetched from global memory (large M*K).
// NW = wavefronts per workgroup (the K split)
template <int QT, int MT, bool GATE_UP, bool XLDS, int NW>
__global__ __launch_bounds__(NW * 64) void gemv_t16_kernel(GemvArgs a) {
  constexpr int GEMV_NW = NW;
  constexpr int NACC = GATE_UP ? 2 : 1;
  constexpr int NSTREAM = GATE_UP ? 2 : 1;
  // weight tiles fetched per wavefront per round trip (per stream): 16 B x 64 lanes x 4 = 4 KiB in flight
  constexpr int U = (GATE_UP || NW > 8) ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // carve: red | ssq | sum | lut | xsum | x
  float* red = (float*)smem;                                  // [NW][NACC*MT][64][4]
  float* ssq_sh = red + GEMV_NW * NACC * MT * 256;            // [MT*16] sum of squares per row
  float* sum_sh = ssq_sh + MT * 16;                           // [MT*16] sum per row (LayerNorm prologue)
  _Float16* nf4_lut = (_Float16*)(sum_sh + MT * 16);          // [64*16] (nf4 only)
  // formats with a block minimum: sum of (scaled) x over every 32-block, [Kp/32 + XTAIL/32][MT*16 rows], so that the
  // "+ min * sum(x)" term is one 16-byte LDS read per MFMA step instead of a cross-lane reduction of the fragment
  float* xsum = (float*)(nf4_lut + (QT == Q_NF4 ? 64 * 16 : 0));
  constexpr bool XSUM = XLDS && has_min<QT>();
  const int xsum_floats = XSUM ? (((a.K + 127) >> 7) * 4 + XTAIL / 32) * MT * 16 : 0;
  _Float16* xs = (_Float16*)(xsum + xsum_floats);             // [M][Kp + XTAIL + XPAD]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int KT = (a.K + 127) >> 7;
  const int Kp = KT * 128;
  const int XS = Kp + XTAIL + XPAD;
  const int64_t nt = blockIdx.x;
  int orow = 0, n_planes = a.N;
  int64_t wt = 0;      // first weight tile row of the selected expert
  float oscale = 1.0f;
  if (a.slot_expert != nullptr) {
    const int slot = blockIdx.y;
    const int e = a.slot_expert[slot];
    orow = slot;
    if (e < 0) {
      const int n0 = (int)nt * 16 + lane;
      if (threadIdx.x < 16 && n0 < a.N) {
        const int64_t oi = (int64_t)slot * a.ldo + n0;
        if (a.out_type == OUT_F16) store_out<OUT_F16>(a.out, oi, 0.0f);
        else if (a.out_type == OUT_BF16) store_out<OUT_BF16>(a.out, oi, 0.0f);
        else store_out<OUT_F32>(a.out, oi, 0.0f);
      }
      return;
    }
    wt = (int64_t)e * (a.N >> 4);
    n_planes = a.n_total;
    a.x += (int64_t)(slot / a.slot_xdiv) * a.ldx;
    if (a.slot_scale != nullptr) oscale = a.slot_scale[slot];
  }
  T16Planes pl[NSTREAM];
  pl[0] = t16_planes(a.w, QT, n_planes, a.K);
  if constexpr (GATE_UP) pl[1] = t16_planes(a.w2, QT, n_planes, a.K);
  const int64_t tile0 = (wt + nt) * KT;

  if constexpr (GEMV_PROBE == 1) return;
  // ---- 1. put the first round of weight loads in flight before anything else ----
  // this wavefront owns k-tiles wave, wave+NW, ...; indices past the end are clamped to the last tile
  // (a valid address) and neutralised by reading the zero tail of x.
  const int ntw = wave < KT ? (KT - wave + GEMV_NW - 1) / GEMV_NW : 0;
  Tile<QT> ta[NSTREAM][U], tb[NSTREAM][U];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int st = 0; st < NSTREAM; ++st) load_tile<QT>(ta[st][u], pl[st], tile0 + min(wave + GEMV_NW * u, KT - 1), lane);

  // ---- 2. stage x (optionally scaled by the RMSNorm gamma) in LDS, all wavefronts together ----
  const bool norm = a.gamma != nullptr;
  const bool lnorm = a.ln_c1 != nullptr;
  const _Float16* lut = nullptr;
  if constexpr (QT == Q_NF4) {
    fill_nf4_banked((uint32_t*)nf4_lut, threadIdx.x, GEMV_NW * 64);
    lut = (const _Float16*)((const uint32_t*)nf4_lut + (lane & 31));
  }
  if (threadIdx.x < 2 * MT * 16) ssq_sh[threadIdx.x] = 0.0f;   // ssq_sh and sum_sh are adjacent
  if constexpr (XSUM) {
    const float4v z4 = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < (xsum_floats >> 2); i += GEMV_NW * 64) *(float4v*)(xsum + 4 * i) = z4;
  }
  if constexpr (XLDS) {
    __syncthreads();   // ssq_sh / xsum zeroed before the atomics and block sums below
    const int vec_per_row = (Kp + XTAIL) >> 3;
    if (a.att_ws == nullptr) {
      // Each wavefront stages whole 64-vector segments (one row, 64 x 16 bytes), up to 4 per round, with
      // UNCONDITIONAL loads from clamped addresses (a load under a per-item condition makes the compiler wait for
      // each one separately) and no dead slots (a redundant clamped load still costs L2 bandwidth): the loads of
      // several rows are in flight together instead of one dependent L2 round trip per row (probe: +1.3 us per
      // extra row of x with the row loop).
      // rows are cut into segments of 64 vectors (one wavefront load); only [0, K) is loaded, the rest of the LDS
      // row (padding up to Kp plus the zero tail) is filled without touching memory
      const int kvec = a.K >> 3;
      const int chunks = (kvec + 63) >> 6;
      const int kmax = a.K - 8;
      const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = threadIdx.x; i < a.M * (vec_per_row - kvec); i += GEMV_NW * 64) {
        const int m = i / (vec_per_row - kvec), v = kvec + i % (vec_per_row - kvec);
        *(half8*)(xs + m * XS + (v << 3)) = zero8;
      }
      // a wavefront owns a column chunk (64 vectors) for all rows: gamma is fetched once per chunk, and the rows of
      // the chunk are loaded SU at a time
      auto stage = [&](auto su_tag, int col, half8 g8, int m0) {
        constexpr int SU = decltype(su_tag)::value;     // rows in flight together, all valid
        const int cl = min(col, kmax);
        const bool inside = col < a.K;
        half8 xv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) xv[u] = *(const half8*)(a.x + (int64_t)(m0 + u) * a.ldx + cl);
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          half8 v = inside ? xv[u] : zero8;
          if (norm) {
            const float ssq = wave_sum(frag_sumsq(v));
            if (lane == 0) atomicAdd(&ssq_sh[m0 + u], ssq);
            if (lnorm) {
              const float sm = wave_sum(frag_sum(v));
              if (lane == 0) atomicAdd(&sum_sh[m0 + u], sm);
            }
            v = v * g8;
          }
          if constexpr (XSUM) {
            float bs = frag_sum(v);                       // 4 consecutive lanes = one 32-block of this row
            bs += __shfl_xor(bs, 1);
            bs += __shfl_xor(bs, 2);
            if (inside && (lane & 3) == 0) xsum[(col >> 5) * (MT * 16) + m0 + u] = bs;
          }
          if (inside) *(half8*)(xs + (m0 + u) * XS + col) = v;
        }
      };
      for (int c = wave; c < chunks; c += GEMV_NW) {
        const int col = (c * 64 + lane) << 3;
        half8 g8 = zero8;
        if (norm) {
          g8 = *(const half8*)(a.gamma + min(col, kmax));
          if (a.gamma_offset != 0.0f) g8 = g8 + (_Float16)a.gamma_offset;
        }
        int m0 = 0;
        for (; m0 + 3 < a.M; m0 += 4) stage(std::integral_constant<int, 4>{}, col, g8, m0);
        for (; m0 + 1 < a.M; m0 += 2) stage(std::integral_constant<int, 2>{}, col, g8, m0);
        for (; m0 < a.M; ++m0) stage(std::integral_constant<int, 1>{}, col, g8, m0);
      }
    } else {
    for (int m = 0; m < a.M; ++m) {
      for (int v = threadIdx.x; v < vec_per_row; v += GEMV_NW * 64) {
        const int c = v << 3;
        half8 xv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c < a.K) {
          const int hq = c / a.att_D, d = c % a.att_D;
          const int bb = m / a.att_qlen, tok = m % a.att_qlen;
          const int64_t bh = (int64_t)bb * a.att_Hkv + hq / a.att_G;
          const int row = tok * a.att_G + hq % a.att_G;
          const int64_t stride = (int64_t)a.att_rows_alloc * (a.att_D + 4);
          const float* p0 = a.att_ws + (bh * a.att_splits * a.att_rows_alloc + row) * (a.att_D + 4);
          float mm = -1.0e30f;
          for (int sp = 0; sp < a.att_splits; ++sp) mm = fmaxf(mm, p0[sp * stride + a.att_D]);
          float lt = 0.0f, o8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (int sp = 0; sp < a.att_splits; ++sp) {
            const float* ps = p0 + sp * stride;
            const float f = exp2f(ps[a.att_D] - mm);
            lt += f * ps[a.att_D + 1];
            const float4v v0 = *(const float4v*)(ps + d), v1 = *(const float4v*)(ps + d + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) { o8[q] += f * v0[q]; o8[4 + q] += f * v1[q]; }
          }
          const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;
#pragma unroll
          for (int q = 0; q < 8; ++q) xv[q] = (_Float16)(o8[q] * inv);
        }
        if constexpr (XSUM) {
          float bs = frag_sum(xv);
          bs += __shfl_xor(bs, 1);
          bs += __shfl_xor(bs, 2);
          if ((threadIdx.x & 3) == 0) xsum[(c >> 5) * (MT * 16) + m] = bs;
        }
        *(half8*)(xs + m * XS + c) = xv;
      }
    }
    }
  }
  __syncthreads();

  float2v acc[NACC][MT][2];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j][0] = acc[i][j][1] = float2v{0, 0};
  float ssq_g[MT];   // !XLDS path only
#pragma unroll
  for (int j = 0; j < MT; ++j) ssq_g[j] = 0.0f;
  const float4v zero4 = {0, 0, 0, 0};

  // LDS row of x read by this lane for each m-tile (rows >= M alias the last valid row: their results
  // are never stored, and MFMA rows do not mix)
  int xrow[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) xrow[j] = min(16 * j + r, a.M - 1) * XS + 8 * g;

  auto consume = [&](Tile<QT> (&tt)[NSTREAM][U], int base) {
    if constexpr (GEMV_PROBE >= 2) {
      uint32_t x = 0;
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int st = 0; st < NSTREAM; ++st) {
          const Tile<QT>& t = tt[st][u];
          x ^= t.lo0[0] ^ t.lo0[1] ^ t.lo0[2] ^ t.lo0[3] ^ (uint32_t)(float)t.d[0] ^ (uint32_t)(float)t.d[3];
        }
      acc[0][0][0][0] += __uint_as_float(x & 0x3f800000u);
      return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kt = wave + GEMV_NW * (base + u);
      const int ktx = min(kt, KT);   // KT = zero tail
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        half8 xa[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          if constexpr (XLDS) {
            xa[j] = *(const half8*)(xs + xrow[j] + ktx * 128 + 32 * s);
          } else {
            const int m = 16 * j + r;
            const int k0 = kt * 128 + 32 * s + 8 * g;
            half8 xv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (m < a.M && k0 < a.K && kt < KT) xv = *(const half8*)(a.x + (int64_t)m * a.ldx + k0);
            if (norm) {
              ssq_g[j] += frag_sumsq(xv);
              half8 gm = (k0 < a.K && kt < KT) ? *(const half8*)(a.gamma + k0) : xv;
              if (a.gamma_offset != 0.0f) gm = gm + (_Float16)a.gamma_offset;
              xv = xv * gm;
            }
            xa[j] = xv;
          }
        }
#pragma unroll
        for (int st = 0; st < NSTREAM; ++st) {
          const Tile<QT>& t = tt[st][u];
          const half8 bf = decode_step<QT>(t, s, lut);
          const float d = (float)t.d[s] * post_scale<QT>();
          const float2v d2 = {d, d};
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            const float4v c1 = mfma16(xa[j], bf, zero4);
            acc[st][j][0] = __builtin_elementwise_fma(float2v{c1[0], c1[1]}, d2, acc[st][j][0]);
            acc[st][j][1] = __builtin_elementwise_fma(float2v{c1[2], c1[3]}, d2, acc[st][j][1]);
            if constexpr (XSUM) {
              // + min * sum_k x[m][k] over the block, block sums precomputed at staging: rows 4g .. 4g+3
              const float4v x4 = *(const float4v*)(xsum + (ktx * 4 + s) * (MT * 16) + 16 * j + 4 * g);
              const float mn = (float)t.m[s];
              acc[st][j][0] = __builtin_elementwise_fma(float2v{x4[0], x4[1]}, float2v{mn, mn}, acc[st][j][0]);
              acc[st][j][1] = __builtin_elementwise_fma(float2v{x4[2], x4[3]}, float2v{mn, mn}, acc[st][j][1]);
            } else if constexpr (has_min<QT>()) {
              // + min * sum_k x[m][k] over the block: per-row sums live on lanes (row m = lane&15),
              // the result needs them on (col n = lane&15, rows 4g+q): exchange through the wave
              float sx = frag_sum(xa[j]);
              sx += __shfl_xor(sx, 16);
              sx += __shfl_xor(sx, 32);
              const float mn = (float)t.m[s];
              acc[st][j][0][0] += mn * __shfl(sx, 4 * g + 0);
              acc[st][j][0][1] += mn * __shfl(sx, 4 * g + 1);
              acc[st][j][1][0] += mn * __shfl(sx, 4 * g + 2);
              acc[st][j][1][1] += mn * __shfl(sx, 4 * g + 3);
            }
          }
        }
      }
    }
  };
  auto fetch = [&](Tile<QT> (&tt)[NSTREAM][U], int base) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int st = 0; st < NSTREAM; ++st)
        load_tile<QT>(tt[st][u], pl[st], tile0 + min(wave + GEMV_NW * (base + u), KT - 1), lane);
  };

  // ---- 3. main loop: ping-pong between two register sets; a round is fetched while the other is consumed ----
  for (int base = 0; base < (GEMV_PROBE == 2 ? min(ntw, 1) : ntw); base += 2 * U) {
    if (GEMV_PROBE != 2 && base + U < ntw) fetch(tb, base + U);
    consume(ta, base);
    if (GEMV_PROBE != 2 && base + U < ntw) {
      if (base + 2 * U < ntw) fetch(ta, base + 2 * U);
      consume(tb, base + U);
    }
  }

  // ---- 4. combine the K-split across wavefronts ----
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const float4v v = {acc[i][j][0][0], acc[i][j][0][1], acc[i][j][1][0], acc[i][j][1][1]};
      *(float4v*)(red + ((wave * NACC * MT + i * MT + j) * 64 + lane) * 4) = v;
    }
  if constexpr (!XLDS) {
    if (norm) {
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        float s = ssq_g[j];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (g == 0) atomicAdd(&ssq_sh[16 * j + r], s);
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
      float rs = 1.0f, mean = 0.0f;
      if (lnorm) {
        mean = sum_sh[m] / (float)a.K;
        rs = rsqrtf(fmaxf(ssq_sh[m] / (float)a.K - mean * mean, 0.0f) + a.eps);
      } else if (norm) {
        rs = rsqrtf(ssq_sh[m] / (float)a.K + a.eps);
      }
      float v = o[0][q] * rs;
      if (lnorm) v += a.ln_c2[n] - rs * mean * a.ln_c1[n];
      if (a.bias) v += (float)((const _Float16*)a.bias)[n];
      if constexpr (GATE_UP) {
        float u = o[1][q] * rs;
        if (lnorm) u += a.ln2_c2[n] - rs * mean * a.ln2_c1[n];
        if (a.bias2) u += (float)((const _Float16*)a.bias2)[n];
        v = act_apply(a.act, v) * u;
      } else {
        v = act_apply(a.act, v);
      }
      v *= oscale;
      const int64_t oi = (int64_t)(m + orow) * a.ldo + n;
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
// ... or, for at most 8 rows, this one (K = 11008 down projections at 3-4 rows: 21.5 -> about 12 us per launch; K = 4096
// at 6-8 rows; decode step 2.22 -> 1.92 ms at batch 4 and 2.98 -> 2.69 ms at batch 8).  With more rows the single
// workgroup per CU that such a footprint allows loses more than the LDS staging wins (measured at 16 rows).
constexpr size_t GEMV_XLDS_LIMIT_FEW_ROWS = 100 * 1024;

template <typename KernelT>
static inline bool gemv_allow_lds(KernelT kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return true;          // more dynamic LDS has to be requested once per kernel
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) return false;
    done = true;
  }
  return bytes <= 128 * 1024;
}

template <int QT, int MT, bool GATE_UP>
static int launch_x(const GemvArgs& a, hipStream_t st) {
  const dim3 grid((a.N + 15) / 16, a.slot_expert ? a.n_slots : 1);
  const int Kp = (a.K + 127) / 128 * 128;
  const size_t xbytes = (size_t)a.M * (Kp + XTAIL + XPAD) * 2;
  auto fixed = [&](int nw, bool xlds = true) {
    return (size_t)nw * (GATE_UP ? 2 : 1) * MT * 256 * 4 + 2 * MT * 16 * 4 + (QT == Q_NF4 ? 64 * 16 * 2 : 0) +
           (has_min_of(QT) && xlds ? (size_t)(Kp / 32 + XTAIL / 32) * MT * 16 * 4 : 0);
  };
  // 16 wavefronts when the grid is small (about one workgroup per CU) and every wavefront still gets
  // at least two k-tiles: halves the per-wavefront instruction stream of the latency-bound small GEMVs
  const bool wide = MT == 1 && !GATE_UP && xbytes <= GEMV_XLDS_LIMIT && grid.x <= 256 && Kp / 128 >= 64;
  if (wide) {
    if constexpr (MT == 1 && !GATE_UP)
      hipLaunchKernelGGL((gemv_t16_kernel<QT, MT, GATE_UP, true, 16>), grid, dim3(1024), fixed(16) + xbytes, st, a);
  } else if (xbytes <= GEMV_XLDS_LIMIT ||
             (a.M <= 8 && xbytes <= GEMV_XLDS_LIMIT_FEW_ROWS && gemv_allow_lds(gemv_t16_kernel<QT, MT, GATE_UP, true, 8>, fixed(8) + xbytes))) {
    hipLaunchKernelGGL((gemv_t16_kernel<QT, MT, GATE_UP, true, 8>), grid, dim3(512), fixed(8) + xbytes, st, a);
  } else {
    if (a.ln_c1 != nullptr) return -7;     // the LayerNorm prologue lives in the LDS staging of x
    hipLaunchKernelGGL((gemv_t16_kernel<QT, MT, GATE_UP, false, 8>), grid, dim3(512), fixed(8, false), st, a);
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
