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
// Tuning knobs of benchmark/layer_probe.hip (the defaults are the production configuration):
// GEMV_FORCE_OCC = 1 caps the single-tile 8-wavefront kernels at 80 VGPRs (6 wavefronts per SIMD);
// GEMV_NW_MAIN = wavefronts per workgroup of the kernels whose x fits the LDS budget.
// GEMV_STAMP = 1: diagnostic build, every wavefront writes 100 MHz wall-clock stamps of its phases to g_gemv_stamps
// ([workgroup][wavefront][8]; 0 entry, 1 loads issued, 2 x staged, 3 after barrier, 4 main loop done, 5 exit).
#ifndef GEMV_STAMP
#define GEMV_STAMP 0
#endif
#if GEMV_STAMP
__device__ unsigned long long* g_gemv_stamps;
#define GEMV_STAMP_AT(i)                                                                                       \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                            \
    if (lane == 0) g_gemv_stamps[((size_t)blockIdx.x * GEMV_NW + wave) * 8 + (i)] = t_;                        \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#else
#define GEMV_STAMP_AT(i)
#endif
#ifndef GEMV_FORCE_OCC
#define GEMV_FORCE_OCC 0
#endif
#ifndef GEMV_SCLDS
#define GEMV_SCLDS 0
#endif
#ifndef GEMV_NW_MAIN
#define GEMV_NW_MAIN 8
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
  // slot_cat = top_k > 0 (kernel template parameter CAT): blockIdx.y = token; the token's top_k routed products are summed
  // inside the launch by running them as ONE product over the concatenated reduction axis, k-tile kt belonging to
  // expert slot_expert[token * top_k + kt / (KT / top_k)].  x is [tokens * top_k, K / top_k] contiguous (ldx = K), already
  // scaled by the routing weights; rows of absent experts (negative id) must be zero.  K / top_k % 128 == 0.
  int slot_cat;
  // CU-resident kernels (template parameter CU): output tiles whose K-split partials fit the LDS at once
  int cu_ntc;
};

template <int QT>
struct Tile {
  // this lane's share of one 16x128 weight tile
  uint4v lo0, lo1;     // 4-bit: lo0 only; 8-bit: lo0 = steps 0,1  lo1 = steps 2,3
  uint32_t hi0, hi1;   // 1-bit plane: hi0 (4 bytes); 2-bit plane: hi0,hi1 (8 bytes)
  half4 d, m;          // block scales / mins of the lane's row for the 4 steps
};

#ifndef GEMV_NT
#define GEMV_NT 1
#endif
template <typename T>
__device__ __forceinline__ T ldg_nt(const void* p) {
  // weights are streamed exactly once per call: non-temporal loads keep them from displacing x/L2 state
#if GEMV_NT
  return __builtin_nontemporal_load((const T*)p);
#else
  return *(const T*)p;
#endif
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
#ifdef GEMV_ABL_NOSC
  t.d = half4{1, 1, 1, 1};
  return;
#endif
  t.d = ldg_nt<half4>(p.sc + tile * 64 + r * 4);
  if constexpr (QT == Q_ASYM_INT4 || QT == Q_ASYM_INT5 || QT == Q_WOQ_INT4) t.m = ldg_nt<half4>(p.mn + tile * 64 + r * 4);
}

// The GEMV fetches codes and scales separately: load_tile_codes per tile, then ONE scale load per round of up to four
// tiles (RoundScales): lane group g = lane / 16 fetches the four block scales of row lane % 16 of the round's tile
// g, and the consumer pulls the scales of tile j from lane group j with two ds_bpermute.  One 512-byte scale request
// per round instead of a 4x redundant 64-byte one per tile: half as many memory instructions per wavefront and 6
// fewer live registers per round, which is what lets three 8-wavefront workgroups share a CU.
template <int QT>
__device__ __forceinline__ void load_tile_codes(Tile<QT>& t, const T16Planes& p, int64_t tile, int lane) {
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
}

struct RoundScales {
  uint2v d, m;     // raw fp16 x 4: scales / mins of (tile = lane group, row = lane % 16)
};

// the four scales of row `lane % 16` of the round's tile j, from the lanes that loaded them
__device__ __forceinline__ half4 round_scale(uint2v raw, int j, int lane) {
  const int src = ((lane & 15) + 16 * j) << 2;
  union { uint2v u; half4 h; } c;
  c.u.x = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)raw.x);
  c.u.y = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)raw.y);
  return c.h;
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
// ATT: x is the split-KV attention workspace (a.att_ws), combined while it is staged (needs XLDS, no norm)
// Single-tile kernels with 8 wavefronts are launched three workgroups to a CU (N / 16 = 688 .. 768 workgroups on 256
// CUs): 6 wavefronts per SIMD, i.e. at most 80 VGPRs, keeps all of them resident at once.
// CU: "one workgroup per CU" form for up to 16 rows: the grid is min(#CUs, N / 16) workgroups of 16 wavefronts, workgroup b
// owns the 16-row weight tiles b, b + grid, b + 2 grid, ... and streams them back to back through the same two register
// sets, x is staged once per CU instead of once per tile, and the K-split partials of all its tiles are combined after
// ONE barrier, one wavefront per tile.  With 1024 threads the register budget is 128 VGPRs whatever the tile count,
// so residency no longer depends on the allocator staying under an occupancy step.
// LEAN: every (chunk, row) item of x is one of the EX early items of its wavefront (M * ceil(ceil(K / 512) / NW) <= EX: one
// row up to K = 8192, two up to K = 4096); the general staging loop is compiled out, which keeps the tile-per-workgroup kernel
// within the 80 VGPRs that let three of its workgroups share a CU.
template <int QT, int MT, bool GATE_UP, bool XLDS, int NW, bool ATT = false, bool CU = false, bool LEAN = false, bool CAT = false>
#ifndef GEMV_GU_OCC
#define GEMV_GU_OCC 5
#endif
__global__ __launch_bounds__(NW * 64, LEAN ? 6 : (NW + 3) / 4) void gemv_t16_kernel(GemvArgs a) {
  static_assert(!ATT || XLDS, "the attention combine lives in the LDS staging of x");
  static_assert(!CU || (MT == 1 && XLDS && NW == 16), "CU form: at most 16 rows, x in LDS, 16 wavefronts");
  static_assert(!CAT || (MT == 1 && XLDS && !GATE_UP && !ATT && !CU && !LEAN && !GEMV_SCLDS), "summed routed products: one row, x in LDS");
  constexpr int GEMV_NW = NW;
  constexpr int NACC = GATE_UP ? 2 : 1;
  constexpr int NSTREAM = GATE_UP ? 2 : 1;
  // weight tiles per wavefront per round (per stream).  Two rounds (register sets ta / tb) are requested before anything
  // else happens, so a wavefront has 4 KiB (8 KiB with two streams) of codes in flight from its first instructions on.
#ifdef GEMV_U
  constexpr int U = GEMV_U;
#else
  constexpr int U = 2;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // carve: red | ssq | sum | lut | xsum | x
  float* red = (float*)smem;                                  // [NW][NACC*MT][64][4]; CU: [ntc][NW][NACC][rows_g * 16][4]
  const int rows_g = (a.M + 3) >> 2;                          // CU: lane groups (4 rows each) that hold valid rows
  const int red_floats = CU ? a.cu_ntc * GEMV_NW * NACC * rows_g * 64 : GEMV_NW * NACC * MT * 256;
  // per-wavefront partial sums of x^2 and x per row (RMSNorm / LayerNorm prologue): every wavefront only ever touches
  // its own slice, so the staging needs neither atomics nor a barrier between zeroing and accumulation
  float* ssq_sh = red + red_floats;                           // [NW][MT*16]
  float* sum_sh = ssq_sh + GEMV_NW * MT * 16;                 // [NW][MT*16]
  _Float16* nf4_lut = (_Float16*)(sum_sh + GEMV_NW * MT * 16);   // [64*16] (nf4 only)
  // formats with a block minimum: sum of (scaled) x over every 32-block, [Kp/32 + XTAIL/32][MT*16 rows], so that the
  // "+ min * sum(x)" term is one 16-byte LDS read per MFMA step instead of a cross-lane reduction of the fragment
  float* xsum = (float*)(nf4_lut + (QT == Q_NF4 ? 64 * 16 : 0));
  constexpr bool XSUM = XLDS && has_min<QT>();
  const int xsum_floats = XSUM ? (((a.K + 127) >> 7) * 4 + XTAIL / 32) * MT * 16 : 0;
  // tile-per-workgroup form: the block scales (and minima) of the workgroup's whole tile row are staged in LDS by a
  // few full-width loads (they are contiguous per tile row: KT x 128 bytes) instead of one 64-byte request per k-tile
#ifndef GEMV_SCLDS
#define GEMV_SCLDS 0
#endif
  constexpr bool SCLDS = GEMV_SCLDS && !CU;
  constexpr int NPL = has_min<QT>() ? 2 : 1;                  // planes: scales, minima
  const int KT_ = (a.K + 127) >> 7;
  _Float16* sc_lds = (_Float16*)(xsum + xsum_floats);         // [NPL][NSTREAM][KT][16][4]
  _Float16* xs = sc_lds + (SCLDS ? NPL * NSTREAM * KT_ * 64 : 0);   // [M][Kp + XTAIL + XPAD]

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
  int cat_e[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // CAT: first weight tile row of each of the token's experts
  int cat_kt = 1;                              // CAT: k-tiles per expert
  if constexpr (CAT) {
    const int tok = blockIdx.y;
    orow = tok;
    n_planes = a.n_total;
    a.x += (int64_t)tok * a.ldx;
    cat_kt = (a.K >> 7) / a.slot_cat;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < a.slot_cat) cat_e[j] = max(__builtin_amdgcn_readfirstlane(a.slot_expert[tok * a.slot_cat + j]), 0) * (a.N >> 4);
  } else if (a.slot_expert != nullptr) {
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
  // (CAT: a.K spans the token's top_k experts; the stacked matrices themselves are K / top_k wide)
  const int k_planes = CAT ? a.K / a.slot_cat : a.K;
  pl[0] = t16_planes(a.w, QT, n_planes, k_planes);
  if constexpr (GATE_UP) pl[1] = t16_planes(a.w2, QT, n_planes, k_planes);
  const int64_t tile0 = (wt + nt) * KT;
  // CU form: this workgroup's tiles are nt + i * gridDim.x, i < cnt; a tile takes RPN rounds of U k-tiles per wavefront
  const int n_tiles = (a.N + 15) >> 4;
  const int cnt = CU ? (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 1;
  const int RPN = (((KT + GEMV_NW - 1) / GEMV_NW) + U - 1) / U;

  if constexpr (GEMV_PROBE == 1) return;
  GEMV_STAMP_AT(0);
  const bool norm = a.gamma != nullptr;
  const bool lnorm = a.ln_c1 != nullptr;
  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  // x is cut into column chunks of 64 vectors (one 16-byte load per lane); wavefront w stages chunks w, w + NW, ...
  // for all rows.  Only [0, K) is loaded, the rest of an LDS row (padding up to Kp plus the zero tail) is filled
  // without touching memory.
  const int kvec = a.K >> 3;
  const int chunks = (kvec + 63) >> 6;
  const int kmax = a.K - 8;
  const int vec_per_row = (Kp + XTAIL) >> 3;

  // ---- 1. x first: the first EX (chunk, row) items of this wavefront are requested BEFORE the weights.  Vector
  // memory returns in order, so an x load issued behind the weight loads would only be usable once every weight
  // tile of the round has landed: the whole workgroup would sit at the staging barrier for the length of the
  // weight stream and dequantization + MFMA would start after it instead of under it.
  constexpr int EX = 2;
  half8 ex[EX], eg[EX];
  bool ev[EX];
  int ecol[EX], erow[EX];
#ifdef GEMV_ABL_NOX
  constexpr bool ABL_NOX = true;
#else
  constexpr bool ABL_NOX = false;
#endif
  if constexpr (XLDS && !ATT && !ABL_NOX) {
    int ci = 0, m = 0;
#pragma unroll
    for (int e = 0; e < EX; ++e) {
      const int c = wave + GEMV_NW * ci;
      ev[e] = c < chunks;                       // wavefront-uniform
      ecol[e] = (c * 64 + lane) << 3;
      erow[e] = m;
      ex[e] = zero8;
      eg[e] = zero8;
      if (ev[e]) {
        const int cl = min(ecol[e], kmax);
        ex[e] = *(const half8*)(a.x + (int64_t)m * a.ldx + cl);
        if (norm) eg[e] = *(const half8*)(a.gamma + cl);
      }
      if (++m == a.M) { m = 0; ++ci; }
    }
  }

  // scales / minima of the tile row: vector i (16 bytes) of plane pq = i / (KT * 8), one load per thread and pass
  constexpr int ESC = 2;                                      // passes requested before the weights
  uint4v esc[ESC];
  const int sc_vecs = SCLDS ? NPL * NSTREAM * KT * 8 : 0;
  auto sc_src = [&](int i) {
    const int pq = i / (KT * 8), off = i - pq * (KT * 8);
    const _Float16* base = pl[0].sc;
    if constexpr (NSTREAM == 2) {
      if (pq & 1) base = pl[1].sc;
    }
    if constexpr (has_min<QT>()) {
      if (pq >= NSTREAM) {
        base = pl[0].mn;
        if constexpr (NSTREAM == 2) {
          if (pq & 1) base = pl[1].mn;
        }
      }
    }
    return (const uint4v*)(base + tile0 * 64) + off;
  };
  if constexpr (SCLDS) {
#pragma unroll
    for (int e = 0; e < ESC; ++e) {
      const int i = threadIdx.x + e * GEMV_NW * 64;
      esc[e] = uint4v{0, 0, 0, 0};
      if (i < sc_vecs) esc[e] = ldg_nt<uint4v>(sc_src(i));
    }
  }

  // ---- 2. first round of weight loads ----
  // this wavefront owns k-tiles wave, wave+NW, ...; indices past the end are clamped to the last tile
  // (a valid address) and neutralised by reading the zero tail of x.
  const int ntw = wave < KT ? (KT - wave + GEMV_NW - 1) / GEMV_NW : 0;
  // register sets: two (ping-pong) in the tile-per-workgroup form; three in the CU form, where they hold ALL the
  // rounds of the usual decode shapes (three tiles of K = 4096, one of K = 11008), i.e. every weight byte of the
  // workgroup is requested before the first one is used and no request ever waits for dequantization to free a set
  // block scales: RSC = one scale load per round + ds_bpermute (see RoundScales); otherwise one 64-byte request per tile
  // with every lane group fetching the same four scales.  The round form wins where registers decide how many
  // workgroups share a CU (the lean single-stream kernel: 70 instead of 90 VGPRs) and in the CU form; the gate/up and
  // 16-wavefront kernels measured faster with per-tile scale loads.
  constexpr bool RSC = !SCLDS && (CU || (LEAN && !GATE_UP));
  constexpr int NSET = CU ? 3 : 2;
  Tile<QT> ta[NSTREAM][U], tb[NSTREAM][U], tc[CU ? NSTREAM : 1][CU ? U : 1];
  RoundScales sa, sb, sc3;
  constexpr int NT_R = NSTREAM * U;             // tiles per round, slot j = u * NSTREAM + st
  static_assert(NT_R == 2 || NT_R == 4, "one scale load per round covers at most four tiles");
  const int jl = g & (NT_R - 1), ul = jl / NSTREAM, sl = jl % NSTREAM;     // the slot whose scales this lane fetches
  int f_i = 0, f_rd = 0;      // CU: fetch cursor (tile of this workgroup, round within the tile)
  // weight tile of k-tile kt (already clamped to the last one) of the tile row that starts at t0
  auto tile_at = [&](int64_t t0, int kt) -> int64_t {
    if constexpr (CAT) {
      int j = 0, first = cat_e[0];
#pragma unroll
      for (int q = 1; q < 8; ++q)
        if (kt >= q * cat_kt) { j = q; first = cat_e[q]; }
      return ((int64_t)first + nt) * cat_kt + (kt - j * cat_kt);
    } else {
      return t0 + kt;
    }
  };
  auto fetch = [&](Tile<QT> (&tt)[NSTREAM][U], RoundScales& rs, int base) {
    int64_t t0 = tile0;
    if constexpr (CU) {
      // rounds are fetched strictly in order; past the end the last round is fetched again (never consumed)
      t0 = (int64_t)((int)blockIdx.x + min(f_i, cnt - 1) * (int)gridDim.x) * KT;
      base = f_rd * U;
      if (++f_rd == RPN) { f_rd = 0; ++f_i; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int st = 0; st < NSTREAM; ++st)
      {
        if constexpr (RSC || SCLDS) load_tile_codes<QT>(tt[st][u], pl[st], tile_at(t0, min(wave + GEMV_NW * (base + u), KT - 1)), lane);
        else load_tile<QT>(tt[st][u], pl[st], tile_at(t0, min(wave + GEMV_NW * (base + u), KT - 1)), lane);
      }
    if constexpr (!RSC) return;
#ifndef GEMV_ABL_NOSC
    const int64_t tl = tile_at(t0, min(wave + GEMV_NW * (base + ul), KT - 1));
    const _Float16* scp = pl[0].sc;
    const _Float16* mnp = pl[0].mn;
    if constexpr (NSTREAM == 2) {
      if (sl) { scp = pl[1].sc; mnp = pl[1].mn; }
    }
    rs.d = ldg_nt<uint2v>(scp + tl * 64 + r * 4);
    if constexpr (has_min<QT>()) rs.m = ldg_nt<uint2v>(mnp + tl * 64 + r * 4);
#else
    rs.d = uint2v{0x3c003c00u, 0x3c003c00u};
    rs.m = uint2v{0, 0};
#endif
  };
  fetch(ta, sa, 0);
#ifndef GEMV_GU_PF2
#define GEMV_GU_PF2 0
#endif
  // request the second set before x is staged as well: pays in the lean kernel (all of a K = 4096 tile row in flight
  // from the start); the 16-wavefront and gate/up kernels measured slower with it
  constexpr bool PF2 = LEAN || CU || (GATE_UP && GEMV_GU_PF2);
  if (PF2 && (CU ? cnt * RPN > 1 : U < ntw)) fetch(tb, sb, U);
  if constexpr (CU) {
    if (cnt * RPN > 2) fetch(tc, sc3, 0);
  }

  GEMV_STAMP_AT(1);
  // ---- 3. stage x (optionally scaled by the norm's gamma) in LDS, all wavefronts together ----
  const _Float16* lut = nullptr;
  if constexpr (QT == Q_NF4) {
    fill_nf4_banked((uint32_t*)nf4_lut, threadIdx.x, GEMV_NW * 64);
    lut = (const _Float16*)((const uint32_t*)nf4_lut + (lane & 31));
  }
  if constexpr (SCLDS) {
#pragma unroll
    for (int e = 0; e < ESC; ++e) {
      const int i = threadIdx.x + e * GEMV_NW * 64;
      if (i < sc_vecs) *(uint4v*)(sc_lds + 8 * i) = esc[e];
    }
    for (int i = threadIdx.x + ESC * GEMV_NW * 64; i < sc_vecs; i += GEMV_NW * 64) *(uint4v*)(sc_lds + 8 * i) = ldg_nt<uint4v>(sc_src(i));
  }
  if (lane < MT * 16) {
    ssq_sh[wave * (MT * 16) + lane] = 0.0f;
    sum_sh[wave * (MT * 16) + lane] = 0.0f;
  }
  if constexpr (XSUM) {
    const float4v z4 = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < (xsum_floats >> 2); i += GEMV_NW * 64) *(float4v*)(xsum + 4 * i) = z4;
    __syncthreads();   // xsum zeroed before the block sums below
  }
  if constexpr (XLDS && !ABL_NOX) {
    for (int i = threadIdx.x; i < a.M * (vec_per_row - kvec); i += GEMV_NW * 64) {
      const int m = i / (vec_per_row - kvec), v = kvec + i % (vec_per_row - kvec);
      *(half8*)(xs + m * XS + (v << 3)) = zero8;
    }
    // one staged vector: norm statistics into this wavefront's partial slots, gamma, block sums, LDS store
    auto finish = [&](half8 v, half8 g8, int col, int m) {
      const bool inside = col < a.K;
      if (!inside) v = zero8;
      if (norm) {
        const float ssq = wave_sum_dpp(frag_sumsq(v));       // (on the path to the staging barrier)
        if (lane == 0) ssq_sh[wave * (MT * 16) + m] += ssq;
        if (lnorm) {
          const float sm = wave_sum_dpp(frag_sum(v));
          if (lane == 0) sum_sh[wave * (MT * 16) + m] += sm;
        }
        if (a.gamma_offset != 0.0f) g8 = g8 + (_Float16)a.gamma_offset;
        v = v * g8;
      }
      if constexpr (XSUM) {
        float bs = frag_sum(v);                       // 4 consecutive lanes = one 32-block of this row
        bs += __shfl_xor(bs, 1);
        bs += __shfl_xor(bs, 2);
        if (inside && (lane & 3) == 0) xsum[(col >> 5) * (MT * 16) + m] = bs;
      }
      if (inside) *(half8*)(xs + m * XS + col) = v;
    };
    if constexpr (!ATT) {
#pragma unroll
      for (int e = 0; e < EX; ++e)
        if (ev[e]) finish(ex[e], eg[e], ecol[e], erow[e]);
      // items beyond the early ones (many rows, or K above EX * NW * 512): SU rows of a chunk in flight together,
      // with UNCONDITIONAL loads from clamped addresses (a load under a per-lane condition makes the compiler wait for
      // each one separately)
      auto stage = [&](auto su_tag, int col, half8 g8, int m0) {
        constexpr int SU = decltype(su_tag)::value;     // rows in flight together, all valid
        const int cl = min(col, kmax);
        half8 xv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) xv[u] = *(const half8*)(a.x + (int64_t)(m0 + u) * a.ldx + cl);
#pragma unroll
        for (int u = 0; u < SU; ++u) finish(xv[u], g8, col, m0 + u);
      };
      const int ci0 = EX / a.M, mstart = EX % a.M;       // first item after the early ones
      for (int ci = ci0, c = wave + GEMV_NW * ci0; !LEAN && c < chunks; ++ci, c += GEMV_NW) {
        const int col = (c * 64 + lane) << 3;
        half8 g8 = zero8;
        if (norm) g8 = *(const half8*)(a.gamma + min(col, kmax));
        int m0 = ci == ci0 ? mstart : 0;
        for (; m0 + 3 < a.M; m0 += 4) stage(std::integral_constant<int, 4>{}, col, g8, m0);
        for (; m0 + 1 < a.M; m0 += 2) stage(std::integral_constant<int, 2>{}, col, g8, m0);
        for (; m0 < a.M; ++m0) stage(std::integral_constant<int, 1>{}, col, g8, m0);
      }
    } else {
      // x[m][hq * D + d] = softmax-weighted combination of the attention kernel's per-split partial rows.  One thread
      // per 8 channels; the (m, l) pairs and O pieces of 8 splits are requested together (one memory round trip per 8
      // splits instead of two dependent ones per split) and folded in with a running maximum.
      const int D = a.att_D;
      const int64_t stride = (int64_t)a.att_rows_alloc * (D + 4);
      for (int i = threadIdx.x; i < a.M * kvec; i += GEMV_NW * 64) {
        const int m = a.M == 1 ? 0 : i / kvec, c = (a.M == 1 ? i : i % kvec) << 3;
        const int hq = c / D, d = c - hq * D;
        const int bb = a.att_qlen == 1 ? m : m / a.att_qlen, tok = a.att_qlen == 1 ? 0 : m % a.att_qlen;
        const int hk = a.att_G == 1 ? hq : hq / a.att_G, gq = a.att_G == 1 ? 0 : hq % a.att_G;
        const int64_t bh = (int64_t)bb * a.att_Hkv + hk;
        const int row = tok * a.att_G + gq;
        const float* p0 = a.att_ws + (bh * a.att_splits * a.att_rows_alloc + row) * (D + 4);
        float mm = -1.0e30f, lt = 0.0f;
        float4v o0 = {0, 0, 0, 0}, o1 = {0, 0, 0, 0};
        for (int s0 = 0; s0 < a.att_splits; s0 += 8) {
          float2v ml[8];
          float4v v0[8], v1[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float* ps = p0 + min(s0 + j, a.att_splits - 1) * stride;
            ml[j] = *(const float2v*)(ps + D);
            v0[j] = *(const float4v*)(ps + d);
            v1[j] = *(const float4v*)(ps + d + 4);
          }
          float gm = mm;
#pragma unroll
          for (int j = 0; j < 8; ++j) gm = fmaxf(gm, s0 + j < a.att_splits ? ml[j][0] : -1.0e30f);
          const float fo = exp2f(mm - gm);
          lt *= fo;
          o0 *= fo;
          o1 *= fo;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float f = s0 + j < a.att_splits ? exp2f(ml[j][0] - gm) : 0.0f;
            lt += f * ml[j][1];
            o0 += v0[j] * f;
            o1 += v1[j] * f;
          }
          mm = gm;
        }
        const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;
        half8 xv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          xv[q] = (_Float16)(o0[q] * inv);
          xv[4 + q] = (_Float16)(o1[q] * inv);
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
  GEMV_STAMP_AT(2);
  if constexpr (!ABL_NOX) __syncthreads();
  GEMV_STAMP_AT(3);

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

  auto consume = [&](Tile<QT> (&tt)[NSTREAM][U], const RoundScales& rs, int base) {
    if constexpr (GEMV_PROBE >= 2) {
      uint32_t x = 0;
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int st = 0; st < NSTREAM; ++st) {
          const Tile<QT>& t = tt[st][u];
          x ^= t.lo0[0] ^ t.lo0[1] ^ t.lo0[2] ^ t.lo0[3];
        }
      if constexpr (RSC) x ^= rs.d.x ^ rs.d.y;
      acc[0][0][0][0] += __uint_as_float(x & 0x3f800000u);
      return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kt = wave + GEMV_NW * (base + u);
      const int ktx = min(kt, KT);   // KT = zero tail
      half4 dd[NSTREAM], mn4[NSTREAM];
#pragma unroll
      for (int st = 0; st < NSTREAM; ++st) {
        if constexpr (SCLDS) {
          const int kts = min(kt, KT - 1);
          dd[st] = *(const half4*)(sc_lds + ((st * KT + kts) * 16 + r) * 4);
          if constexpr (has_min<QT>()) mn4[st] = *(const half4*)(sc_lds + (((NSTREAM + st) * KT + kts) * 16 + r) * 4);
        } else if constexpr (RSC) {
          dd[st] = round_scale(rs.d, u * NSTREAM + st, lane);
          if constexpr (has_min<QT>()) mn4[st] = round_scale(rs.m, u * NSTREAM + st, lane);
        } else {
          dd[st] = tt[st][u].d;
          if constexpr (has_min<QT>()) mn4[st] = tt[st][u].m;
        }
      }
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
          const float d = (float)dd[st][s] * post_scale<QT>();
          const float2v d2 = {d, d};
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            const float4v c1 = mfma16(xa[j], bf, zero4);
            acc[st][j][0] = __builtin_elementwise_fma(float2v{c1[0], c1[1]}, d2, acc[st][j][0]);
            acc[st][j][1] = __builtin_elementwise_fma(float2v{c1[2], c1[3]}, d2, acc[st][j][1]);
            if constexpr (XSUM) {
              // + min * sum_k x[m][k] over the block, block sums precomputed at staging: rows 4g .. 4g+3
              const float4v x4 = *(const float4v*)(xsum + (ktx * 4 + s) * (MT * 16) + 16 * j + 4 * g);
              const float mn = (float)mn4[st][s];
              acc[st][j][0] = __builtin_elementwise_fma(float2v{x4[0], x4[1]}, float2v{mn, mn}, acc[st][j][0]);
              acc[st][j][1] = __builtin_elementwise_fma(float2v{x4[2], x4[3]}, float2v{mn, mn}, acc[st][j][1]);
            } else if constexpr (has_min<QT>()) {
              // + min * sum_k x[m][k] over the block: per-row sums live on lanes (row m = lane&15),
              // the result needs them on (col n = lane&15, rows 4g+q): exchange through the wave
              float sx = frag_sum(xa[j]);
              sx += __shfl_xor(sx, 16);
              sx += __shfl_xor(sx, 32);
              const float mn = (float)mn4[st][s];
              acc[st][j][0][0] += mn * __shfl(sx, 4 * g + 0);
              acc[st][j][0][1] += mn * __shfl(sx, 4 * g + 1);
              acc[st][j][1][0] += mn * __shfl(sx, 4 * g + 2);
              acc[st][j][1][1] += mn * __shfl(sx, 4 * g + 3);
            }
          }
        }
#ifdef GEMV_SCHED_STEP
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
#ifdef GEMV_SCHED_TILE
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };
  // row statistics of the fused norm, from the per-wavefront partials (valid after the barrier that ends the main loop)
  auto row_stats = [&](int m, float& rs, float& mean) {
    rs = 1.0f;
    mean = 0.0f;
    if (norm) {
      float ssq = 0.0f, sm = 0.0f;
#pragma unroll
      for (int w = 0; w < GEMV_NW; ++w) {
        ssq += ssq_sh[w * (MT * 16) + m];
        sm += sum_sh[w * (MT * 16) + m];
      }
      if (lnorm) {
        mean = sm / (float)a.K;
        rs = rsqrtf(fmaxf(ssq / (float)a.K - mean * mean, 0.0f) + a.eps);
      } else {
        rs = rsqrtf(ssq / (float)a.K + a.eps);
      }
    }
  };
  // epilogue of one output element: norm factors, bias, activation / gate, residual, store
  auto emit = [&](float v, float u, int m, int n) {
    float rs, mean;
    row_stats(m, rs, mean);
    v *= rs;
    if (lnorm) v += a.ln_c2[n] - rs * mean * a.ln_c1[n];
    if (a.bias) v += (float)((const _Float16*)a.bias)[n];
    if constexpr (GATE_UP) {
      u *= rs;
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
  };

  if constexpr (CU) {
    // ---- 3'. all tiles of this workgroup, round after round through the two register sets ----
    const int Q = GEMV_PROBE == 2 ? 1 : cnt * RPN;
    const int ntc = a.cu_ntc;
    int c_i = 0, c_rd = 0;          // consume cursor
    // after the last round of a tile: park the partial sums in the tile's LDS slot; after the last tile of a chunk of
    // ntc tiles (or of the workgroup): barrier, one wavefront per tile adds the 16 partials and writes the outputs
    auto round_done = [&]() {
      if (++c_rd < RPN) return;
      c_rd = 0;
      const int slot = c_i % ntc;
      if (g < rows_g) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          const float4v v = {acc[i][0][0][0], acc[i][0][0][1], acc[i][0][1][0], acc[i][0][1][1]};
          *(float4v*)(red + ((((slot * GEMV_NW + wave) * NACC + i) * rows_g + g) * 16 + r) * 4) = v;
        }
      }
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i][0][0] = acc[i][0][1] = float2v{0, 0};
      ++c_i;
      if (slot != ntc - 1 && c_i != cnt) return;
      __syncthreads();
      const int first = c_i - 1 - slot;           // first tile of the chunk
      if (wave <= slot && g < rows_g) {
        const int n = ((int)blockIdx.x + (first + wave) * (int)gridDim.x) * 16 + r;
        float4v o[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          o[i] = float4v{0, 0, 0, 0};
#pragma unroll
          for (int w = 0; w < GEMV_NW; ++w) o[i] += *(const float4v*)(red + ((((wave * GEMV_NW + w) * NACC + i) * rows_g + g) * 16 + r) * 4);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = 4 * g + q;
          if (m < a.M && n < a.N) emit(o[0][q], o[NACC - 1][q], m, n);
        }
      }
      if (c_i != cnt) __syncthreads();            // the slots are reused by the next chunk
    };
    if constexpr (CU) {
      for (int q = 0; q < Q; q += 3) {
        consume(ta, sa, c_rd * U);
        round_done();
        if (q + 3 < Q) fetch(ta, sa, 0);
        if (q + 1 < Q) {
          consume(tb, sb, c_rd * U);
          round_done();
          if (q + 4 < Q) fetch(tb, sb, 0);
        }
        if (q + 2 < Q) {
          consume(tc, sc3, c_rd * U);
          round_done();
          if (q + 5 < Q) fetch(tc, sc3, 0);
        }
      }
    }
    GEMV_STAMP_AT(5);
    return;
  }

  // ---- 3. main loop: ping-pong between two register sets; a set is re-requested as soon as it has been consumed ----
  for (int base = 0; base < (GEMV_PROBE == 2 ? min(ntw, 1) : ntw); base += 2 * U) {
    if (GEMV_PROBE != 2 && base + U < ntw && !(PF2 && base == 0)) fetch(tb, sb, base + U);
    consume(ta, sa, base);
    if (GEMV_PROBE != 2 && base + U < ntw) {
      if (base + 2 * U < ntw) fetch(ta, sa, base + 2 * U);
      consume(tb, sb, base + U);
    }
  }

  GEMV_STAMP_AT(4);
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
        if (g == 0) ssq_sh[wave * (MT * 16) + 16 * j + r] = s;
      }
    }
  }
  __syncthreads();
  if (wave != 0) {
    GEMV_STAMP_AT(5);
    return;
  }

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
      if (m < a.M && n < a.N) emit(o[0][q], o[NACC - 1][q], m, n);
    }
  }
  GEMV_STAMP_AT(5);
}

constexpr size_t GEMV_XLDS_LIMIT = 48 * 1024;   // stage x in LDS when it fits this budget
// ... or, for at most 8 rows, this one (K = 11008 down projections at 3-4 rows: 21.5 -> about 12 us per launch; K = 4096
// at 6-8 rows; decode step 2.22 -> 1.92 ms at batch 4 and 2.98 -> 2.69 ms at batch 8).  With more rows the single
// workgroup per CU that such a footprint allows loses more than the LDS staging wins (measured at 16 rows).  148 KiB
// (all but the partial-sum slots of the 160 KiB LDS) rather than 100: the K = 11008 products keep their LDS staging at
// 5 and 6 rows (decode step 3.18 -> 2.97 ms at batch 5, 3.24 -> 3.06 at batch 6, same box).
constexpr size_t GEMV_XLDS_LIMIT_FEW_ROWS = 148 * 1024;

// more than 64 KiB of dynamic LDS has to be requested once per kernel
static inline bool gemv_allow_lds(const void* kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return true;
  if (bytes > 160 * 1024) return false;
  static const void* done[256];
  static int n_done = 0;
  for (int i = 0; i < n_done; ++i)
    if (done[i] == kernel) return true;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
  if (n_done < 256) done[n_done++] = kernel;
  return true;
}

static inline int gemv_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

// GEMV_CU_ENABLE: 0 = tile-per-workgroup kernels only (the attention combine then has no kernel), 1 = CU-resident form
// wherever it applies, 2 = CU-resident form only where one workgroup per CU covers the grid anyway (N <= 16 * #CUs)
#ifndef GEMV_CU_ENABLE
#define GEMV_CU_ENABLE 0
#endif
// ... and at run time: the CU-resident form for row counts in [gemv_cu_rows_lo, gemv_cu_rows_hi] (0, 0 = never).  With
// several rows every tile-per-workgroup kernel stages all rows of x (2 bytes of x per byte of 4-bit weights at 8 rows);
// the CU-resident form stages them once per CU.  Llama-2-7B decode step at 1024-token context, one MI355X, tile form ->
// CU form: 2 rows 1.94 -> 2.23 ms, 4: 2.56 -> 2.75, 6: 3.33 -> 3.25, 7: 3.45 -> 3.52, 8: 3.25 -> 3.01, 9: 3.84 -> 3.45,
// 10: 3.88 -> 3.59, 12: 4.03 -> 3.75, 14: 4.15 -> 4.10 (16 rows of K = 4096 no longer fit next to the partial sums).
// Hence 8 .. 13 rows by default (bigdl_gemv_cu_rows / BIGDL_GEMV_CU_ROWS=lo,hi change it).
inline int& gemv_cu_rows_lo() { static int v = 8; return v; }
inline int& gemv_cu_rows_hi() { static int v = 13; return v; }

template <int QT, int MT, bool GATE_UP>
static int launch_x(const GemvArgs& a_in, hipStream_t st) {
  GemvArgs a = a_in;
  a.cu_ntc = 1;
  const dim3 grid((a.N + 15) / 16, a.slot_expert ? a.n_slots : 1);
  const int Kp = (a.K + 127) / 128 * 128;
  const size_t xbytes = (size_t)a.M * (Kp + XTAIL + XPAD) * 2;
  constexpr int NACC = GATE_UP ? 2 : 1;
  auto misc = [&](int nw, bool xlds = true) {
    return (size_t)2 * nw * MT * 16 * 4 + (QT == Q_NF4 ? 64 * 16 * 2 : 0) +
           (has_min_of(QT) && xlds ? (size_t)(Kp / 32 + XTAIL / 32) * MT * 16 * 4 : 0);
  };
  // tile-per-workgroup kernels: partial-sum slots, staged scales (and minima) of the tile row
  auto fixed = [&](int nw, bool xlds = true) {
    return (size_t)nw * NACC * MT * 256 * 4 + misc(nw, xlds) +
           (GEMV_SCLDS ? (size_t)(has_min_of(QT) ? 2 : 1) * NACC * (Kp / 128) * 128 : 0);
  };
  if (a.slot_expert != nullptr && a.slot_cat > 0) {
    // summed routed products (n_slots = tokens): the general 8-wavefront kernel with x in LDS
    if constexpr (MT == 1 && !GATE_UP && !GEMV_SCLDS) {
#ifndef GEMV_CAT_NW
#define GEMV_CAT_NW 16
#endif
      auto k = gemv_t16_kernel<QT, 1, false, true, GEMV_CAT_NW, false, false, false, true>;
      const size_t lds = fixed(GEMV_CAT_NW) + xbytes;
      if (a.M != 1 || a.slot_cat > 8 || (a.K & 127) || (a.K >> 7) % a.slot_cat || !gemv_allow_lds((const void*)k, lds)) return -7;
      hipLaunchKernelGGL(k, grid, dim3(GEMV_CAT_NW * 64), lds, st, a);
      return HIP_CHECK_LAUNCH();
    } else {
      return -7;
    }
  }
  if constexpr (MT == 1) {
    // CU-resident form: one workgroup per CU streams all its output tiles (see the kernel's comment).  Needs x and
    // at least one tile's partial sums in LDS.
    // (products with fewer tiles than CUs stay in the tile form: nothing to share, and half-empty workgroups of 16
    // wavefronts: TinyLlama's 2048-wide products lose 10 - 20 % of a decode step in the CU form)
    // ... and so do short reductions: with K = 2048 each of the 16 wavefronts gets one k-tile per output tile and nothing to
    // overlap (same model: gate / up alone in the CU form still cost 14 %)
    const bool cu_rows = a.M >= gemv_cu_rows_lo() && a.M <= gemv_cu_rows_hi() && (a.N + 15) / 16 >= gemv_cu_count() &&
                         (a.K >> 7) >= 24;
    if ((GEMV_CU_ENABLE || cu_rows) && a.slot_expert == nullptr &&
        (GEMV_CU_ENABLE == 1 || cu_rows || (a.N + 15) / 16 <= gemv_cu_count())) {
      const int n_tiles = (a.N + 15) / 16;
      const int G = n_tiles < gemv_cu_count() ? n_tiles : gemv_cu_count();
      const int cnt_max = (n_tiles + G - 1) / G;
      const size_t slot = (size_t)16 * NACC * ((a.M + 3) / 4) * 64 * 4;
      const size_t budget = 150 * 1024;
      if (xbytes + misc(16) + slot <= budget) {
        size_t ntc = (budget - xbytes - misc(16)) / slot;
        if (ntc > 16) ntc = 16;
        if (ntc > (size_t)cnt_max) ntc = cnt_max;
        a.cu_ntc = (int)ntc;
        const size_t lds = ntc * slot + misc(16) + xbytes;
        if (a.att_ws != nullptr) {
          if constexpr (!GATE_UP) {
            auto k = gemv_t16_kernel<QT, 1, false, true, 16, true, true>;
            if (!gemv_allow_lds((const void*)k, lds)) return -5;
            hipLaunchKernelGGL(k, dim3(G), dim3(1024), lds, st, a);
            return HIP_CHECK_LAUNCH();
          } else {
            return -7;
          }
        }
        auto k = gemv_t16_kernel<QT, 1, GATE_UP, true, 16, false, true>;
        if (!gemv_allow_lds((const void*)k, lds)) return -5;
        hipLaunchKernelGGL(k, dim3(G), dim3(1024), lds, st, a);
        return HIP_CHECK_LAUNCH();
      }
    }
  }
  // 16 wavefronts when the grid is small (about one workgroup per CU) and every wavefront still gets
  // at least two k-tiles: halves the per-wavefront instruction stream of the latency-bound small GEMVs
  const bool wide = MT == 1 && !GATE_UP && xbytes <= GEMV_XLDS_LIMIT && grid.x <= 256 && grid.y == 1 && Kp / 128 >= 32;
  if (a.att_ws != nullptr) {
    // split-KV attention partials combined in the staging prologue (host entry: x fits GEMV_XLDS_LIMIT, at most 16 rows)
    if constexpr (MT == 1 && !GATE_UP) {
      if (wide) hipLaunchKernelGGL((gemv_t16_kernel<QT, 1, false, true, 16, true>), grid, dim3(1024), fixed(16) + xbytes, st, a);
      else hipLaunchKernelGGL((gemv_t16_kernel<QT, 1, false, true, 8, true>), grid, dim3(512), fixed(8) + xbytes, st, a);
      return HIP_CHECK_LAUNCH();
    } else {
      return -7;
    }
  }
  if constexpr (MT == 1) {
    if (wide) {
      if constexpr (!GATE_UP) {
        hipLaunchKernelGGL((gemv_t16_kernel<QT, 1, false, true, 16>), grid, dim3(1024), fixed(16) + xbytes, st, a);
        return HIP_CHECK_LAUNCH();
      }
    }
    // one or two rows of the usual hidden sizes: the lean kernel, three (gate/up: two and a half) workgroups per CU
#ifndef GEMV_GU_LEAN
#define GEMV_GU_LEAN 0
#endif
    if ((!GATE_UP || GEMV_GU_LEAN) && a.M * (((a.K + 511) / 512 + 7) / 8) <= 2 && a.ln_c1 == nullptr && xbytes <= GEMV_XLDS_LIMIT) {
      hipLaunchKernelGGL((gemv_t16_kernel<QT, 1, GATE_UP, true, 8, false, false, true>), grid, dim3(512), fixed(8) + xbytes, st, a);
      return HIP_CHECK_LAUNCH();
    }
  }
  if (xbytes <= GEMV_XLDS_LIMIT ||
      (a.M <= 8 && xbytes <= GEMV_XLDS_LIMIT_FEW_ROWS &&
       gemv_allow_lds((const void*)gemv_t16_kernel<QT, MT, GATE_UP, true, GEMV_NW_MAIN>, fixed(GEMV_NW_MAIN) + xbytes))) {
    hipLaunchKernelGGL((gemv_t16_kernel<QT, MT, GATE_UP, true, GEMV_NW_MAIN>), grid, dim3(GEMV_NW_MAIN * 64),
                       fixed(GEMV_NW_MAIN) + xbytes, st, a);
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
