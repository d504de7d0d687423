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
};

template <int QT>
struct Tile {
  // loads one 16x128 tile share of this lane and decodes MFMA step s
  uint4v lo0, lo1;     // 4-bit: lo0 only; 8-bit: lo0 = steps 0,1  lo1 = steps 2,3
  uint32_t hi0, hi1;   // 1-bit plane: hi0 (4 bytes); 2-bit plane: hi0,hi1 (8 bytes)
  float d[4], m[4];
};

template <int QT>
__device__ __forceinline__ void load_tile(Tile<QT>& t, const T16Planes& p, int64_t tile, int lane) {
  constexpr int LO = (QT == Q_SYM_INT8 || QT == Q_FP8_E4M3 || QT == Q_FP8_E5M2) ? 8 : 4;
  if constexpr (LO == 4) {
    t.lo0 = *(const uint4v*)(p.lo + tile * 1024 + lane * 16);
  } else {
    t.lo0 = *(const uint4v*)(p.lo + tile * 2048 + lane * 16);
    t.lo1 = *(const uint4v*)(p.lo + tile * 2048 + 1024 + lane * 16);
  }
  if constexpr (QT == Q_SYM_INT5 || QT == Q_ASYM_INT5) {
    t.hi0 = *(const uint32_t*)(p.hi + tile * 256 + lane * 4);
  } else if constexpr (QT == Q_FP6) {
    const uint2v h = *(const uint2v*)(p.hi + tile * 512 + lane * 8);
    t.hi0 = h.x;
    t.hi1 = h.y;
  }
  const int r = lane & 15;
  const half4 s4 = *(const half4*)(p.sc + tile * 64 + r * 4);
#pragma unroll
  for (int s = 0; s < 4; ++s) t.d[s] = (float)s4[s];
  if constexpr (QT == Q_ASYM_INT4 || QT == Q_ASYM_INT5 || QT == Q_WOQ_INT4) {
    const half4 m4 = *(const half4*)(p.mn + tile * 64 + r * 4);
#pragma unroll
    for (int s = 0; s < 4; ++s) t.m[s] = (float)m4[s];
  }
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

constexpr int GEMV_NW = 8;  // wavefronts per workgroup

// MT = number of 16-row x tiles (M <= 16*MT).  GATE_UP: second weight stream + act(g)*u.
template <int QT, int MT, bool GATE_UP, int OUT>
__global__ __launch_bounds__(GEMV_NW * 64) void gemv_t16_kernel(GemvArgs a) {
  constexpr int NACC = GATE_UP ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float red[GEMV_NW][NACC * MT][64][4];
  __shared__ float ssq_part[GEMV_NW][MT][16];
  __shared__ _Float16 nf4_lut[64 * 16];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int KT = (a.K + 127) >> 7;
  const int64_t nt = blockIdx.x;
  const T16Planes p1 = t16_planes(a.w, QT, a.N, a.K);
  T16Planes p2 = p1;
  if constexpr (GATE_UP) p2 = t16_planes(a.w2, QT, a.N, a.K);

  const _Float16* lut = nullptr;
  if constexpr (QT == Q_NF4) {
    // one private 16-entry copy per lane (stride 16 halves = 8 banks) keeps lookups conflict-light
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) nf4_lut[i] = (_Float16)NF4_VALUES[i & 15];
    __syncthreads();
    lut = nf4_lut + lane * 16;
  }

  float4v acc[NACC][MT];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = float4v{0, 0, 0, 0};
  float ssq[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) ssq[j] = 0.0f;

  const bool norm = a.gamma != nullptr;
  const float4v zero4 = {0, 0, 0, 0};

  for (int kt = wave; kt < KT; kt += GEMV_NW) {
    Tile<QT> t1, t2;
    load_tile<QT>(t1, p1, nt * KT + kt, lane);
    if constexpr (GATE_UP) load_tile<QT>(t2, p2, nt * KT + kt, lane);
    // x fragments for the 4 steps of this tile
    half8 xa[MT][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k0 = kt * 128 + 32 * s + 8 * g;
      half8 gm;
      if (norm) {
        gm = k0 < a.K ? *(const half8*)(a.gamma + k0) : half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (a.gamma_offset != 0.0f) gm = gm + (_Float16)a.gamma_offset;
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int m = 16 * j + r;
        half8 xv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (m < a.M && k0 < a.K) xv = *(const half8*)(a.x + (int64_t)m * a.ldx + k0);
        if (norm) {
          ssq[j] += frag_sumsq(xv);
          xv = xv * gm;
        }
        xa[j][s] = xv;
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const half8 b1 = decode_step<QT>(t1, s, lut);
      half8 b2;
      if constexpr (GATE_UP) b2 = decode_step<QT>(t2, s, lut);
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const float4v c1 = mfma16(xa[j][s], b1, zero4);
        acc[0][j] += c1 * (t1.d[s] * post_scale<QT>());
        if constexpr (GATE_UP) {
          const float4v c2 = mfma16(xa[j][s], b2, zero4);
          acc[1][j] += c2 * (t2.d[s] * post_scale<QT>());
        }
        if constexpr (has_min<QT>()) {
          // + min * sum_k x[m][k] over the block: per-row sums live on lanes (row m = lane&15),
          // the result needs them on (col n = lane&15, rows 4g+reg): exchange through bpermute
          float sx = frag_sum(xa[j][s]);
          sx += __shfl_xor(sx, 16);
          sx += __shfl_xor(sx, 32);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float sxm = __shfl(sx, 4 * g + q);  // lane (4g+q) holds the sum of row m = 4g+q
            acc[0][j][q] += t1.m[s] * sxm;
            if constexpr (GATE_UP) acc[1][j][q] += t2.m[s] * sxm;
          }
        }
      }
    }
  }

  // ---- combine the K-split across wavefronts ----
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) *(float4v*)red[wave][i * MT + j][lane] = acc[i][j];
  if (norm) {
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      float s = ssq[j];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (g == 0) ssq_part[wave][j][r] = s;
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
      o[i] = *(const float4v*)red[0][i * MT + j][lane];
#pragma unroll
      for (int w = 1; w < GEMV_NW; ++w) o[i] += *(const float4v*)red[w][i * MT + j][lane];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = 16 * j + 4 * g + q;
      if (m >= a.M || n >= a.N) continue;
      float rs = 1.0f;
      if (norm) {
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < GEMV_NW; ++w) tot += ssq_part[w][j][4 * g + q];
        rs = rsqrtf(tot / (float)a.K + a.eps);
      }
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
      if (a.residual) v += load_out<OUT>(a.residual, oi);
      store_out<OUT>(a.out, oi, v);
    }
  }
}

template <int QT, int MT, bool GATE_UP>
static int launch_out(const GemvArgs& a, int out_type, hipStream_t st) {
  const dim3 grid((a.N + 15) / 16), block(GEMV_NW * 64);
  switch (out_type) {
    case OUT_F16: hipLaunchKernelGGL((gemv_t16_kernel<QT, MT, GATE_UP, OUT_F16>), grid, block, 0, st, a); break;
    case OUT_BF16: hipLaunchKernelGGL((gemv_t16_kernel<QT, MT, GATE_UP, OUT_BF16>), grid, block, 0, st, a); break;
    case OUT_F32: hipLaunchKernelGGL((gemv_t16_kernel<QT, MT, GATE_UP, OUT_F32>), grid, block, 0, st, a); break;
    default: return -3;
  }
  return HIP_CHECK_LAUNCH();
}

template <int QT>
static int launch_qt(const GemvArgs& a, bool gate_up, int out_type, hipStream_t st) {
  const int mt = (a.M + 15) / 16;
  if (gate_up) {
    switch (mt) {
      case 1: return launch_out<QT, 1, true>(a, out_type, st);
      case 2: return launch_out<QT, 2, true>(a, out_type, st);
      default: return -4;
    }
  }
  switch (mt) {
    case 1: return launch_out<QT, 1, false>(a, out_type, st);
    case 2: return launch_out<QT, 2, false>(a, out_type, st);
    case 3: case 4: return launch_out<QT, 4, false>(a, out_type, st);
    default: return -4;
  }
}

}  // namespace bigdl

using namespace bigdl;

// Returns 0 on success; negative = rejected on the host before any launch.
// M <= 64 (<= 32 with w2).  x fp16 with ldx % 8 == 0 and 16-byte aligned rows; K % 32 == 0.
extern "C" int bigdl_gemv_lowbit(const void* x, const void* w, const void* w2, void* out, const void* bias,
                                 const void* bias2, const void* residual, const void* gamma, float eps,
                                 float gamma_offset, int M, int N, int K, int ldx, int ldo, int qtype, int act,
                                 int out_type, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K & 31) || (ldx & 7) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return -1;
  if (gamma && ((uintptr_t)gamma & 15)) return -1;
  if (w2 && ((uintptr_t)w2 & 15)) return -1;
  if (M > (w2 ? 32 : 64)) return -4;
  GemvArgs a;
  a.x = (const _Float16*)x; a.w = w; a.w2 = w2; a.out = out; a.bias = bias; a.bias2 = bias2;
  a.residual = residual; a.gamma = (const _Float16*)gamma; a.eps = eps; a.gamma_offset = gamma_offset;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.act = act;
  const bool gu = w2 != nullptr;
  switch (qtype) {
    case Q_SYM_INT4: return launch_qt<Q_SYM_INT4>(a, gu, out_type, stream);
    case Q_ASYM_INT4: return launch_qt<Q_ASYM_INT4>(a, gu, out_type, stream);
    case Q_WOQ_INT4: return launch_qt<Q_WOQ_INT4>(a, gu, out_type, stream);
    case Q_FP4: return launch_qt<Q_FP4>(a, gu, out_type, stream);
    case Q_NF4: return launch_qt<Q_NF4>(a, gu, out_type, stream);
    case Q_SYM_INT5: return launch_qt<Q_SYM_INT5>(a, gu, out_type, stream);
    case Q_ASYM_INT5: return launch_qt<Q_ASYM_INT5>(a, gu, out_type, stream);
    case Q_FP6: return launch_qt<Q_FP6>(a, gu, out_type, stream);
    case Q_SYM_INT8: return launch_qt<Q_SYM_INT8>(a, gu, out_type, stream);
    case Q_FP8_E4M3: return launch_qt<Q_FP8_E4M3>(a, gu, out_type, stream);
    case Q_FP8_E5M2: return launch_qt<Q_FP8_E5M2>(a, gu, out_type, stream);
    default: return -2;
  }
}
