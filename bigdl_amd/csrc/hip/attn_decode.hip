// Decode-phase attention for gfx950 (q_len * group <= 64 rows per kv head), split over the KV
// length, reading fp16 or fp8(e5m2) KV caches directly; plus the RoPE + KV-append kernel that
// feeds it.
//
// Replaces the reference's closed xe_addons.sdp / sdp_fp8 (decode SDPA, q_len <= 32),
// rotary_half_inplaced / rotary_two_inplaced and quantize_key_value
// (python/llm/src/ipex_llm/transformers/xpu_ops.py:64-135; call sites models/common.py:248-253,
// models/llama.py:153-167, models/utils.py:117-139).
//
// Cache layout (owned by this library, sized once for the whole 288 GB device):
//   K  : [B, Hkv, Lmax, D]   row per key        (fp16 or e5m2 bytes)
//   V^T: [B, Hkv, D, Lmax]   row per channel    (fp16 or e5m2 bytes)
// V is kept transposed so that both MFMA products read their cache operand with plain
// per-lane 16-byte (8-byte for fp8) loads and the softmax probabilities never leave registers:
//   S^T[key][row] = K . Q^T      (A = K rows, B = Q^T)
//   O^T[d][row]  += V^T . P^T    (A = V^T rows, B = P^T taken straight from the S^T accumulators)
// int4 caches (I4): K [B, Hkv, Lmax, D/2] bytes with one fp16 scale per key and 32 channels (ks [B, Hkv, Lmax, D/32]):
// the 8 channels 8i .. 8i+7 of a key form one 32-bit word, channel j in nibble (j >> 1) + 4 (j & 1), value
// (nibble - 8) * scale.  V^T [B, Hkv, D/2, Lmax] bytes: row 16m + r holds channel 32m + r (low nibble) and 32m + 16 + r
// (high nibble) of every key, scales vs [B, Hkv, D/32, Lmax] per key and 32 channels.  Appending a token writes whole
// bytes in both caches; the V scale is applied to the probabilities (once per 32-channel group), the K scale to the
// dequantized fragment.
// In both products the query row is the MFMA column (= lane & 15), so the online-softmax
// statistics and the O rescale are lane-local.  A 32-key step feeds the two 16-key S^T tiles
// with keys {8g+q} and {8g+4+q} so that lane group g ends up owning 8 consecutive keys, which
// is exactly the k-slice its P^T fragment must hold for the O^T product.
#include "common.h"
#include <stdlib.h>

namespace bigdl {

struct AttnArgs {
  _Float16* qkv;         // [B*q_total, ld]  q heads | k heads | v heads
  void* kc;              // K cache
  void* vc;              // V^T cache
  _Float16* out;         // [B*q_total, Hq*D]
  float* ws;             // partials [B*Hkv][splits][16*RT rows][D + 4]  (O row, then m, l)
  int* counters;         // [B*Hkv] zero before first use; reset by the last arriver
  const int* past_len;   // [B] tokens in the cache before this step (device, graph-replayable)
  const _Float16* mask;  // optional additive mask [B, q_total, mask_ld] or null
  const float* rope;     // fused RoPE: table [max_pos][rot/2][2] (cos, sin) or null (q/k already rotated+appended)
  const int* position_ids;  // optional [B*q_total] rope positions (default: cache position)
  const float* alibi;       // optional [Hq] ALiBi slopes: score += slope[head] * (key - query position)
  _Float16* ks;             // int4 caches: K scales [B, Hkv, Lmax, D/32]
  _Float16* vs;             // int4 caches: V scales [B, Hkv, D/32, Lmax]
  int mask_ld;
  float scale;           // softmax scale
  float softcap;         // gemma2 logit soft-capping (0 = off)
  int B, Hq, Hkv, ld, Lmax, splits;
  int q_total, q_off, q_cnt;   // this launch handles new tokens [q_off, q_off+q_cnt) of q_total
  int window;            // sliding window (0 = off): keys < pos - window + 1 are masked
  int rot, interleaved, rope_max_pos;
  int defer_combine;     // 1: stop after writing the per-split partials (the consumer GEMV combines them)
  int nchunk, gchunk;    // query-head chunks per kv head and heads per chunk (1, Hq/Hkv unless Hq/Hkv > 32)
  int ablate;            // measurement builds only (bigdl_attn_set_ablation): 1 = K / V loads without the products
};

constexpr float NEG_BIG = -1.0e30f;
constexpr int ATT_NW = 8;          // wavefronts per workgroup

template <bool FP8>
__device__ __forceinline__ half8 load_kv8(const void* base, int64_t elem_off) {
  if constexpr (FP8) {
    const uint2v w = *(const uint2v*)((const uint8_t*)base + elem_off);
    return dec_e5m2(w.x, w.y);
  } else {
    return *(const half8*)((const _Float16*)base + elem_off);
  }
}

// 8 int4 values of one cache word -> fp16 (nibble k -> element 2k, nibble k + 4 -> element 2k + 1), minus the zero
// point 8, times `sc`.  0x6400 | n is the fp16 number 1024 + n.
__device__ __forceinline__ half8 dq_i4(uint32_t w, _Float16 sc) {
  H8 r;
  const half2v off = {(_Float16)1032.0f, (_Float16)1032.0f}, s2 = {sc, sc};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t t = ((w >> (4 * k)) & 0x000F000Fu) | 0x64006400u;
    r.h2[k] = (__builtin_bit_cast(half2v, t) - off) * s2;
  }
  return r.v;
}

// symmetric int4 code of x for the scale 1 / inv
__device__ __forceinline__ uint32_t q_i4(float x, float inv) {
  return (uint32_t)(int)fminf(fmaxf(rintf(x * inv) + 8.0f, 0.0f), 15.0f);
}

// e5m2 caches feed the matrix cores as they are: v_mfma_f32_16x16x32_bf8_fp8 takes the cache bytes (bf8 = OCP e5m2) as
// its A operand, 8 bytes per lane and k-step, and an e4m3 B operand.  The fp32 query row and the probabilities are
// split into two e4m3 terms each (value ~ hi + lo, about 7 significant bits; two products into the same accumulator),
// after a per-row scale that puts the row maximum at 256: no byte -> fp16 expansion, and the K / V^T fragments stay in
// registers as the bytes that were loaded (a quarter of the registers of the fp16 fragments).
__device__ __forceinline__ float4v mfma_bf8_fp8(uint2v a, uint2v b, float4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), c, 0, 0, 0);
}
// 8 floats -> 8 e4m3 bytes (hi) and the e4m3 bytes of what the first rounding left over (lo)
__device__ __forceinline__ void split_e4m3(const float (&x)[8], uint2v& hi, uint2v& lo) {
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    int h = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * w], x[4 * w + 1], 0, false);
    h = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * w + 2], x[4 * w + 3], h, true);
    const float r0 = x[4 * w] - __builtin_amdgcn_cvt_f32_fp8(h, 0), r1 = x[4 * w + 1] - __builtin_amdgcn_cvt_f32_fp8(h, 1);
    const float r2 = x[4 * w + 2] - __builtin_amdgcn_cvt_f32_fp8(h, 2), r3 = x[4 * w + 3] - __builtin_amdgcn_cvt_f32_fp8(h, 3);
    int l = __builtin_amdgcn_cvt_pk_fp8_f32(r0, r1, 0, false);
    l = __builtin_amdgcn_cvt_pk_fp8_f32(r2, r3, l, true);
    hi[w] = (uint32_t)h;
    lo[w] = (uint32_t)l;
  }
}
constexpr float P8_SCALE = 256.0f;     // probabilities are carried as 256 p (and the row sums with them)

// D = head dim (multiple of 32), RT = number of 16-row tiles, KW = keys per wavefront per iteration.
// KW = 64 (e5m2 caches, D % 128 == 0): every lane loads 16 bytes of a key row / of a V^T channel row instead of 8, i.e.
// half as many load instructions and 64-byte instead of 32-byte contiguous pieces per 4 lanes.  With 8-byte pieces the
// e5m2 path is bound by the number of memory instructions, not by bytes (batch 64, 1024 keys: 184 us per layer where
// the fp16 cache needs 258 us for twice the bytes).  The MFMA k index <-> head channel assignment is free as long as
// Q and K agree, so lane group g takes channels [16 (g + 4 c2), +16) of a row for the k-steps 2 c2 and 2 c2 + 1.
template <int D, bool FP8, int RT, int KW, bool I4, int NWT = ATT_NW, int MINW = 1>
__global__ __launch_bounds__(NWT * 64, MINW) void attn_decode_kernel(AttnArgs a) {
  constexpr int DC = D / 32;   // QK k-steps
  constexpr int DT = D / 16;   // O^T d-tiles
  constexpr int NW = NWT;
  constexpr int ATT_KEYS = NW * KW;    // keys per workgroup iteration
  constexpr int NU = KW / 16;          // 16-key score tiles per wavefront
  constexpr int KL = KW / 4;           // consecutive keys per lane
  constexpr bool WIDE = KW == 64;
  // e5m2 bytes straight into the fp8 matrix instruction (below): the many-keys-per-workgroup form.  The latency-bound
  // batch-1 form keeps fp16 fragments: splitting q into two e4m3 terms in its prologue cost 0.2 us per launch there.
  constexpr bool F8M = FP8 && WIDE;
  static_assert(!WIDE || (FP8 && D % 128 == 0), "16-byte fragments: e5m2 cache, rotate-half partner in k-step c + DC/2");
  static_assert(!I4 || (!FP8 && KW == 32), "int4 caches: 32 keys per wavefront");
  // first head channel of this lane's 8 values in k-step c.  int4: lane group g owns the D/4 contiguous channels
  // (D/4) g .. of a key, one load of D/8 bytes
  auto chan = [](int c, int g) {
    return I4 ? (D / 4) * g + 8 * c : WIDE ? 64 * (c >> 1) + 16 * g + 8 * (c & 1) : 32 * c + 8 * g;
  };
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rr = lane & 15, g = lane >> 4;
  const int split = blockIdx.x, b = blockIdx.z;
  // a kv head's group of query heads is processed in `nchunk` chunks of at most `gchunk` heads, one workgroup column
  // each (multi-query models with 48 / 71 query heads per kv head do not fit one 32-row tile pair)
  const int G = a.Hq / a.Hkv;
  const int hkv = blockIdx.y / a.nchunk, g0 = (blockIdx.y % a.nchunk) * a.gchunk;
  const int Gc = min(a.gchunk, G - g0);
  const int rows = Gc * a.q_cnt;
  // merge buffers: per wave m,l per row, and O rows (only `rows` rows are ever touched)
  float* s_m = (float*)smem;                       // [NW][RT*16]
  float* s_l = s_m + NW * RT * 16;                 // [NW][RT*16]
  float* s_o = s_l + NW * RT * 16;                 // [NW][rows][D]
  int* s_flag = (int*)(s_o + NW * rows * D);

  const int past = a.past_len[b];
  const int kv_len = past + a.q_total;
  // split ranges in units of ATT_KEYS keys
  const int nblk = (kv_len + ATT_KEYS - 1) / ATT_KEYS;
  const int per = (nblk + a.splits - 1) / a.splits;
  const int blk0 = split * per, blk1 = min(nblk, blk0 + per);

  const int64_t kbase = ((int64_t)b * a.Hkv + hkv) * (int64_t)a.Lmax * D;
  char* kc = (char*)a.kc + (I4 ? kbase / 2 : FP8 ? kbase : kbase * 2);
  char* vc = (char*)a.vc + (I4 ? kbase / 2 : FP8 ? kbase : kbase * 2);
  _Float16* ks = I4 ? a.ks + kbase / 32 : nullptr;
  _Float16* vs = I4 ? a.vs + kbase / 32 : nullptr;
  const int half_rot = a.rot >> 1;

  // e5m2 caches: one block's K / V^T fragments as the bytes that were loaded
  struct RawK { uint2v w[F8M ? NU : 1][F8M ? DC : 1]; };
  struct RawV { uint2v w[F8M ? DT : 1][F8M ? KL / 8 : 1]; };
  auto load_raw = [&](const int kb, RawK& kr, RawV& vr) {
    const int key0 = kb + KL * (rr >> 2) + (rr & 3);
    if constexpr (WIDE) {
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int c2 = 0; c2 < DC / 2; ++c2) {
          const uint4v w = *(const uint4v*)((const uint8_t*)kc + (int64_t)(key0 + 4 * u) * D + 64 * c2 + 16 * g);
          kr.w[u][2 * c2] = uint2v{w[0], w[1]};
          kr.w[u][2 * c2 + 1] = uint2v{w[2], w[3]};
        }
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const uint4v w = *(const uint4v*)((const uint8_t*)vc + (int64_t)(16 * d + rr) * a.Lmax + kb + 16 * g);
        vr.w[d][0] = uint2v{w[0], w[1]};
        vr.w[d][1] = uint2v{w[2], w[3]};
      }
    }
  };
  // ---- fused RoPE(k) + KV append for the new tokens that fall into this workgroup's key range ----
  if (a.rope != nullptr || a.rot == 0) {
    if (a.q_off == 0) {
      const int lo = blk0 * ATT_KEYS, hi = blk1 * ATT_KEYS;
      for (int t = wave; t < a.q_total; t += NW) {
        const int p = past + t;
        if (p < lo || p >= hi || p >= a.Lmax) continue;
        const int tokidx = b * a.q_total + t;
        const int pos = a.position_ids ? a.position_ids[tokidx] : p;
        const _Float16* krow = a.qkv + (int64_t)tokidx * a.ld + (int64_t)(a.Hq + hkv) * D;
        const _Float16* vrow = krow + (int64_t)a.Hkv * D;
        _Float16* stage = (_Float16*)s_o + wave * D;     // int4: the rotated key, quantized below
        for (int i = lane; i < D / 2; i += 64) {
          int i0, i1;
          const bool rot = i < half_rot;
          if (rot) {
            if (a.interleaved) { i0 = 2 * i; i1 = 2 * i + 1; }
            else { i0 = i; i1 = i + half_rot; }
          } else {
            i0 = a.rot + 2 * (i - half_rot);
            i1 = i0 + 1;
          }
          float x0 = (float)krow[i0], x1 = (float)krow[i1];
          if (rot) {
            const float2 cs = *(const float2*)(a.rope + ((int64_t)min(pos, a.rope_max_pos - 1) * half_rot + i) * 2);
            const float y0 = x0 * cs.x - x1 * cs.y, y1 = x1 * cs.x + x0 * cs.y;
            x0 = y0;
            x1 = y1;
          }
          if constexpr (I4) {
            stage[i0] = (_Float16)x0;
            stage[i1] = (_Float16)x1;
          } else if constexpr (FP8) {
            ((uint8_t*)kc)[(int64_t)p * D + i0] = f16_to_e5m2((_Float16)x0);
            ((uint8_t*)kc)[(int64_t)p * D + i1] = f16_to_e5m2((_Float16)x1);
          } else {
            ((_Float16*)kc)[(int64_t)p * D + i0] = (_Float16)x0;
            ((_Float16*)kc)[(int64_t)p * D + i1] = (_Float16)x1;
          }
        }
        if constexpr (I4) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          __builtin_amdgcn_wave_barrier();
          // K: lane l < D/8 packs channels 8l .. 8l+7; the four lanes of a 32-channel group share the scale
          {
            const bool on = lane < D / 8;
            half8 kv8 = {0, 0, 0, 0, 0, 0, 0, 0};
            if (on) kv8 = *(const half8*)(stage + 8 * lane);
            float am = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf((float)kv8[j]));
            am = fmaxf(am, __shfl_xor(am, 1));
            am = fmaxf(am, __shfl_xor(am, 2));
            const _Float16 sc = (_Float16)(am * (1.0f / 7.0f));
            const float inv = (float)sc > 0.0f ? 1.0f / (float)sc : 0.0f;
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) w |= q_i4((float)kv8[j], inv) << (4 * ((j >> 1) + 4 * (j & 1)));
            if (on) {
              *(uint32_t*)(kc + (int64_t)p * (D / 2) + 4 * lane) = w;
              if ((lane & 3) == 0) ks[(int64_t)p * (D / 32) + (lane >> 2)] = sc;
            }
          }
          // V: lane r < D/2 writes the byte of V^T row r = 16m + rr: channels 32m + rr and 32m + 16 + rr
#pragma unroll
          for (int r0 = 0; r0 < D / 2; r0 += 64) {
            const int r = r0 + lane;
            const bool on = r < D / 2;
            const int m = r >> 4, c0 = 32 * m + (r & 15);
            const float lo = on ? (float)vrow[c0] : 0.0f, hi = on ? (float)vrow[c0 + 16] : 0.0f;
            float am = fmaxf(fabsf(lo), fabsf(hi));
#pragma unroll
            for (int x = 1; x < 16; x <<= 1) am = fmaxf(am, __shfl_xor(am, x));
            const _Float16 sc = (_Float16)(am * (1.0f / 7.0f));
            const float inv = (float)sc > 0.0f ? 1.0f / (float)sc : 0.0f;
            if (on) {
              ((uint8_t*)vc)[(int64_t)r * a.Lmax + p] = (uint8_t)(q_i4(lo, inv) | (q_i4(hi, inv) << 4));
              if ((r & 15) == 0) vs[(int64_t)m * a.Lmax + p] = sc;
            }
          }
          continue;
        }
        if constexpr (F8M)
          if (a.ablate & 2) continue;   // (measurement: without the V^T byte stores, 128 partly written lines per key)
        for (int d = lane; d < D; d += 64) {
          if constexpr (FP8) ((uint8_t*)vc)[(int64_t)d * a.Lmax + p] = f16_to_e5m2(vrow[d]);
          else ((_Float16*)vc)[(int64_t)d * a.Lmax + p] = vrow[d];
        }
      }
    }
    __syncthreads();   // appended keys become visible to the whole workgroup (vmcnt(0) + barrier)
  }

  const float qscale = a.softcap > 0.0f ? a.scale : a.scale * 1.4426950408889634f;

  // Q^T fragments (B operand): lane (row, g) holds d = 32c + 8g + j
  half8 qf[F8M ? 1 : RT][F8M ? 1 : DC];
  uint2v qh[F8M ? RT : 1][F8M ? DC : 1], ql[F8M ? RT : 1][F8M ? DC : 1];   // e5m2 caches: e4m3 hi / lo terms of 256 q / max|q|
  float q_unscale[RT];                                                       // max|q| / 256 (folds the softmax scale)
  int lim[RT];  // last visible key position per row (causal within the new tokens)
  int lo_lim[RT];
  float slope[RT];   // ALiBi slope of the row's head, in log2 units (0 without ALiBi)
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int row = 16 * t + rr;
    const int tok = a.q_off + row / Gc, gq = g0 + row % Gc;
    const bool valid = row < rows;
    slope[t] = a.alibi != nullptr ? a.alibi[hkv * G + gq] * 1.4426950408889634f : 0.0f;
    const int tokidx = b * a.q_total + tok;
    lim[t] = valid ? past + tok : -1;
    lo_lim[t] = (a.window > 0 && valid) ? max(0, past + tok - a.window + 1) : 0;
    const _Float16* qp = a.qkv + (int64_t)tokidx * a.ld + (hkv * G + gq) * D;
    float qv[DC][8];
#pragma unroll
    for (int c = 0; c < DC; ++c) {
      half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (valid) v = *(const half8*)(qp + chan(c, g));
#pragma unroll
      for (int j = 0; j < 8; ++j) qv[c][j] = (float)v[j];
    }
    if (a.rope != nullptr && valid) {
      const int pos = min(a.position_ids ? a.position_ids[tokidx] : past + tok, a.rope_max_pos - 1);
      const float* tab = a.rope + (int64_t)pos * half_rot * 2;
      if (a.interleaved) {
#pragma unroll
        for (int c = 0; c < DC; ++c) {
          const int d0 = chan(c, g);
          if (d0 < a.rot) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              const float2 cs = *(const float2*)(tab + (d0 + j));   // pair index (d0+j)/2 -> float offset d0+j
              const float x0 = qv[c][j], x1 = qv[c][j + 1];
              qv[c][j] = x0 * cs.x - x1 * cs.y;
              qv[c][j + 1] = x1 * cs.x + x0 * cs.y;
            }
          }
        }
      } else if constexpr (I4) {
        // rotate-half pairs (d, d + D/2) with the int4 channel order: the partner of lane group g is group g ^ 2
        // (lane ^ 32), same k-step.  Invalid rows take part in the shuffles below with zeros.
      } else {
        // rotate-half pairs (d, d + D/2): the host only enables the fused path for full rotary
        // (rot == D), so the partner lives in chunk c + DC/2 of the same lane
#pragma unroll
        for (int c = 0; c < DC / 2; ++c) {
          const int d0 = chan(c, g);          // < D/2 in both channel orders; the partner d0 + D/2 is k-step c + DC/2
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float2 cs = *(const float2*)(tab + (d0 + j) * 2);
            const float x0 = qv[c][j], x1 = qv[c + DC / 2][j];
            qv[c][j] = x0 * cs.x - x1 * cs.y;
            qv[c + DC / 2][j] = x1 * cs.x + x0 * cs.y;
          }
        }
      }
    }
    if constexpr (I4) {
      if (a.rope != nullptr && !a.interleaved) {
        const int pos = min(a.position_ids ? a.position_ids[valid ? tokidx : 0] : past + tok, a.rope_max_pos - 1);
        const float* tab = a.rope + (int64_t)max(pos, 0) * half_rot * 2;
        const float sgn = g < 2 ? -1.0f : 1.0f;
#pragma unroll
        for (int c = 0; c < DC; ++c) {
          const int d0 = chan(c, g) & (D / 2 - 1);       // index of the pair in the table
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float other = __shfl_xor(qv[c][j], 32);
            const float2 cs = *(const float2*)(tab + (d0 + j) * 2);
            qv[c][j] = qv[c][j] * cs.x + sgn * other * cs.y;
          }
        }
      }
    }
    if constexpr (F8M) {
      float am = 0.0f;
#pragma unroll
      for (int c = 0; c < DC; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf(qv[c][j]));
      am = fmaxf(am, __shfl_xor(am, 16));
      am = fmaxf(am, __shfl_xor(am, 32));
      const float f = am > 0.0f ? P8_SCALE / am : 0.0f;
      q_unscale[t] = am * (qscale / P8_SCALE);
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = qv[c][j] * f;
        split_e4m3(x, qh[t][c], ql[t][c]);
      }
    } else {
      q_unscale[t] = 1.0f;
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)(qv[c][j] * qscale);
        qf[t][c] = v;
      }
    }
  }

  float m_run[RT], l_run[RT];
  float4v o[RT][DT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    m_run[t] = NEG_BIG;
    l_run[t] = 0.0f;
#pragma unroll
    for (int d = 0; d < DT; ++d) o[t][d] = float4v{0, 0, 0, 0};
  }
  const float4v zero4 = {0, 0, 0, 0};

  // the KW keys from kb on: K / V^T fragments loaded (e5m2: by load_raw) and folded into the running softmax state
  auto process = [&](const int kb, const RawK& kr, const RawV& vr) {
    // ---- S^T = K . Q^T for NU 16-key tiles ----
    // A-row i of tile u  <->  key kb + KL*(i>>2) + 4*u + (i&3), so that lane group g ends up with the KL consecutive
    // keys kb + KL*g .. of every query row
    const int key0 = kb + KL * (rr >> 2) + (rr & 3);
    half8 kf[F8M ? 1 : NU][F8M ? 1 : DC];
    if constexpr (I4) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int64_t key = key0 + 4 * u;
        uint32_t w[DC];
        const uint8_t* src = (const uint8_t*)kc + key * (D / 2) + (D / 8) * g;
        if constexpr (DC == 2) {
          const uint2v t = *(const uint2v*)src;
          w[0] = t.x; w[1] = t.y;
        } else {
#pragma unroll
          for (int c4 = 0; c4 < DC / 4; ++c4) {
            const uint4v t = *(const uint4v*)(src + 16 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[4 * c4 + e] = t[e];
          }
        }
        constexpr int SPL = D >= 128 ? D / 128 : 1;       // scales per lane (32 channels each)
        _Float16 sc[SPL];
#pragma unroll
        for (int e = 0; e < SPL; ++e) sc[e] = ks[key * (D / 32) + ((D / 4) * g) / 32 + e];
#pragma unroll
        for (int c = 0; c < DC; ++c) kf[u][c] = dq_i4(w[c], sc[SPL > 1 ? c / 4 : 0]);
      }
    } else if constexpr (!F8M) {
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int c = 0; c < DC; ++c) kf[u][c] = load_kv8<FP8>(kc, (int64_t)(key0 + 4 * u) * D + 32 * c + 8 * g);
    }
    // V^T fragments: lane (d_local, g) holds keys kb + KL*g .. of channel 16*dt + d_local (KL/8 k-steps)
    half8 vf[F8M ? 1 : DT][F8M ? 1 : KL / 8];
    half8 vsc[I4 ? DT / 2 : 1];     // int4: V scales of this lane's 8 keys, per 32-channel group
    if constexpr (I4) {
      const _Float16 one = (_Float16)1.0f;
      const half2v off = {(_Float16)1032.0f, (_Float16)1032.0f};
#pragma unroll
      for (int m = 0; m < DT / 2; ++m) {
        const uint2v t = *(const uint2v*)((const uint8_t*)vc + (int64_t)(16 * m + rr) * a.Lmax + kb + 8 * g);
        vsc[m] = *(const half8*)(vs + (int64_t)m * a.Lmax + kb + 8 * g);
        H8 lo, hi;
        // byte q of the pair of words is key q: pairs come out as keys (0,2) (1,3) (4,6) (5,7)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const uint32_t w = e ? t.y : t.x;
          lo.h2[2 * e] = __builtin_bit_cast(half2v, (w & 0x000F000Fu) | 0x64006400u) - off;
          hi.h2[2 * e] = __builtin_bit_cast(half2v, ((w >> 4) & 0x000F000Fu) | 0x64006400u) - off;
          lo.h2[2 * e + 1] = __builtin_bit_cast(half2v, ((w >> 8) & 0x000F000Fu) | 0x64006400u) - off;
          hi.h2[2 * e + 1] = __builtin_bit_cast(half2v, ((w >> 12) & 0x000F000Fu) | 0x64006400u) - off;
        }
        vf[2 * m][0] = lo.v;
        vf[2 * m + 1][0] = hi.v;
      }
      (void)one;
    } else if constexpr (!F8M) {
#pragma unroll
      for (int d = 0; d < DT; ++d) vf[d][0] = load_kv8<FP8>(vc, (int64_t)(16 * d + rr) * a.Lmax + kb + 8 * g);
    }

    if constexpr (F8M) {
      if (a.ablate & 1) {
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int c = 0; c < DC; ++c) asm volatile("" ::"v"(kr.w[u][c].x), "v"(kr.w[u][c].y));
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int h = 0; h < KL / 8; ++h) asm volatile("" ::"v"(vr.w[d][h].x), "v"(vr.w[d][h].y));
        return;
      }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      // lane (row rr, g): tile u, register q  <->  key kb + KL*g + 4*u + q
      float sv[KL];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        float4v su = zero4;
        if constexpr (F8M) {
#pragma unroll
          for (int c = 0; c < DC; ++c) {
            su = mfma_bf8_fp8(kr.w[u][c], qh[t][c], su);
            su = mfma_bf8_fp8(kr.w[u][c], ql[t][c], su);
          }
        } else {
#pragma unroll
          for (int c = 0; c < DC; ++c) su = mfma16(kf[u][c], qf[t][c], su);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sv[4 * u + q] = F8M ? su[q] * q_unscale[t] : su[q];
      }
      if (a.softcap > 0.0f) {
#pragma unroll
        for (int q = 0; q < KL; ++q) sv[q] = a.softcap * tanhf(sv[q] / a.softcap) * 1.4426950408889634f;
      }
      const int kfirst = kb + KL * g;
      if (a.alibi != nullptr) {
#pragma unroll
        for (int q = 0; q < KL; ++q) sv[q] += slope[t] * (float)(kfirst + q - lim[t]);   // slope * (key - query)
      }
      if (a.mask) {
        const int row = 16 * t + rr;
        if (row < rows) {
          const _Float16* mp = a.mask + ((int64_t)b * a.q_total + a.q_off + row / Gc) * a.mask_ld + kfirst;
#pragma unroll
          for (int q = 0; q < KL; ++q)
            if (kfirst + q < a.mask_ld) sv[q] += (float)mp[q] * 1.4426950408889634f;
        }
      }
      float mx = NEG_BIG;
#pragma unroll
      for (int q = 0; q < KL; ++q) {
        const int key = kfirst + q;
        if (key > lim[t] || key < lo_lim[t]) sv[q] = NEG_BIG;
        mx = fmaxf(mx, sv[q]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[t], mx);
      const float alpha = exp2f(m_run[t] - m_new);
      m_run[t] = m_new;
      H8 pf[KL / 8];
      float ps = 0.0f;
      if constexpr (F8M) {
        // masked entries: sv = NEG_BIG; when the whole row is still masked m_new = NEG_BIG too and exp2(0) would be 1,
        // so masked entries are forced to 0 explicitly
#pragma unroll
        for (int q = 0; q < KL; ++q) {
          sv[q] = sv[q] <= NEG_BIG ? 0.0f : P8_SCALE * exp2f(sv[q] - m_new);
          ps += sv[q];
        }
      } else {
#pragma unroll
        for (int q = 0; q < KL; q += 2) {
          const float p0 = sv[q] <= NEG_BIG ? 0.0f : exp2f(sv[q] - m_new);
          const float p1 = sv[q + 1] <= NEG_BIG ? 0.0f : exp2f(sv[q + 1] - m_new);
          ps += p0 + p1;
          pf[q >> 3].h2[(q & 7) >> 1] = half2v{(_Float16)p0, (_Float16)p1};
        }
      }
      l_run[t] = l_run[t] * alpha + ps;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int d = 0; d < DT; ++d) o[t][d] *= alpha;
      }
      if constexpr (F8M) {
#pragma unroll
        for (int h = 0; h < KL / 8; ++h) {
          uint2v ph, pl;
          const float(&p8)[8] = *(const float(*)[8])&sv[8 * h];
          split_e4m3(p8, ph, pl);
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            o[t][d] = mfma_bf8_fp8(vr.w[d][h], ph, o[t][d]);
            o[t][d] = mfma_bf8_fp8(vr.w[d][h], pl, o[t][d]);
          }
        }
        continue;
      }
      if constexpr (I4) {
        // probabilities in the key order of the V^T fragments, times the V scale of each 32-channel group
        const half8 p8 = pf[0].v;
        H8 pp;
        pp.h2[0] = half2v{p8[0], p8[2]};
        pp.h2[1] = half2v{p8[1], p8[3]};
        pp.h2[2] = half2v{p8[4], p8[6]};
        pp.h2[3] = half2v{p8[5], p8[7]};
#pragma unroll
        for (int m = 0; m < DT / 2; ++m) {
          H8 pm;
          pm.h2[0] = pp.h2[0] * half2v{vsc[m][0], vsc[m][2]};
          pm.h2[1] = pp.h2[1] * half2v{vsc[m][1], vsc[m][3]};
          pm.h2[2] = pp.h2[2] * half2v{vsc[m][4], vsc[m][6]};
          pm.h2[3] = pp.h2[3] * half2v{vsc[m][5], vsc[m][7]};
          o[t][2 * m] = mfma16(vf[2 * m][0], pm.v, o[t][2 * m]);
          o[t][2 * m + 1] = mfma16(vf[2 * m + 1][0], pm.v, o[t][2 * m + 1]);
        }
        continue;
      }
#pragma unroll
      for (int h = 0; h < KL / 8; ++h)
#pragma unroll
        for (int d = 0; d < DT; ++d) o[t][d] = mfma16(vf[d][h], pf[h].v, o[t][d]);
    }
  };

  if constexpr (F8M) {
    // Requesting block i + 1 before block i is worked on (a second fragment set, 224 registers) and requesting the
    // first block ahead of the append prologue were both measured: no gain at batch 64 (the loop already streams at the
    // rate this access pattern reaches, 5.3 TB/s) and a slower batch-1 launch (vector memory returns in order, so the
    // prologue's small dependent loads queue behind 16 KiB of K / V per wavefront).
    for (int blk = blk0; blk < blk1; ++blk) {
      const int kb = blk * ATT_KEYS + wave * KW;
      if (kb >= kv_len) break;
      RawK kr;
      RawV vr;
      load_raw(kb, kr, vr);
      process(kb, kr, vr);
    }
  } else {
    for (int blk = blk0; blk < blk1; ++blk) {
      const int kb = blk * ATT_KEYS + wave * KW;
      if (kb >= kv_len) break;
      process(kb, RawK{}, RawV{});
    }
  }

  // ---- merge the wavefronts through LDS ----
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    float l = l_run[t];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (g == 0) {
      s_m[wave * RT * 16 + 16 * t + rr] = m_run[t];
      s_l[wave * RT * 16 + 16 * t + rr] = l;
    }
    if (16 * t + rr < rows) {
#pragma unroll
      for (int d = 0; d < DT; ++d)
        *(float4v*)&s_o[((wave * rows) + 16 * t + rr) * D + 16 * d + 4 * g] = o[t][d];
    }
  }
  __syncthreads();

  const int bh = b * gridDim.y + blockIdx.y;
  const bool single = a.splits == 1;
  float* wsp = a.ws + ((int64_t)bh * a.splits + split) * (RT * 16) * (D + 4);
  // thread -> (row, d-chunk): merge the NW wavefronts; with one split this is the final result
  for (int idx = threadIdx.x; idx < rows * (D / 4); idx += NW * 64) {
    const int row = idx / (D / 4), d4 = (idx % (D / 4)) * 4;
    float mw[NW], mm = NEG_BIG;
#pragma unroll
    for (int w = 0; w < NW; ++w) { mw[w] = s_m[w * RT * 16 + row]; mm = fmaxf(mm, mw[w]); }
    float lt = 0.0f;
    float4v acc = {0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float f = exp2f(mw[w] - mm);
      lt += f * s_l[w * RT * 16 + row];
      acc += *(const float4v*)&s_o[(w * rows + row) * D + d4] * f;
    }
    if (single) {
      const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;
      const int tok = a.q_off + row / Gc, gq = g0 + row % Gc;
      _Float16* dst = a.out + (int64_t)(b * a.q_total + tok) * (a.Hq * D) + (hkv * G + gq) * D + d4;
      half4 hv = {(_Float16)(acc[0] * inv), (_Float16)(acc[1] * inv), (_Float16)(acc[2] * inv), (_Float16)(acc[3] * inv)};
      *(half4*)dst = hv;
    } else {
      float* dst = wsp + row * (D + 4);
      *(float4v*)(dst + d4) = acc;
      if (d4 == 0) { dst[D] = mm; dst[D + 1] = lt; }
    }
  }
  if (single || a.defer_combine) return;

  // ---- cross-split combine by the last-arriving workgroup of this (b, hkv) ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int ticket = __hip_atomic_fetch_add(&a.counters[bh], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == a.splits - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&a.counters[bh], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *s_flag = last;
  }
  __syncthreads();
  if (!*s_flag) return;

  const float* wbase = a.ws + (int64_t)bh * a.splits * (RT * 16) * (D + 4);
  for (int idx = threadIdx.x; idx < rows * (D / 4); idx += NW * 64) {
    const int row = idx / (D / 4), d4 = (idx % (D / 4)) * 4;
    float mm = NEG_BIG;
    for (int s = 0; s < a.splits; ++s) mm = fmaxf(mm, wbase[((int64_t)s * RT * 16 + row) * (D + 4) + D]);
    float lt = 0.0f;
    float4v acc = {0, 0, 0, 0};
    for (int s = 0; s < a.splits; ++s) {
      const float* src = wbase + ((int64_t)s * RT * 16 + row) * (D + 4);
      const float f = exp2f(src[D] - mm);
      lt += f * src[D + 1];
      acc += *(const float4v*)(src + d4) * f;
    }
    const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;
    const int tok = a.q_off + row / Gc, gq = g0 + row % Gc;
    _Float16* dst = a.out + (int64_t)(b * a.q_total + tok) * (a.Hq * D) + (hkv * G + gq) * D + d4;
    half4 hv = {(_Float16)(acc[0] * inv), (_Float16)(acc[1] * inv), (_Float16)(acc[2] * inv), (_Float16)(acc[3] * inv)};
    *(half4*)dst = hv;
  }
}

template <int D, bool FP8, int RT, int KW, bool I4 = false, int NWT = ATT_NW, int MINW = 1>
static int launch_one(const AttnArgs& a, size_t, hipStream_t st) {
  const dim3 grid(a.splits, a.Hkv * a.nchunk, a.B), block(NWT * 64);
  const int rows = a.gchunk * a.q_cnt;
  const size_t smem = (size_t)(NWT * RT * 16 * 2 + NWT * rows * D) * 4 + 16;
  // more than 64 KiB of dynamic LDS has to be requested explicitly (once per kernel)
  static bool attr = false;
  if (!attr && smem > 65536) {
    if (hipFuncSetAttribute((const void*)attn_decode_kernel<D, FP8, RT, KW, I4, NWT, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return -5;
    attr = true;
  }
  hipLaunchKernelGGL((attn_decode_kernel<D, FP8, RT, KW, I4, NWT, MINW>), grid, block, smem, st, a);
  return HIP_CHECK_LAUNCH();
}

template <int D, bool FP8, bool I4 = false>
static int launch_rt(const AttnArgs& a, hipStream_t st) {
  const int rows = a.gchunk * a.q_cnt;
  const int rt = (rows + 15) / 16;
  const size_t smem = (size_t)(ATT_NW * rt * 16 * 2 + ATT_NW * rows * D) * 4 + 16;
  if constexpr (I4) {
    if (rt == 1) return launch_one<D, false, 1, 32, true>(a, smem, st);
    if (rt == 2) return launch_one<D, false, 2, 32, true>(a, smem, st);
    return -4;
  }
  if (rt == 1) {
    // 64 keys per wavefront (16-byte fragments).  With one KV split and fewer than 512 (sequence, head) pairs the
    // workgroup has ATT_NW wavefronts; with several splits or 512 pairs and more it has four (below).
    if constexpr (FP8 && D % 128 == 0) {
      static const char* force = getenv("BIGDL_ATTN_KEYS");      // "32" / "64": override for measurements
      const bool wide = force ? atoi(force) == 64 : true;
      // batched decode (512 or more (sequence, head) workgroups, one KV split): four-wavefront workgroups, so
      // that two of them share a CU and one streams keys while the other is in its softmax / merge phase.  Measured
      // on Llama-2-7B at 1024-token context: batch 64 9.36 -> 8.65 ms per step, batch 16 4.34 -> 4.20.  Tried on top
      // and dropped: two-wavefront workgroups (8.94), keys / values kept as bytes in registers (no fewer live
      // registers: 186 either way), a second register set that prefetches the next block (256 registers, 8.83).
      static const char* nw4 = getenv("BIGDL_ATTN_NW4");           // "0" / "1": override for measurements
      // Several KV splits (few pairs, each split a short key range): the same four-wavefront form, one 256-key pass
      // per wavefront round instead of ATT_NW x 32 keys, and a four-way merge in LDS instead of an ATT_NW-way one.
      // Measured per decode step on one MI355X, same box, 32-key ATT_NW-wavefront form -> this one: Llama-2-7B batch 1
      // 1.551 -> 1.511 ms at 1024 keys, 1.875 -> 1.781 at 4096; Llama-3-8B 1.642 -> 1.612 / 1.834 -> 1.756;
      // Qwen2-7B 1.557 -> 1.532 / 1.714 -> 1.637.  With ONE split and fewer than 512 pairs it loses (Llama-2-7B
      // batch 2 at 1024 keys 1.857 -> 1.899), so that case keeps ATT_NW wavefronts.
      const bool small = nw4 ? atoi(nw4) == 1 : (a.splits > 1 || (int64_t)a.B * a.Hkv * a.nchunk >= 512);
      if constexpr (D == 128) {
        static const char* occ = getenv("BIGDL_ATTN_OCC");         // wavefronts per SIMD the register budget allows
        if (wide && small && occ && atoi(occ) == 3) return launch_one<D, FP8, 1, 64, false, 4, 3>(a, smem, st);
        if (wide && small && occ && atoi(occ) == 4) return launch_one<D, FP8, 1, 64, false, 4, 4>(a, smem, st);
      }
      if (wide && small) return launch_one<D, FP8, 1, 64, false, 4, (D <= 128 ? 2 : 1)>(a, smem, st);
      if (wide) return launch_one<D, FP8, 1, 64>(a, smem, st);
    }
    return launch_one<D, FP8, 1, 32>(a, smem, st);
  }
  if (rt == 2) return launch_one<D, FP8, 2, 32>(a, smem, st);
  return -4;
}

// ---------------------------------------------------------------------------------------------
// RoPE (in place on q, and on k while it is written to the cache) + V^T append
// ---------------------------------------------------------------------------------------------
struct RopeAppendArgs {
  _Float16* qkv;          // [B*q_len, ld]
  void* kc;
  void* vc;
  const int* past_len;    // [B]
  const float* inv_freq;  // [rot_dim/2] or null (no rotation)
  const int* position_ids;  // optional [B*q_len] explicit positions (else past + t)
  float attn_scaling;     // multiplies cos/sin (yarn / longrope), usually 1
  int B, Hq, Hkv, q_len, ld, Lmax, D, rot_dim;
  int interleaved;        // 0: neox "rotate half" pairs (i, i+rot/2); 1: gpt-j pairs (2i, 2i+1)
  int fp8;
};

// grid: (B*q_len, Hq + 2*Hkv); block: 64 threads (one per pair / per two channels)
__global__ __launch_bounds__(64) void rope_append_kernel(RopeAppendArgs a) {
  const int tokidx = blockIdx.x, head = blockIdx.y;
  const int b = tokidx / a.q_len, t = tokidx % a.q_len;
  const int pos_cache = a.past_len[b] + t;
  const int pos = a.position_ids ? a.position_ids[tokidx] : pos_cache;
  _Float16* row = a.qkv + (int64_t)tokidx * a.ld + (int64_t)head * a.D;
  const bool is_q = head < a.Hq, is_k = !is_q && head < a.Hq + a.Hkv;
  const int half_rot = a.rot_dim / 2;
  if (pos_cache >= a.Lmax) return;  // host guarantees capacity; never write out of bounds

  if (is_q || is_k) {
    const int hk = head - a.Hq;
    const int64_t kdst = (((int64_t)b * a.Hkv + hk) * a.Lmax + pos_cache) * a.D;
    for (int i = threadIdx.x; i < a.D / 2; i += 64) {
      // pair (i0, i1)
      int i0, i1;
      bool rot = a.inv_freq != nullptr && i < half_rot;
      if (i < half_rot) {
        if (a.interleaved) { i0 = 2 * i; i1 = 2 * i + 1; }
        else { i0 = i; i1 = i + half_rot; }
      } else {  // channels beyond the rotary dim: pass through, handled pairwise
        i0 = a.rot_dim + 2 * (i - half_rot);
        i1 = i0 + 1;
      }
      float x0 = (float)row[i0], x1 = (float)row[i1];
      if (rot) {
        float sn, cs;
        sincosf((float)pos * a.inv_freq[i], &sn, &cs);
        sn *= a.attn_scaling;
        cs *= a.attn_scaling;
        const float y0 = x0 * cs - x1 * sn, y1 = x1 * cs + x0 * sn;
        x0 = y0;
        x1 = y1;
      }
      if (is_q) {
        row[i0] = (_Float16)x0;
        row[i1] = (_Float16)x1;
      } else if (a.fp8) {
        ((uint8_t*)a.kc)[kdst + i0] = f16_to_e5m2((_Float16)x0);
        ((uint8_t*)a.kc)[kdst + i1] = f16_to_e5m2((_Float16)x1);
      } else {
        ((_Float16*)a.kc)[kdst + i0] = (_Float16)x0;
        ((_Float16*)a.kc)[kdst + i1] = (_Float16)x1;
      }
    }
  } else {
    const int hv = head - a.Hq - a.Hkv;
    const int64_t vbase = ((int64_t)b * a.Hkv + hv) * (int64_t)a.D * a.Lmax + pos_cache;
    for (int d = threadIdx.x; d < a.D; d += 64) {
      if (a.fp8) ((uint8_t*)a.vc)[vbase + (int64_t)d * a.Lmax] = f16_to_e5m2(row[d]);
      else ((_Float16*)a.vc)[vbase + (int64_t)d * a.Lmax] = row[d];
    }
  }
}

}  // namespace bigdl

using namespace bigdl;

static int g_attn_ablate = 0;
extern "C" void bigdl_attn_set_ablation(int v) { g_attn_ablate = v; }

// q tokens [q_off, q_off+q_cnt) of the q_total tokens appended this step; (Hq/Hkv)*q_cnt <= 32.
// rope != null: q is rotated in registers, k is rotated and k/v appended to the caches by this launch
// (requires q_off == 0 for the append; rot/2 % 32 == 0 unless interleaved).
extern "C" int bigdl_attn_decode(void* qkv, void* kc, void* vc, void* out, float* ws, int* counters,
                                 const int* past_len, const void* mask, int mask_ld, const float* alibi,
                                 const float* rope, const int* position_ids, int rope_max_pos, int rot, int interleaved, float scale,
                                 float softcap, int B, int Hq, int Hkv, int q_total, int q_off, int q_cnt, int ld,
                                 int Lmax, int D, int splits, int window, int fp8, int append, int defer_combine,
                                 void* k_scale, void* v_scale, hipStream_t stream) {
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || q_cnt <= 0 || q_off < 0 || q_off + q_cnt > q_total || splits <= 0 ||
      (Lmax & 255) || (ld & 7)) return -1;
  if (((uintptr_t)qkv | (uintptr_t)kc | (uintptr_t)vc | (uintptr_t)ws | (uintptr_t)out) & 15) return -1;
  const int G = Hq / Hkv;
  const int nchunk = (G + 31) / 32, gchunk = (G + nchunk - 1) / nchunk;
  if (gchunk * q_cnt > 32) return -4;
  if (nchunk > 1 && defer_combine) return -4;     // the o_proj-side combine indexes the workspace per kv head
  if (rope) {
    if (rot <= 0 || rot > D || (rot & 7) || rope_max_pos <= 0) return -1;
    if (!interleaved && rot != D) return -6;
  }
  AttnArgs a;
  a.qkv = (_Float16*)qkv; a.kc = kc; a.vc = vc; a.out = (_Float16*)out; a.ws = ws; a.counters = counters;
  a.past_len = past_len; a.mask = (const _Float16*)mask; a.mask_ld = mask_ld; a.scale = scale; a.softcap = softcap;
  a.rope = rope; a.position_ids = position_ids; a.rope_max_pos = rope_max_pos; a.alibi = alibi;
  a.rot = rope ? rot : (append ? 0 : -1); a.interleaved = interleaved;
  a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.ld = ld; a.Lmax = Lmax; a.splits = splits; a.window = window;
  a.q_total = q_total; a.q_off = q_off; a.q_cnt = q_cnt; a.defer_combine = defer_combine;
  a.nchunk = nchunk; a.gchunk = gchunk; a.ablate = g_attn_ablate;
  a.ks = (_Float16*)k_scale; a.vs = (_Float16*)v_scale;
  if (k_scale != nullptr || v_scale != nullptr) {
    // int4 caches: the kernel appends itself (quantization needs the whole rotated key), full rotary only
    if (k_scale == nullptr || v_scale == nullptr || fp8 || a.rot < 0 || (((uintptr_t)k_scale | (uintptr_t)v_scale) & 15)) return -1;
    switch (D) {
      case 64: return launch_rt<64, false, true>(a, stream);
      case 128: return launch_rt<128, false, true>(a, stream);
      case 256: return launch_rt<256, false, true>(a, stream);
      default: return -2;
    }
  }
#define DISPATCH_D(DD)                                                   \
  case DD: return fp8 ? launch_rt<DD, true>(a, stream) : launch_rt<DD, false>(a, stream);
  switch (D) {
    DISPATCH_D(64)
    DISPATCH_D(96)
    DISPATCH_D(128)
    DISPATCH_D(256)
    default: return -2;
  }
#undef DISPATCH_D
}

// workspace floats needed by bigdl_attn_decode
extern "C" int64_t bigdl_attn_decode_ws_floats(int B, int Hq, int Hkv, int q_len, int D, int splits) {
  const int G = Hq / Hkv;
  const int nchunk = (G + 31) / 32, gchunk = (G + nchunk - 1) / nchunk;
  const int rows = gchunk * q_len > 32 ? 32 : gchunk * q_len;
  const int rt = (rows + 15) / 16;
  return (int64_t)B * Hkv * nchunk * splits * rt * 16 * (D + 4);
}

// int32 arrival counters needed by bigdl_attn_decode (zero-initialized by the caller, left zero by every launch)
extern "C" int bigdl_attn_decode_counters(int B, int Hq, int Hkv) {
  return B * Hkv * ((Hq / Hkv + 31) / 32);
}

extern "C" int bigdl_rope_append(void* qkv, void* kc, void* vc, const int* past_len, const float* inv_freq,
                                 const int* position_ids, float attn_scaling, int B, int Hq, int Hkv, int q_len,
                                 int ld, int Lmax, int D, int rot_dim, int interleaved, int fp8, hipStream_t stream) {
  if (B <= 0 || q_len <= 0 || (D & 1) || rot_dim > D || (rot_dim & 1)) return -1;
  RopeAppendArgs a;
  a.qkv = (_Float16*)qkv; a.kc = kc; a.vc = vc; a.past_len = past_len; a.inv_freq = inv_freq;
  a.position_ids = position_ids; a.attn_scaling = attn_scaling; a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.q_len = q_len;
  a.ld = ld; a.Lmax = Lmax; a.D = D; a.rot_dim = inv_freq ? rot_dim : 0; a.interleaved = interleaved; a.fp8 = fp8;
  hipLaunchKernelGGL(rope_append_kernel, dim3(B * q_len, Hq + 2 * Hkv), dim3(64), 0, stream, a);
  return HIP_CHECK_LAUNCH();
}
