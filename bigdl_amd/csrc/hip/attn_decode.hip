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
// In both products the query row is the MFMA column (= lane & 15), so the online-softmax
// statistics and the O rescale are lane-local.  A 32-key step feeds the two 16-key S^T tiles
// with keys {8g+q} and {8g+4+q} so that lane group g ends up owning 8 consecutive keys, which
// is exactly the k-slice its P^T fragment must hold for the O^T product.
#include "common.h"

namespace bigdl {

struct AttnArgs {
  const _Float16* qkv;   // [B*q_len, ld]  q heads | k heads | v heads (q already rotated)
  const void* kc;        // K cache
  const void* vc;        // V^T cache
  _Float16* out;         // [B*q_len, Hq*D]
  float* ws;             // partials [B*Hkv][splits][16*RT rows][D + 4]  (O row, then m, l)
  int* counters;         // [B*Hkv] zero before first use; reset by the last arriver
  const int* past_len;   // [B] tokens in the cache before this step (device, graph-replayable)
  const _Float16* mask;  // optional additive mask [B, q_len, mask_ld] or null
  int mask_ld;
  float scale;           // softmax scale
  float softcap;         // gemma2 logit soft-capping (0 = off)
  int B, Hq, Hkv, q_len, ld, Lmax, splits;
  int window;            // sliding window (0 = off): keys < pos - window + 1 are masked
};

constexpr float NEG_BIG = -1.0e30f;

template <bool FP8>
__device__ __forceinline__ half8 load_kv8(const void* base, int64_t elem_off) {
  if constexpr (FP8) {
    const uint2v w = *(const uint2v*)((const uint8_t*)base + elem_off);
    return dec_e5m2(w.x, w.y);
  } else {
    return *(const half8*)((const _Float16*)base + elem_off);
  }
}

// D = head dim (multiple of 32), RT = number of 16-row tiles
template <int D, bool FP8, int RT>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs a) {
  constexpr int DC = D / 32;   // QK k-steps
  constexpr int DT = D / 16;   // O^T d-tiles
  constexpr int NW = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // merge buffers: per wave m,l per row, and O^T
  float* s_m = (float*)smem;                       // [NW][RT*16]
  float* s_l = s_m + NW * RT * 16;                 // [NW][RT*16]
  float* s_o = s_l + NW * RT * 16;                 // [NW][RT*16][D]
  int* s_flag = (int*)(s_o + NW * RT * 16 * D);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rr = lane & 15, g = lane >> 4;
  const int split = blockIdx.x, hkv = blockIdx.y, b = blockIdx.z;
  const int G = a.Hq / a.Hkv;
  const int rows = G * a.q_len;
  const int past = a.past_len[b];
  const int kv_len = past + a.q_len;
  // split ranges in units of 128 keys (4 waves x 32)
  const int nblk = (kv_len + 127) >> 7;
  const int per = (nblk + a.splits - 1) / a.splits;
  const int blk0 = split * per, blk1 = min(nblk, blk0 + per);

  const float qscale = a.softcap > 0.0f ? a.scale : a.scale * 1.4426950408889634f;

  // Q^T fragments (B operand): lane (row, g) holds d = 32c + 8g + j
  half8 qf[RT][DC];
  int lim[RT];  // last visible key position per row (causal within the new tokens)
  int lo_lim[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int row = 16 * t + rr;
    const int tok = row / G, gq = row % G;
    const bool valid = row < rows;
    lim[t] = valid ? past + tok : -1;
    lo_lim[t] = (a.window > 0 && valid) ? max(0, past + tok - a.window + 1) : 0;
    const _Float16* qp = a.qkv + (int64_t)(b * a.q_len + tok) * a.ld + (hkv * G + gq) * D;
#pragma unroll
    for (int c = 0; c < DC; ++c) {
      half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (valid) v = *(const half8*)(qp + 32 * c + 8 * g);
      qf[t][c] = v * (_Float16)qscale;
    }
  }

  const int64_t kbase = ((int64_t)b * a.Hkv + hkv) * (int64_t)a.Lmax * D;
  const char* kc = (const char*)a.kc + (FP8 ? kbase : kbase * 2);
  const char* vc = (const char*)a.vc + (FP8 ? kbase : kbase * 2);

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

  for (int blk = blk0; blk < blk1; ++blk) {
    const int kb = blk * 128 + wave * 32;
    if (kb >= kv_len) break;
    // ---- S^T = K . Q^T for two 16-key tiles ----
    // A-row i of tile u  <->  key kb + 8*(i>>2) + 4*u + (i&3)
    const int key0 = kb + 8 * (rr >> 2) + (rr & 3);
    half8 kf0[DC], kf1[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) {
      kf0[c] = load_kv8<FP8>(kc, (int64_t)key0 * D + 32 * c + 8 * g);
      kf1[c] = load_kv8<FP8>(kc, (int64_t)(key0 + 4) * D + 32 * c + 8 * g);
    }
    // V^T fragments: lane (d_local, g) holds keys kb + 8g .. +7 of channel 16*dt + d_local
    half8 vf[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) vf[d] = load_kv8<FP8>(vc, (int64_t)(16 * d + rr) * a.Lmax + kb + 8 * g);

#pragma unroll
    for (int t = 0; t < RT; ++t) {
      float4v s0 = zero4, s1 = zero4;
#pragma unroll
      for (int c = 0; c < DC; ++c) {
        s0 = mfma16(kf0[c], qf[t][c], s0);
        s1 = mfma16(kf1[c], qf[t][c], s1);
      }
      // lane (row rr, g): s0[q] <-> key kb+8g+q ; s1[q] <-> key kb+8g+4+q
      float sv[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { sv[q] = s0[q]; sv[4 + q] = s1[q]; }
      if (a.softcap > 0.0f) {
#pragma unroll
        for (int q = 0; q < 8; ++q) sv[q] = a.softcap * tanhf(sv[q] / a.softcap) * 1.4426950408889634f;
      }
      const int kfirst = kb + 8 * g;
      if (a.mask) {
        const int row = 16 * t + rr;
        if (row < rows) {
          const _Float16* mp = a.mask + ((int64_t)b * a.q_len + row / G) * a.mask_ld + kfirst;
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (kfirst + q < a.mask_ld) sv[q] += (float)mp[q] * 1.4426950408889634f;
        }
      }
      float mx = NEG_BIG;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int key = kfirst + q;
        if (key > lim[t] || key < lo_lim[t]) sv[q] = NEG_BIG;
        mx = fmaxf(mx, sv[q]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[t], mx);
      const float alpha = exp2f(m_run[t] - m_new);
      m_run[t] = m_new;
      H8 pf;
      float ps = 0.0f;
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        // masked entries: sv = NEG_BIG; when the whole row is still masked m_new = NEG_BIG too and
        // exp2(0) would be 1, so masked entries are forced to 0 explicitly
        const float p0 = sv[q] <= NEG_BIG ? 0.0f : exp2f(sv[q] - m_new);
        const float p1 = sv[q + 1] <= NEG_BIG ? 0.0f : exp2f(sv[q + 1] - m_new);
        ps += p0 + p1;
        pf.h2[q >> 1] = half2v{(_Float16)p0, (_Float16)p1};
      }
      l_run[t] = l_run[t] * alpha + ps;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int d = 0; d < DT; ++d) o[t][d] *= alpha;
      }
#pragma unroll
      for (int d = 0; d < DT; ++d) o[t][d] = mfma16(vf[d], pf.v, o[t][d]);
    }
  }

  // ---- merge the 4 wavefronts through LDS ----
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
        *(float4v*)&s_o[((wave * RT * 16) + 16 * t + rr) * D + 16 * d + 4 * g] = o[t][d];
    }
  }
  __syncthreads();

  const int bh = b * a.Hkv + hkv;
  float* wsp = a.ws + ((int64_t)bh * a.splits + split) * (RT * 16) * (D + 4);
  // thread -> (row, d-chunk): write merged partial for this split
  for (int idx = threadIdx.x; idx < rows * (D / 4); idx += 256) {
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
      acc += *(const float4v*)&s_o[(w * RT * 16 + row) * D + d4] * f;
    }
    float* dst = wsp + row * (D + 4);
    *(float4v*)(dst + d4) = acc;
    if (d4 == 0) { dst[D] = mm; dst[D + 1] = lt; }
  }

  // ---- cross-split combine by the last-arriving workgroup of this (b, hkv) ----
  if (a.splits > 1) {
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
  }

  const float* wbase = a.ws + (int64_t)bh * a.splits * (RT * 16) * (D + 4);
  for (int idx = threadIdx.x; idx < rows * (D / 4); idx += 256) {
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
    const int tok = row / G, gq = row % G;
    _Float16* dst = a.out + (int64_t)(b * a.q_len + tok) * (a.Hq * D) + (hkv * G + gq) * D + d4;
    half4 hv = {(_Float16)(acc[0] * inv), (_Float16)(acc[1] * inv), (_Float16)(acc[2] * inv), (_Float16)(acc[3] * inv)};
    *(half4*)dst = hv;
  }
}

template <int D, bool FP8>
static int launch_rt(const AttnArgs& a, hipStream_t st) {
  const int rows = (a.Hq / a.Hkv) * a.q_len;
  const int rt = (rows + 15) / 16;
  const dim3 grid(a.splits, a.Hkv, a.B), block(256);
  auto smem = [&](int RT) { return (size_t)(4 * RT * 16 * 2 + 4 * RT * 16 * D) * 4 + 16; };
  // more than 64 KiB of dynamic LDS has to be requested explicitly (once per kernel)
  static bool attr1 = false, attr2 = false;
  switch (rt) {
    case 1:
      if (!attr1 && smem(1) > 65536) {
        if (hipFuncSetAttribute((const void*)attn_decode_kernel<D, FP8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem(1)) != hipSuccess) return -5;
        attr1 = true;
      }
      hipLaunchKernelGGL((attn_decode_kernel<D, FP8, 1>), grid, block, smem(1), st, a);
      break;
    case 2:
      if (!attr2 && smem(2) > 65536) {
        if (hipFuncSetAttribute((const void*)attn_decode_kernel<D, FP8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem(2)) != hipSuccess) return -5;
        attr2 = true;
      }
      hipLaunchKernelGGL((attn_decode_kernel<D, FP8, 2>), grid, block, smem(2), st, a);
      break;
    default: return -4;
  }
  return HIP_CHECK_LAUNCH();
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

__device__ __forceinline__ uint8_t f16_to_e5m2(_Float16 h) {
  // round-to-nearest-even on the dropped 8 mantissa bits, saturate to the largest finite value
  union { _Float16 h; uint16_t u; } c;
  c.h = h;
  uint32_t u = c.u;
  const uint32_t sign = u & 0x8000u;
  uint32_t mag = u & 0x7fffu;
  if (mag > 0x7c00u) return (uint8_t)((sign >> 8) | 0x7f);   // NaN
  mag += 0x7fu + ((mag >> 8) & 1u);
  if (mag >= 0x7c00u) mag = 0x7b00u;
  return (uint8_t)((sign | mag) >> 8);
}

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

extern "C" int bigdl_attn_decode(const void* qkv, const void* kc, const void* vc, void* out, float* ws, int* counters,
                                 const int* past_len, const void* mask, int mask_ld, float scale, float softcap,
                                 int B, int Hq, int Hkv, int q_len, int ld, int Lmax, int D, int splits, int window,
                                 int fp8, hipStream_t stream) {
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || q_len <= 0 || splits <= 0 || (Lmax & 127) || (ld & 7)) return -1;
  if (((uintptr_t)qkv | (uintptr_t)kc | (uintptr_t)vc | (uintptr_t)ws) & 15) return -1;
  if ((Hq / Hkv) * q_len > 32) return -4;
  AttnArgs a;
  a.qkv = (const _Float16*)qkv; a.kc = kc; a.vc = vc; a.out = (_Float16*)out; a.ws = ws; a.counters = counters;
  a.past_len = past_len; a.mask = (const _Float16*)mask; a.mask_ld = mask_ld; a.scale = scale; a.softcap = softcap;
  a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.q_len = q_len; a.ld = ld; a.Lmax = Lmax; a.splits = splits; a.window = window;
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
  const int rows = (Hq / Hkv) * q_len;
  const int rt = (rows + 15) / 16;
  return (int64_t)B * Hkv * splits * rt * 16 * (D + 4);
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
