// Prefill / multi-token attention on gfx950: flash-style (online softmax, no S x L matrix), GQA-aware, causal or
// bidirectional, optional additive mask, sliding window and tanh logit soft-capping.
//
// Replaces the reference's closed xe_addons.sdp_causal / sdp_non_causal / gemma2_sdp_causal kernels
// (signatures python/llm/src/ipex_llm/transformers/xpu_ops.py:80-112, dispatcher
// python/llm/src/ipex_llm/transformers/models/common.py:222-278).
//
// Mapping (wave64, v_mfma_f32_16x16x32_f16):
//  * workgroup = 64 queries of one head (4 wavefronts x 16 queries), K/V visited in tiles of 64 keys staged in LDS
//    (K row-major, V transposed while it is staged).
//  * scores are computed transposed, S^T = K . Q^T: the MFMA result then holds, per lane, one query (column) and
//    16 keys, which is exactly the B-operand layout of the second product O^T += V^T . P^T.  P never leaves
//    registers, and every per-query quantity of the online softmax (running max, normaliser, rescale factor) is a
//    per-lane scalar shared by the 4 lanes of a query via two row swaps (v_permlane16_swap / v_permlane32_swap).
//  * the softmax is kept off the VALU as far as it goes: bare v_exp_f32, the scale folded into the exponent's
//    multiply-add, accumulators rescaled only when a tile's maximum exceeds the running one by more than 2^8, the
//    normaliser summed per lane and reduced once at the end, MFMA accumulators in VGPRs (per-file build flag) so that
//    the rare rescale needs no AGPR copies: 400 -> 140 VALU instructions per wavefront and key tile.
//  * tiles are double-buffered in LDS (one barrier per tile); whole tiles load without per-lane bounds.
//  * heads are placed per XCD (one L2 each); query blocks are scheduled heaviest first (causal: the last block sees the
//    most keys), alternating direction so that workgroups sharing a CU add up to similar work.
//  * measured on MI355X, Llama-2-7B heads, causal: 27 us at 1024 tokens, 276 us at 4096 (500 TFLOP/s; PyTorch SDPA:
//    60 / 420 us).  Counters at 4096: the SIMDs issue 55 % of the time (MFMA 20 %, VALU about 33 %), the rest is
//    waiting on LDS reads, dependent results and the per-tile barrier with two wavefronts per SIMD.  Variants measured
//    and not adopted: two query blocks per wavefront (halves the LDS reads per MFMA; 256 VGPRs: 36 / 308 us), a second
//    group of wavefronts taking alternate key tiles (51 / 588 us on the earlier kernel), tiles requested two phases
//    ahead through a second register set (29 / 295 us), the normaliser as an extra MFMA row block (no change).
#include "common.h"

namespace bigdl {

struct PrefillArgs {
  const _Float16* q; const _Float16* k; const _Float16* v; _Float16* out;
  const _Float16* mask;              // additive [B, (Hq,) Sq, Lk] (strides m_sb, m_sh, m_sq; last dim contiguous) or null
  const float* alibi;                // [Hq] ALiBi slopes (score += slope * (key - query position)) or null
  int B, Hq, Hkv, Sq, Lk;
  int64_t q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_ss, m_sb, m_sq, m_sh;
  float scale, softcap;
  int causal, window;
  int v_transposed;                  // v is [B, Hkv, D, Lk] (v_ss = stride of d, keys contiguous)
  int kv_e5m2;                       // k / v hold e5m2 bytes (strides in bytes); needs v_transposed
};

constexpr int PF_BN = 64;

// maximum over the 4 lanes that share a query (lanes r, r + 16, r + 32, r + 48): the two row swaps of gfx950 exchange
// register halves between lanes without a trip through the LDS crossbar
__device__ __forceinline__ float pf_max4(float x) {
  uint2v t = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(t.x), __uint_as_float(t.y));
  t = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(t.x), __uint_as_float(t.y));
}
__device__ __forceinline__ float pf_sum4(float x) {
  uint2v t = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(t.x) + __uint_as_float(t.y);
  t = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(t.x) + __uint_as_float(t.y);
}
// 2^x as the bare v_exp_f32 (exp2f adds a range fix-up for denormal results, which a softmax weight does not need)
__device__ __forceinline__ float pf_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// how far a tile's maximum may exceed the running one before the accumulators are rescaled: weights stay <= 2^8
constexpr float PF_LAZY = 8.0f;

// QB = 16-query blocks per wavefront: every K / V^T fragment read from LDS feeds QB MFMAs, which is what keeps the
// kernel off the LDS-bandwidth limit (QB = 2: 128 queries per workgroup).
// DR = the model's head_dim when it is not a multiple of 32 (72: SigLIP / MiniCPM-V vision towers, 80: phi-2): the
// tiles are D = 96 wide, loads beyond channel DR read as zero (they add nothing to q . k and their output columns
// are not stored), so such heads need neither a padded copy of q / k / v nor a fallback path.
// KV8: k and v are e5m2 bytes (the engine's fp8 KV cache, V transposed): a later chunk of a long prompt attends to the
// cached prefix where it lies, the bytes widen to fp16 on their way into LDS (an e5m2 code is the top byte of an
// fp16), instead of a dequantized fp16 copy of the whole prefix per layer and chunk.
template <int D, bool HAS_MASK, bool VT, int QB, int DR = D, bool KV8 = false>
__global__ __launch_bounds__(256) void attn_prefill_kernel(PrefillArgs a) {
  static_assert(!KV8 || VT, "byte caches store V transposed");
  static_assert(DR <= D && DR % 8 == 0, "head_dim must be a multiple of 8 within the tile width");
  constexpr int BM = 64 * QB;
  // LDS images, read as 16-byte MFMA fragments without bank conflicts:
  //  K   [64 keys][D]: rows of 2 D bytes, 16-byte chunk c of a key at chunk c ^ swz(key) (D = 96: 192-byte rows, padded
  //      by 16 bytes instead -- an XOR needs a power-of-two chunk count)
  //  V^T [D][64 keys]: rows of 128 bytes with the keys of a row reordered so that the 8 keys one lane feeds to an MFMA
  //      (32 t + 4 g + j and 32 t + 16 + 4 g + j, j < 4) are one chunk, chunk 4 t + g; chunks XOR-swizzled like K's
  constexpr bool KSWZ = D != 96;
  constexpr int KS = KSWZ ? D : D + 8;   // halves per K row in LDS
  constexpr int VS = PF_BN;              // halves per V^T row
  constexpr int DB = D / 16, STEPS = D / 32, CH = D / 8;
  // dynamic LDS: head_dim 256 needs 69 KiB (above the 64 KiB a kernel gets without asking)
  // two tile buffers up to head_dim 128 (one barrier per tile: the next tile is stored while slower wavefronts still
  // read the current one); head_dim 256 keeps one buffer of 64 KiB and two barriers
  constexpr int NBUF = D <= 128 ? 2 : 1;
  constexpr int TILE = PF_BN * KS + D * VS;  // halves per buffer: K [PF_BN][KS], then V^T [D][VS]
  extern __shared__ __attribute__((aligned(16))) _Float16 pf_smem[];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  // swizzle of a row: rows of 128 bytes alternate between the two halves of the 64 banks, so neighbours share a value
  auto swz128 = [](int row) { return (row >> 1) & 7; };
  auto kchunk = [&](int key, int c) {      // halves offset of chunk c of a key's row
    if constexpr (!KSWZ) return key * KS + 8 * c;
    else if constexpr (D == 64) return key * KS + 8 * (c ^ swz128(key));
    else return key * KS + 8 * (c ^ (key & 15));
  };
  auto vchunk = [&](int d, int c) { return d * VS + 8 * (c ^ swz128(d)); };
  // Workgroups are dispatched x-fastest and dealt to the 8 XCDs in turn, each with an L2 of its own.  When the number
  // of (batch, head) pairs divides by 8, XCD x takes a contiguous eighth of them (with grouped KV heads: whole groups),
  // one head after the other, so the K / V of about one head at a time is what that L2 has to hold; otherwise the grid
  // is used as it comes.  Within a head the causal work grows with the query block index: blocks are walked
  // heaviest-first, and lightest-first in every other run of 32 workgroups (an XCD's CUs take one workgroup each
  // before any takes a second) or every other head, so that workgroups sharing a CU add up to similar key counts.
  const int nqb = gridDim.x, nbh = gridDim.y * gridDim.z;
  const int lin = blockIdx.x + nqb * (blockIdx.y + gridDim.y * blockIdx.z);
  int bh = lin / nqb, qslot = lin % nqb;
  bool flip = 2 * bh >= nbh;
  if ((nbh & 7) == 0) {
    const int idx = lin >> 3;
    bh = (lin & 7) * (nbh >> 3) + idx / nqb;
    qslot = idx % nqb;
    flip = ((idx - qslot) / max(32, nqb)) & 1;     // by the head's first workgroup: one direction per head
  }
  const int hq = bh % (int)gridDim.y, b = bh / (int)gridDim.y;
  const int qb = flip ? qslot : nqb - 1 - qslot;
  const int hkv = hq / (a.Hq / a.Hkv);
  const int q0 = qb * BM;
  const int offset = a.Lk - a.Sq;                    // causal alignment: query i sits at key position i + offset
  int qi[QB], pos[QB];
#pragma unroll
  for (int u = 0; u < QB; ++u) {
    qi[u] = q0 + (wave * QB + u) * 16 + r;           // this lane's queries
    pos[u] = qi[u] + offset;
  }

  half8 qf[QB][STEPS];
#pragma unroll
  for (int u = 0; u < QB; ++u)
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      const bool in_head = DR == D || 32 * s + 8 * g < DR;
      qf[u][s] = qi[u] < a.Sq && in_head ? *(const half8*)(a.q + b * a.q_sb + hq * a.q_sh + (int64_t)qi[u] * a.q_ss + 32 * s + 8 * g) : z;
    }
  float4v o[QB][DB];
  float m[QB], l[QB];
#pragma unroll
  for (int u = 0; u < QB; ++u) {
    m[u] = -INFINITY;
    l[u] = 0.0f;
#pragma unroll
    for (int i = 0; i < DB; ++i) o[u][i] = float4v{0, 0, 0, 0};
  }
  const float LOG2E = 1.4426950408889634f;
  const float slope = a.alibi != nullptr ? a.alibi[hq] : 0.0f;

  int kbeg = 0, kend = a.Lk;
  if (a.causal) kend = min(a.Lk, q0 + BM + offset);
  if (a.window > 0) kbeg = max(0, q0 + offset - a.window + 1) / PF_BN * PF_BN;
  const _Float16* kbase = a.k + b * a.k_sb + hkv * a.k_sh;
  const _Float16* vbase = a.v + b * a.v_sb + hkv * a.v_sh;
  // (byte caches: the same strides count bytes)
  const uint8_t* kbase8 = (const uint8_t*)a.k + b * a.k_sb + hkv * a.k_sh;
  const uint8_t* vbase8 = (const uint8_t*)a.v + b * a.v_sb + hkv * a.v_sh;

  // global -> registers for one K/V tile (issued one tile ahead of its use)
  constexpr int NCH = PF_BN * CH / 256;
  half8 kreg[NCH], vreg[NCH];
  const _Float16* kfast = kbase + (int64_t)(threadIdx.x / CH) * a.k_ss + 8 * (threadIdx.x % CH);
  const _Float16* vfast = vbase + (int64_t)(threadIdx.x / (PF_BN / 8)) * a.v_ss + 8 * (threadIdx.x % (PF_BN / 8));
  auto load_tile = [&](int k0) {
    if constexpr (DR == D && !KV8 && VT && 256 % CH == 0) {
      if (k0 + PF_BN <= a.Lk) {          // whole tile inside the sequence (uniform): no per-lane bounds
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          // chunk c = tid + 256 i: key / row advance by a whole number per i, so each address is one per-thread
          // pointer plus a wavefront-uniform offset
          kreg[i] = *(const half8*)(kfast + (int64_t)(k0 + (256 / CH) * i) * a.k_ss);
          vreg[i] = *(const half8*)(vfast + (int64_t)(256 / (PF_BN / 8)) * i * a.v_ss + k0);
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = threadIdx.x + 256 * i;
      const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      if constexpr (KV8) {
        const int key = c / CH, dc = c % CH;
        const bool live = k0 + key < a.Lk && (DR == D || 8 * dc < DR);
        kreg[i] = z;
        if (live) {
          const uint2v w = *(const uint2v*)(kbase8 + (int64_t)(k0 + key) * a.k_ss + 8 * dc);
          kreg[i] = dec_e5m2(w.x, w.y);
        }
        const int d = c / (PF_BN / 8), kc = c % (PF_BN / 8);
        const uint8_t* vp = vbase8 + (int64_t)d * a.v_ss + k0 + 8 * kc;
        vreg[i] = z;
        if (DR == D || d < DR) {
          if (k0 + 8 * kc + 8 <= a.Lk) {
            const uint2v w = *(const uint2v*)vp;
            vreg[i] = dec_e5m2(w.x, w.y);
          } else {
            H8 t;
            t.v = z;
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (k0 + 8 * kc + e < a.Lk) t.u[e >> 1] |= (uint32_t)vp[e] << (16 * (e & 1) + 8);
            vreg[i] = t.v;
          }
        }
        continue;
      }
      {
        const int key = c / CH, dc = c % CH;
        const bool live = k0 + key < a.Lk && (DR == D || 8 * dc < DR);
        kreg[i] = live ? *(const half8*)(kbase + (int64_t)(k0 + key) * a.k_ss + 8 * dc) : z;
        if constexpr (!VT) vreg[i] = live ? *(const half8*)(vbase + (int64_t)(k0 + key) * a.v_ss + 8 * dc) : z;
      }
      if constexpr (VT) {
        const int d = c / (PF_BN / 8), kc = c % (PF_BN / 8);
        const _Float16* vp = vbase + (int64_t)d * a.v_ss + k0 + 8 * kc;
        if (DR != D && d >= DR) {
          vreg[i] = z;
        } else if (k0 + 8 * kc + 8 <= a.Lk) {
          vreg[i] = *(const half8*)vp;
        } else {
          half8 t = z;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (k0 + 8 * kc + e < a.Lk) t[e] = vp[e];
          vreg[i] = t;
        }
      }
    }
  };
  auto store_tile = [&](int buf) {
    _Float16* Ks = pf_smem + buf * TILE;
    _Float16* Vt = Ks + PF_BN * KS;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = threadIdx.x + 256 * i;
      const int key = c / CH, dc = c % CH;
      *(half8*)(Ks + kchunk(key, dc)) = kreg[i];
      if constexpr (VT) {
        // keys 8 kc .. 8 kc + 7 of row d: two groups of 4 that belong to neighbouring chunks (same half h of each)
        const int d = c / (PF_BN / 8), kc = c % (PF_BN / 8);
        const int ca = 4 * (kc >> 2) + 2 * (kc & 1), hh = (kc >> 1) & 1;
        const half8 v8 = vreg[i];
        *(half4*)(Vt + vchunk(d, ca) + 4 * hh) = half4{v8[0], v8[1], v8[2], v8[3]};
        *(half4*)(Vt + vchunk(d, ca + 1) + 4 * hh) = half4{v8[4], v8[5], v8[6], v8[7]};
      } else {
        const int ck = 4 * (key >> 5) + ((key >> 2) & 3), within = 4 * ((key >> 4) & 1) + (key & 3);
#pragma unroll
        for (int e = 0; e < 8; ++e) Vt[vchunk(8 * dc + e, ck) + within] = vreg[i][e];
      }
    }
  };

  if (kbeg < kend) {
    load_tile(kbeg);
    store_tile(0);
    if (kbeg + PF_BN < kend) load_tile(kbeg + PF_BN);
  }
  __syncthreads();
  const float sc2 = a.scale * LOG2E;
  const bool plain = !HAS_MASK && a.alibi == nullptr && a.softcap <= 0.0f && a.scale > 0.0f;
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += PF_BN) {
    const _Float16* Ks = pf_smem + cur * TILE;
    const _Float16* Vt = Ks + PF_BN * KS;
    // wavefronts whose queries all precede this tile have nothing to add (uniform per wavefront)
    const int wq_last = q0 + (wave * QB + QB) * 16 - 1 + offset;
    if (!(a.causal && k0 > wq_last)) {
      // ---- S^T = K . Q^T : lane -> query r of block u, keys 16*kb + 4g + reg ----
      float4v s[QB][4];
#pragma unroll
      for (int u = 0; u < QB; ++u)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) s[u][kb] = float4v{0, 0, 0, 0};
#pragma unroll
      for (int st = 0; st < STEPS; ++st)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const half8 kf = *(const half8*)(Ks + kchunk(16 * kb + r, 4 * st + g));
#pragma unroll
          for (int u = 0; u < QB; ++u) s[u][kb] = mfma16(kf, qf[u][st], s[u][kb]);
        }

      // ---- scale, soft-cap, mask, online softmax (all per-lane scalars per query block) ----
      half8 pf[QB][2];
#pragma unroll
      for (int u = 0; u < QB; ++u) {
        float mloc = -INFINITY;
        float mul;                     // weight = 2^(s * mul - running maximum)
        const int wq0 = q0 + (wave * QB + u) * 16 + offset;
        // interior tile: every key of the tile is visible to every query of this block (uniform branch)
        const bool interior = plain && k0 + PF_BN <= a.Lk && (!a.causal || k0 + PF_BN - 1 <= wq0) &&
                              (a.window <= 0 || k0 > wq0 + 15 - a.window);
        if (interior) {
          // the scale is folded into the exponent's multiply-add; the maximum is taken over the raw products
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) mloc = fmaxf(mloc, s[u][kb][e]);
          mloc *= sc2;
          mul = sc2;
        } else {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int key = k0 + 16 * kb + 4 * g + e;
              float v = s[u][kb][e] * a.scale;
              if (a.softcap > 0.0f) v = a.softcap * tanhf(v / a.softcap);
              if constexpr (HAS_MASK) {
                if (qi[u] < a.Sq && key < a.Lk) v += (float)a.mask[b * a.m_sb + hq * a.m_sh + (int64_t)qi[u] * a.m_sq + key];
              }
              v += slope * (float)(key - pos[u]);   // ALiBi: slope * (key - query position)
              bool ok = key < a.Lk;
              if (a.causal) ok = ok && key <= pos[u];
              if (a.window > 0) ok = ok && key > pos[u] - a.window;
              v = ok ? v * LOG2E : -INFINITY;
              s[u][kb][e] = v;
              mloc = fmaxf(mloc, v);
            }
          mul = 1.0f;
        }
        mloc = pf_max4(mloc);
        // The running maximum only follows a tile that exceeds it by more than 2^PF_LAZY: weights up to 2^8 are exact
        // in fp16 / fp32 all the same, the result (o / l) does not depend on the reference point, and most tiles skip
        // the pass over the accumulators.  The test is per wavefront so that the skip is a uniform branch.
        const float mnew = fmaxf(m[u], mloc);
        if (__builtin_amdgcn_ballot_w64(mnew > m[u] + PF_LAZY) != 0) {      // (-inf + 8 = -inf: any finite maximum)
          const float alpha = pf_exp2(m[u] - (mnew == -INFINITY ? 0.0f : mnew));   // m = -inf -> 0
          l[u] *= alpha;
#pragma unroll
          for (int i = 0; i < DB; ++i) o[u][i] *= alpha;
          m[u] = mnew;
        }
        const float mref = m[u] == -INFINITY ? 0.0f : m[u];
        float lsum = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p = pf_exp2(__builtin_fmaf(s[u][kb][e], mul, -mref));
            s[u][kb][e] = p;
            lsum += p;
          }
        l[u] += lsum;                  // this lane's keys only: the 4 lanes of a query are added up after the loop
        // P^T as B operand: k slot (g, j) = key 32t + 4g + j (j < 4) / 32t + 16 + 4g + j - 4
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pf[u][t][e] = (_Float16)s[u][2 * t][e];
            pf[u][t][4 + e] = (_Float16)s[u][2 * t + 1][e];
          }
      }

      // ---- O^T += V^T . P^T ----
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int d0 = 0; d0 < DB; d0 += 8) {         // fragments are fetched 8 at a time, then multiplied
          constexpr int NV = DB < 8 ? DB : 8;
          half8 vf[NV];
#pragma unroll
          for (int i = 0; i < NV; ++i)
            if (d0 + i < DB) vf[i] = *(const half8*)(Vt + vchunk(16 * (d0 + i) + r, 4 * t + g));
#pragma unroll
          for (int i = 0; i < NV; ++i)
            if (d0 + i < DB) {
#pragma unroll
              for (int u = 0; u < QB; ++u) o[u][d0 + i] = mfma16(vf[i], pf[u][t], o[u][d0 + i]);
            }
        }
    }
    if (k0 + PF_BN < kend) {
      if constexpr (NBUF == 1) __syncthreads();      // every wavefront is done with the only buffer
      store_tile(cur ^ (NBUF - 1));
      if (k0 + 2 * PF_BN < kend) load_tile(k0 + 2 * PF_BN);
    }
    __syncthreads();
    cur ^= NBUF - 1;
  }
#pragma unroll
  for (int u = 0; u < QB; ++u) l[u] = pf_sum4(l[u]);

#pragma unroll
  for (int u = 0; u < QB; ++u)
    if (qi[u] < a.Sq) {
      const float inv = l[u] > 0.0f ? 1.0f / l[u] : 0.0f;
      _Float16* op = a.out + b * a.o_sb + (int64_t)qi[u] * a.o_ss + hq * DR;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        if (DR != D && 16 * db + 4 * g >= DR) continue;
        half4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (_Float16)(o[u][db][e] * inv);
        *(half4*)(op + 16 * db + 4 * g) = h;
      }
    }
}

// ---------------------------------------------------------------------------------------------
// RoPE + KV-cache append for a whole prompt, one launch per layer:
//   q heads : rotated in place in the fused qkv buffer
//   k heads : rotated in place (the attention of this call reads them unquantized) and copied to the K cache
//             [B, Hkv, Lmax, D] as fp16 or e5m2
//   v heads : transposed through LDS and written to the V^T cache [B, Hkv, D, Lmax] (fp16 or e5m2) and, when the
//             cache is e5m2, also to an fp16 V^T scratch [B, Hkv, D, S_alloc] for this call's attention
// cos/sin come from a table [max_pos, rot/2, 2] (fp32, attention scaling folded in).
// ---------------------------------------------------------------------------------------------
struct PrefillRopeArgs {
  _Float16* qkv; void* kc; void* vc; _Float16* vt_scratch;
  const float* table; const int64_t* position_ids;     // [B, S] or null (past + t)
  int B, Hq, Hkv, S, D, past, Lmax, rot, interleaved, fp8;
  int64_t ld, vt_ld;                                   // qkv row stride; scratch row stride (keys per d row)
  int max_pos;
};

constexpr int PR_TOK = 64;

__global__ __launch_bounds__(256) void prefill_rope_append_kernel(PrefillRopeArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 tile[PR_TOK * (256 + 8)];   // V transpose staging, D <= 256
  const int t0 = blockIdx.x * PR_TOK, head = blockIdx.y, b = blockIdx.z;
  const int D = a.D, CH = D / 8;
  const bool is_q = head < a.Hq, is_k = !is_q && head < a.Hq + a.Hkv;
  _Float16* base = a.qkv + ((int64_t)b * a.S) * a.ld + (int64_t)head * D;
  if (is_q || is_k) {
    const int hk = head - a.Hq;
    const int half_rot = a.rot / 2;
    for (int it = threadIdx.x; it < PR_TOK * CH; it += 256) {
      const int tl = it / CH, c = it % CH, t = t0 + tl;
      if (t >= a.S) continue;
      const int d0 = 8 * c;
      int pos = a.past + t;
      if (a.position_ids) pos = (int)a.position_ids[(int64_t)b * a.S + t];
      pos = min(max(pos, 0), a.max_pos - 1);
      _Float16* row = base + (int64_t)t * a.ld;
      const int64_t kdst = (((int64_t)b * a.Hkv + hk) * a.Lmax + a.past + t) * D;
      auto put_k = [&](int d, half8 v) {
        if (!is_k || a.kc == nullptr) return;
        if (a.fp8) {
          uint2v pk;
          uint32_t w[2] = {0, 0};
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e >> 2] |= (uint32_t)f16_to_e5m2(v[e]) << (8 * (e & 3));
          pk.x = w[0]; pk.y = w[1];
          *(uint2v*)((uint8_t*)a.kc + kdst + d) = pk;
        } else {
          *(half8*)((_Float16*)a.kc + kdst + d) = v;
        }
      };
      if (d0 >= a.rot) {                                     // beyond the rotary dim: pass through
        put_k(d0, *(const half8*)(row + d0));
      } else if (a.interleaved) {                            // adjacent pairs inside the chunk
        half8 x = *(const half8*)(row + d0), y;
        const float* cs = a.table + ((int64_t)pos * half_rot + d0 / 2) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float c0 = cs[2 * e], s0 = cs[2 * e + 1];
          const float x0 = (float)x[2 * e], x1 = (float)x[2 * e + 1];
          y[2 * e] = (_Float16)(x0 * c0 - x1 * s0);
          y[2 * e + 1] = (_Float16)(x1 * c0 + x0 * s0);
        }
        *(half8*)(row + d0) = y;
        put_k(d0, y);
      } else if (d0 < half_rot) {                            // half layout: pair (i, i + rot/2)
        half8 x = *(const half8*)(row + d0), z = *(const half8*)(row + d0 + half_rot), y0, y1;
        const float* cs = a.table + ((int64_t)pos * half_rot + d0) * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float c0 = cs[2 * e], s0 = cs[2 * e + 1];
          const float x0 = (float)x[e], x1 = (float)z[e];
          y0[e] = (_Float16)(x0 * c0 - x1 * s0);
          y1[e] = (_Float16)(x1 * c0 + x0 * s0);
        }
        *(half8*)(row + d0) = y0;
        *(half8*)(row + d0 + half_rot) = y1;
        put_k(d0, y0);
        put_k(d0 + half_rot, y1);
      }                                                      // chunks in [rot/2, rot) are written by their partner
    }
    return;
  }
  // ---- V: [tokens, D] -> V^T rows ----
  const int hv = head - a.Hq - a.Hkv;
  const int TS = D + 8;
  // the transposed read below walks 8 token rows per thread with the token group (tc) varying fastest across lanes:
  // rows 8 apart are a multiple of 32 banks apart, so the 16-byte chunk index is XOR-ed with the token group to
  // spread the 8 groups over all banks (needs the chunk count to be a multiple of 8)
  const int swz = (CH & 7) == 0 ? 7 : 0;
  for (int it = threadIdx.x; it < PR_TOK * CH; it += 256) {
    const int tl = it / CH, c = it % CH, t = t0 + tl;
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t < a.S) v = *(const half8*)(base + (int64_t)t * a.ld + 8 * c);
    *(half8*)(tile + tl * TS + 8 * (c ^ ((tl >> 3) & swz))) = v;
  }
  __syncthreads();
  const int ntok = min(PR_TOK, a.S - t0);
  // each item: one d row, 8 consecutive tokens
  for (int it = threadIdx.x; it < D * (PR_TOK / 8); it += 256) {
    const int d = it / (PR_TOK / 8), tc = it % (PR_TOK / 8);
    if (8 * tc >= ntok) continue;
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[(8 * tc + e) * TS + (((d >> 3) ^ (tc & swz)) << 3) + (d & 7)];
    const int64_t row = ((int64_t)b * a.Hkv + hv) * D + d;
    const int key = a.past + t0 + 8 * tc;
    const bool full = 8 * tc + 8 <= ntok;
    if (a.vt_scratch) {
      _Float16* sd = a.vt_scratch + row * a.vt_ld + t0 + 8 * tc;
      if (full && (((uintptr_t)sd) & 15) == 0) *(half8*)sd = v;
      else for (int e = 0; e < 8 && 8 * tc + e < ntok; ++e) sd[e] = v[e];
    }
    if (a.vc == nullptr) continue;
    if (a.fp8) {
      uint8_t* dst = (uint8_t*)a.vc + row * a.Lmax + key;
      if (full && (((uintptr_t)dst) & 7) == 0) {
        uint32_t w[2] = {0, 0};
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e >> 2] |= (uint32_t)f16_to_e5m2(v[e]) << (8 * (e & 3));
        uint2v pk; pk.x = w[0]; pk.y = w[1];
        *(uint2v*)dst = pk;
      } else {
        for (int e = 0; e < 8 && 8 * tc + e < ntok; ++e) dst[e] = f16_to_e5m2(v[e]);
      }
    } else {
      _Float16* dst = (_Float16*)a.vc + row * a.Lmax + key;
      if (full && (((uintptr_t)dst) & 15) == 0) *(half8*)dst = v;
      else for (int e = 0; e < 8 && 8 * tc + e < ntok; ++e) dst[e] = v[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Prompt keys / values -> int4 caches (format: attn_decode.hip).  One thread per (token, kv head, 32-channel group):
// consecutive threads take consecutive tokens, so the byte stores into the V^T rows are contiguous across a wavefront.
// ---------------------------------------------------------------------------------------------
struct KvInt4Args {
  const _Float16* qkv; uint8_t* kc; uint8_t* vc; _Float16* ks; _Float16* vs;
  int B, Hq, Hkv, S, D, past, Lmax;
  int64_t ld;
};

__device__ __forceinline__ uint32_t pf_q_i4(float x, float inv) {
  return (uint32_t)(int)fminf(fmaxf(rintf(x * inv) + 8.0f, 0.0f), 15.0f);
}

__global__ __launch_bounds__(256) void kv_int4_append_kernel(KvInt4Args a) {
  const int NG = a.D / 32;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int t = (int)(idx % a.S);
  const int m = (int)((idx / a.S) % NG);
  const int hk = blockIdx.y, b = blockIdx.z;
  if (idx >= (int64_t)a.S * NG) return;
  const _Float16* krow = a.qkv + ((int64_t)b * a.S + t) * a.ld + (int64_t)(a.Hq + hk) * a.D + 32 * m;
  const _Float16* vrow = krow + (int64_t)a.Hkv * a.D;
  const int64_t bh = (int64_t)b * a.Hkv + hk;
  const int p = a.past + t;
  half8 x[4];
  float am = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[i] = *(const half8*)(krow + 8 * i);
#pragma unroll
    for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf((float)x[i][j]));
  }
  _Float16 sc = (_Float16)(am * (1.0f / 7.0f));
  float inv = (float)sc > 0.0f ? 1.0f / (float)sc : 0.0f;
  uint4v kw;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) w |= pf_q_i4((float)x[i][j], inv) << (4 * ((j >> 1) + 4 * (j & 1)));
    kw[i] = w;
  }
  *(uint4v*)(a.kc + (bh * a.Lmax + p) * (a.D / 2) + 16 * m) = kw;
  a.ks[(bh * a.Lmax + p) * NG + m] = sc;
  am = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[i] = *(const half8*)(vrow + 8 * i);
#pragma unroll
    for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf((float)x[i][j]));
  }
  sc = (_Float16)(am * (1.0f / 7.0f));
  inv = (float)sc > 0.0f ? 1.0f / (float)sc : 0.0f;
  uint8_t* vdst = a.vc + (bh * (a.D / 2) + 16 * m) * a.Lmax + p;
#pragma unroll
  for (int r = 0; r < 16; ++r)      // channels 32m + r (low nibble) and 32m + 16 + r (high nibble)
    vdst[(int64_t)r * a.Lmax] = (uint8_t)(pf_q_i4((float)x[r >> 3][r & 7], inv) | (pf_q_i4((float)x[2 + (r >> 3)][r & 7], inv) << 4));
  a.vs[(bh * NG + m) * a.Lmax + p] = sc;
}

}  // namespace bigdl

using namespace bigdl;

extern "C" int bigdl_kv_int4_append(const void* qkv, void* kc, void* vc, void* ks, void* vs, int B, int Hq, int Hkv, int S,
                                    int D, int past, int Lmax, int64_t ld, hipStream_t st) {
  if (B <= 0 || S <= 0 || Hkv <= 0 || (D != 64 && D != 128 && D != 256) || past < 0 || past + S > Lmax || (ld & 7)) return -1;
  if (((uintptr_t)qkv | (uintptr_t)kc | (uintptr_t)vc | (uintptr_t)ks | (uintptr_t)vs) & 15) return -1;
  if (kc == nullptr || vc == nullptr || ks == nullptr || vs == nullptr || B > 65535 || Hkv > 65535) return -1;
  KvInt4Args a;
  a.qkv = (const _Float16*)qkv; a.kc = (uint8_t*)kc; a.vc = (uint8_t*)vc; a.ks = (_Float16*)ks; a.vs = (_Float16*)vs;
  a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.S = S; a.D = D; a.past = past; a.Lmax = Lmax; a.ld = ld;
  const int64_t items = (int64_t)S * (D / 32);
  hipLaunchKernelGGL(kv_int4_append_kernel, dim3((unsigned)((items + 255) / 256), Hkv, B), dim3(256), 0, st, a);
  return HIP_CHECK_LAUNCH();
}

// qkv [B, S, ld] fp16 (heads: Hq q, Hkv k, Hkv v; head h at column h*D).  past + S <= Lmax (validated here).
extern "C" int bigdl_prefill_rope_append(void* qkv, void* kc, void* vc, void* vt_scratch, const float* table,
                                         const int64_t* position_ids, int B, int Hq, int Hkv, int S, int D, int past,
                                         int Lmax, int rot, int interleaved, int fp8, int64_t ld, int64_t vt_ld,
                                         int max_pos, hipStream_t st) {
  if (B <= 0 || S <= 0 || D <= 0 || (D & 7) || D > 256 || rot < 0 || rot > D || (rot & 15) || past < 0 || past + S > Lmax) return -1;
  if ((ld & 7) || ((uintptr_t)qkv & 15) || ((uintptr_t)kc & 15) || ((uintptr_t)vc & 15) || table == nullptr || max_pos <= 0) return -1;
  if (B > 65535 || Hq + 2 * Hkv > 65535) return -1;
  PrefillRopeArgs a;
  a.qkv = (_Float16*)qkv; a.kc = kc; a.vc = vc; a.vt_scratch = (_Float16*)vt_scratch; a.table = table;
  a.position_ids = position_ids; a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.S = S; a.D = D; a.past = past; a.Lmax = Lmax;
  a.rot = rot; a.interleaved = interleaved; a.fp8 = fp8; a.ld = ld; a.vt_ld = vt_ld; a.max_pos = max_pos;
  hipLaunchKernelGGL(prefill_rope_append_kernel, dim3((S + PR_TOK - 1) / PR_TOK, Hq + 2 * Hkv, B), dim3(256), 0, st, a);
  return HIP_CHECK_LAUNCH();
}

template <int D, bool HAS_MASK, bool VT, int QB, int DR = D, bool KV8 = false>
static int launch_pf_one(const PrefillArgs& a, hipStream_t st) {
  const dim3 grid((a.Sq + 64 * QB - 1) / (64 * QB), a.Hq, a.B), block(256);
  constexpr size_t lds = ((size_t)PF_BN * (D != 96 ? D : D + 8) + (size_t)D * PF_BN) * 2 * (D <= 128 ? 2 : 1);
  static bool attr = false;
  if (!attr && lds > 65536) {
    if (hipFuncSetAttribute((const void*)attn_prefill_kernel<D, HAS_MASK, VT, QB, DR, KV8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return -5;
    attr = true;
  }
  hipLaunchKernelGGL((attn_prefill_kernel<D, HAS_MASK, VT, QB, DR, KV8>), grid, block, lds, st, a);
  return HIP_CHECK_LAUNCH();
}

template <int D, int QB, int DR = D>
static int launch_pf_qb(const PrefillArgs& a, hipStream_t st) {
  if (a.kv_e5m2) {
    if constexpr (DR == D) {
      if (!a.v_transposed) return -6;
      return a.mask ? launch_pf_one<D, true, true, QB, DR, true>(a, st) : launch_pf_one<D, false, true, QB, DR, true>(a, st);
    } else {
      return -6;
    }
  }
  if (a.mask) return a.v_transposed ? launch_pf_one<D, true, true, QB, DR>(a, st) : launch_pf_one<D, true, false, QB, DR>(a, st);
  return a.v_transposed ? launch_pf_one<D, false, true, QB, DR>(a, st) : launch_pf_one<D, false, false, QB, DR>(a, st);
}

template <int D, int DR = D>
static int launch_pf(const PrefillArgs& a, hipStream_t st) {
  // QB = 2 (128 queries per workgroup) halves the LDS traffic per MFMA but needs 240 VGPRs (one wavefront per
  // SIMD); measured slower than QB = 1 from 1K to 8K tokens on MI355X, so QB = 1 is used throughout.
  return launch_pf_qb<D, 1, DR>(a, st);
}

// q [B, Hq, Sq, D], k / v [B, Hkv, Lk, D] as strided fp16 views (element strides, last dim contiguous);
// with v_transposed, v is [B, Hkv, D, Lk] (v_ss = stride between d rows; the layout of the engine's V^T cache,
// which saves the in-kernel transpose).  out [B, Sq, Hq * D] (o_sb, o_ss; heads contiguous).
// Query i attends to keys <= i + (Lk - Sq) when causal.
extern "C" int bigdl_attn_prefill(const void* q, const void* k, const void* v, void* out, const void* mask, int B, int Hq,
                                  int Hkv, int Sq, int Lk, int D, int64_t q_sb, int64_t q_sh, int64_t q_ss, int64_t k_sb,
                                  int64_t k_sh, int64_t k_ss, int64_t v_sb, int64_t v_sh, int64_t v_ss, int64_t o_sb,
                                  int64_t o_ss, int64_t m_sb, int64_t m_sq, int64_t m_sh, const float* alibi, float scale,
                                  float softcap, int causal, int window, int v_transposed, int kv_e5m2, hipStream_t st) {
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || Sq <= 0 || Lk <= 0 || B > 65535 || Hq > 65535) return -1;
  if (causal && Lk < Sq) return -1;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)out & 7)) return -1;
  if ((q_sb | q_sh | q_ss | k_sb | k_sh | k_ss | v_sb | v_sh | v_ss) & 7) return -1;    // 16-byte (8-byte: e5m2) row starts
  if ((o_sb | o_ss) & 3) return -1;
  PrefillArgs a;
  a.q = (const _Float16*)q; a.k = (const _Float16*)k; a.v = (const _Float16*)v; a.out = (_Float16*)out;
  a.mask = (const _Float16*)mask; a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.Sq = Sq; a.Lk = Lk;
  a.q_sb = q_sb; a.q_sh = q_sh; a.q_ss = q_ss; a.k_sb = k_sb; a.k_sh = k_sh; a.k_ss = k_ss;
  a.v_sb = v_sb; a.v_sh = v_sh; a.v_ss = v_ss; a.o_sb = o_sb; a.o_ss = o_ss; a.m_sb = m_sb; a.m_sq = m_sq; a.m_sh = m_sh; a.alibi = alibi;
  a.scale = scale; a.softcap = softcap; a.causal = causal; a.window = window; a.v_transposed = v_transposed; a.kv_e5m2 = kv_e5m2;
  switch (D) {
    case 64: return launch_pf<64>(a, st);
    case 72: return launch_pf<96, 72>(a, st);
    case 80: return launch_pf<96, 80>(a, st);
    case 96: return launch_pf<96>(a, st);
    case 128: return launch_pf<128>(a, st);
    case 256: return launch_pf<256>(a, st);
    default: return -2;
  }
}
