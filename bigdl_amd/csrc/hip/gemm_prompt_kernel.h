// Dequant-fused MFMA GEMM for prompt-sized row counts (M > 256) on gfx950:  y[M,N] = f(x[M,K] . dequant(W)[N,K]^T).
//
// The prompt-side counterpart of the reference's closed xe_linear.forward_new
// (python/llm/src/ipex_llm/transformers/low_bit_linear.py:657-668): the weights stay in their low-bit T16 tiles in HBM
// and are expanded on the way to the matrix cores, so a prompt pass neither writes nor re-reads an fp16 copy of the
// model and calls no library GEMM.
//
// Structure (one workgroup = WM x WN wavefronts = one BM x BN output tile, K walked in 64-deep tiles):
//  * x tile [BM][64 k] goes global -> LDS directly (global_load_lds, 16 bytes per lane), one tile ahead, two buffers.
//    The LDS image is lane-linear as the instruction requires; the bank swizzle (16-byte slot ^= row & 7) is applied
//    to the per-lane SOURCE address and again on the fragment reads.
//  * the weight tile [BN][64 k] is expanded ONCE per workgroup: every wavefront owns a share of the block's 16-row weight
//    tile rows, loads their raw T16 tiles (codes + block scales, 16-byte loads, one whole 128-deep tile per two k tiles)
//    two tiles ahead, expands them with the register bit tricks of the skinny kernels, folds the block scale in as a
//    packed fp16 multiply and writes the fp16 fragments to the other LDS buffer with one 16-byte store per unit.
//    (The earlier form expanded weights per wavefront in registers: every weight was decoded by both row halves of the
//    workgroup and the decode stood in the MFMA's issue slots.)
//  * fragments: both operands are read with ds_read_b128 from 128-byte rows; slot ^= row & 7 makes the four 16-lane
//    groups of the read and the eight 8-lane groups of the expansion's ds_write_b128 conflict free.
//  * 8-wavefront tiles run their two row halves as a ping-pong pair, one segment apart: while one wavefront of a SIMD
//    issues MFMAs its partner reads fragments, expands weights and requests the next tiles (comments at the loop).
//  * the weight fragment is the MFMA's A operand, so a lane holds 4 consecutive output columns of one row; the epilogue
//    stages the wavefront's tile through LDS and stores whole 128-byte row segments; the gate / up pair form combines the
//    two products of a column there (act(g) * u), so neither reaches memory.
//  * the block index is remapped so that the row blocks that share a weight column panel run on one XCD (one HBM read
//    of the panel per L2).
// Measured and dropped: stream-K over 256 x 256 tiles (every CU the same number of k tiles whatever the tile count; the
// 256 KiB fp32 slab per shared tile costs more than the idle CUs it removes: 130 instead of 98 us on the qkv shape),
// x requests or the expansion moved into the MFMA segment (no change / slower), wavefront priorities (no change).
#pragma once
#include "gemm_kernel.h"

// PROMPT_ABL (measurement builds, results wrong): 1 = no weight loads in the loop, 2 = no x tile loads in the loop,
// 4 = no expansion (codes written as they are), 8 = no barrier in the loop, 16 = no MFMA, 32 = no LDS fragment reads
#ifndef PROMPT_ABL
#define PROMPT_ABL 0
#endif
// phases per 64-deep tile of the 256 x 256 / 256 x 192 configurations (1 or 2)
#ifndef PROMPT_PH0
#define PROMPT_PH0 2
#endif
#ifndef PROMPT_PH1
#define PROMPT_PH1 1
#endif

namespace bigdl {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_cvoid_t;

// LDS bytes of the tile buffers: two x + two weight buffers, or the epilogue's staging area (rows padded by 16 bytes)
constexpr int prompt_lds_tiles(int BM, int BN, int NW, int MTW, int NTW) {
  const int a = 2 * (BM + BN) * 128, b = NW * MTW * 16 * (32 * NTW + 16);
  return a > b ? a : b;
}

// s_waitcnt immediates (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8])
constexpr int waitcnt_vm(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }
constexpr int waitcnt_vm_lgkm0(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4); }

// one expansion unit = the 8 codes per lane of one (16-row weight tile, 32-deep k step) -> the fp16 fragment with the
// block scale (and minimum) folded in.  `s` is the step inside the T16 tile (0 .. 3), static.
template <int QT, int S>
__device__ __forceinline__ half8 unit_expand(const Tile<QT>& t, const _Float16* lut) {
  const half8 code = decode_step<QT>(t, S, lut);
  _Float16 d = t.d[S];
  if constexpr (post_scale<QT>() != 1.0f) d = (_Float16)fminf((float)d * post_scale<QT>(), 65504.0f);
  if constexpr (has_min<QT>()) return code * d + t.m[S];
  return code * d;
}

// WM x WN wavefronts; a wavefront owns MTW x NTW 16 x 16 output tiles
template <int QT, int WM, int WN, int MTW, int NTW, int PH>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_prompt_kernel(GemmArgs a) {
  constexpr int NW = WM * WN;
  constexpr int BM = WM * MTW * 16, BN = WN * NTW * 16;
  constexpr int NA = BM / (8 * NW);                 // global_load_lds per lane and x tile
  constexpr bool PP = NW == 8;                      // two wavefronts per SIMD: the row halves run as a ping-pong pair
  // weight tile rows (16 rows each) of the block over the wavefronts: FULL whole rows per wavefront, and with 1.5 rows per
  // wavefront (BN = 192 on 8 wavefronts) a pair of wavefronts shares a third row, one k half of the T16 tile each
  constexpr int RT = BN / 16;
  constexpr int FULL = RT / NW, HALF = (RT % NW) ? 1 : 0;
  static_assert(BM % (8 * NW) == 0 && (RT % NW == 0 || (2 * RT == 3 * NW && NW % 2 == 0)), "tile does not divide over the wavefronts");
  constexpr int SLOTS = FULL + HALF, NUW = 2 * SLOTS;  // expansion units per wavefront and 64-deep tile (at most)
  constexpr int ABYTES = BM * 128, BBYTES = BN * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const As = smem;                            // [2][BM][128 B]
  char* const Bs = smem + 2 * ABYTES;               // [2][BN][128 B]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 15, g = lane >> 4;
  const int KT = a.K >> 7;                          // host: K % 128 == 0
  // pair form (a.w2: gate / up): the block's first RT / 2 weight tile rows come from w, the others from the same rows of
  // w2, and the epilogue writes act(first) * second: BN / 2 output columns per block
  const bool pair = a.w2 != nullptr;
  const int BNO = pair ? BN / 2 : BN;
  const int nmb = (a.M + BM - 1) / BM, nnb = (a.N + BNO - 1) / BNO;

  // XCD-aware tile order: blocks b and b + 8 share an L2; consecutive tile ids (row blocks of one column panel first)
  // go to one XCD (bijective also when the grid is not a multiple of 8)
  int tid;
  {
    const int nblk = gridDim.x, b = blockIdx.x, q = nblk >> 3, rr = nblk & 7, xcd = b & 7;
    tid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
  }
  const int mb = tid % nmb, nb = (tid / nmb) % nnb, sp = tid / (nmb * nnb);
  const int row0 = mb * BM, rows = min(BM, a.M - row0);
  const int kt0 = sp * a.kt_per_split, kt1 = min(KT, kt0 + a.kt_per_split);
  const int nkt = kt1 - kt0, T = 2 * nkt;            // T16 k tiles / 64-deep tiles of this workgroup
  const T16Planes pl = t16_planes(a.w, QT, a.N, a.K);
  const T16Planes pl2 = t16_planes(pair ? a.w2 : a.w, QT, a.N, a.K);

  const _Float16* lut = nullptr;
  if constexpr (QT == Q_NF4) {
    uint32_t* table = (uint32_t*)(smem + prompt_lds_tiles(BM, BN, NW, MTW, NTW));
    fill_nf4_banked(table, threadIdx.x, 64 * NW);
    lut = (const _Float16*)(table + (lane & 31));
    __syncthreads();       // every wavefront fills a share of the table and the first expansion reads all of it
  }

  // ---- x tile: every wavefront stages rows of its own row half (the half it reads fragments from, so that the two
  // halves of the workgroup may run out of step); instruction p fills rows 8 (p WN + wn) .. + 7 of the half; lane i
  // brings the 16 bytes that belong in slot i % 8 of row i / 8, i.e. k slot (i % 8) ^ (i / 8) of that row (rows past the
  // block re-read its last row: never stored)
  uint32_t xoff[NA];
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    const int row = wm * (BM / WM) + (p * WN + wn) * 8 + (lane >> 3);
    xoff[p] = (uint32_t)((int64_t)(row0 + min(row, rows - 1)) * a.ldx) + (uint32_t)((((lane & 7) ^ (lane >> 3)) & 7) << 3);
  }
  auto stage_a = [&](int t, int buf) {
    const _Float16* src = a.x + (int64_t)(2 * kt0 + t) * 64;
    char* dst = As + buf * ABYTES + (wm * (BM / WM) + wn * 8) * 128;     // wave-uniform; the instruction adds lane * 16
#pragma unroll
    for (int p = 0; p < NA; ++p)
      __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(src + xoff[p]), (lds_void_t*)(dst + p * WN * 1024), 16, 0, 0);
  };

  // ---- weight tile rows of this wavefront (block-relative index; clamped in the buffer: a row past N is loaded from the
  // last valid one and never stored) and their raw data: one whole T16 tile (128 k = two 64-deep tiles) per slot, loaded
  // once per pair of tiles with full 16-byte code loads
  const int ntiles = (a.N + 15) >> 4;
  static_assert(RT % 2 == 0, "the pair form splits the block's tile rows in two");
  int slot_rt[SLOTS], slot_second[SLOTS];
  int64_t slot_tile[SLOTS];
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    if constexpr (HALF) slot_rt[i] = (wave >> 1) * 3 + (i < FULL ? (wave & 1) : 2);
    else slot_rt[i] = wave * FULL + i;
    const int second = pair && slot_rt[i] >= RT / 2;
    const int trow = pair ? nb * (RT / 2) + slot_rt[i] - second * (RT / 2) : nb * RT + slot_rt[i];
    slot_tile[i] = (int64_t)min(trow, ntiles - 1) * KT + kt0;
    slot_second[i] = second;
  }
  const int half_sel = wave & 1;                     // which k half of the shared row this wavefront expands
  Tile<QT> rawC[SLOTS], rawN[SLOTS];                 // T16 tile being expanded / the next one (in flight)
  auto fetch_raw = [&](Tile<QT> (&raw)[SLOTS], int kt) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) load_tile<QT>(raw[i], slot_second[i] ? pl2 : pl, slot_tile[i] + kt, lane);
  };

  // fragment slot of k step s for this lane: (4 s + g) ^ (r & 7); s = 1 flips bit 2
  const int fsw0 = ((g ^ (r & 7)) & 7) << 4;
  const int frag_off[2] = {r * 128 + fsw0, r * 128 + (fsw0 ^ 64)};

  // expansion of units [u0, u1) of the 64-deep tile with parity h (= its LDS buffer) from the current raw set: unit u is
  // k step u % 2 of slot u / 2; the shared row's units exist only in the wavefront that owns that k half
  auto expand = [&](auto hc, auto u0c, auto u1c) {
    constexpr int h = decltype(hc)::value, u0 = decltype(u0c)::value, u1 = decltype(u1c)::value;
    char* base = Bs + h * BBYTES;
#pragma unroll
    for (int u = u0; u < u1; ++u) {
      const int i = u >> 1;
      if (i >= FULL && half_sel != h) continue;        // wave-uniform
      half8 v0, v1;
      // (both steps of the parity are instantiated, the static one is kept: decode_step wants a constant step)
      if (u & 1) v0 = unit_expand<QT, 2 * h + 1>(rawC[i], lut);
      else v0 = unit_expand<QT, 2 * h>(rawC[i], lut);
      if constexpr (PROMPT_ABL & 4) v0 = __builtin_bit_cast(half8, uint4v{rawC[i].lo0[0], rawC[i].lo0[1], 0x3c003c00u, 0x3c003c00u});
      (void)v1;
      *(half8*)(base + slot_rt[i] * 2048 + frag_off[u & 1]) = v0;
    }
  };

  float4v acc[MTW][NTW];
#pragma unroll
  for (int j = 0; j < MTW; ++j)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[j][n] = float4v{0, 0, 0, 0};

  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  auto lds_w = [&](int buf, int s) { return Bs + buf * BBYTES + wn * (NTW * 2048) + frag_off[s]; };
  auto lds_x = [&](int buf, int s) { return As + buf * ABYTES + wm * (MTW * 2048) + frag_off[s]; };
  auto frag = [&](const char* p, int i) {
    if constexpr (PROMPT_ABL & 32) return __builtin_bit_cast(half8, uint4v{(uint32_t)(uintptr_t)p + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u});
    else return *(const half8*)(p + i * 2048);
  };
  auto mfma_or_not = [&](float4v& c, const half8& w, const half8& x) {
    if constexpr (PROMPT_ABL & 16) {
      const uint4v v = __builtin_bit_cast(uint4v, x) ^ __builtin_bit_cast(uint4v, w);
      c[0] += __uint_as_float((v[0] ^ v[1] ^ v[2] ^ v[3]) & 0x3f800000u);
    } else {
      c = mfma16(w, x, c);                             // D[n = 4 g + q][m = r]
    }
  };

  if constexpr (PP) {
    // ---- ping-pong pipeline (8 wavefronts = two per SIMD).  A 64-deep tile runs as PH phases (PH = 1: both k steps at
    // once, PH = 2: one k step each); a phase is a LOAD segment (the phase's fragment reads, its share of the expansion
    // of tile t + 1 into the other weight buffer, the tile's global requests) and a COMPUTE segment (the phase's MFMAs,
    // nothing else), with a workgroup barrier after every segment.  The wavefronts of row half 1 (waves 4 - 7: the SIMD
    // partners of waves 0 - 3) pass one extra barrier before the loop, so they are always one segment behind: while one
    // wavefront of a SIMD holds the matrix pipe its partner reads LDS, expands weights and issues loads, and neither
    // sits in the other's issue slots.  What makes the skew safe:
    //  * x rows are staged by the row half that reads them, and a half is in step with itself;
    //  * a weight buffer is last read in the last load segment of its tile and first rewritten in the first load
    //    segment of the next tile, when the trailing half is already in the compute segment behind that last read; its
    //    expansion ends in the last load segment of the tile before, one segment before the leading half's first read;
    //  * LDS stores and fragment reads are waited for (lgkmcnt(0)) before the barrier that ends their segment, global
    //    requests (x tile t + 1, issued in the first segment of tile t; raw weight tiles, issued two tiles ahead)
    //    before the last barrier of the tile.
    constexpr int SP = 2 / PH;                       // k steps per phase
    half8 wf[SP][NTW], xf[SP][MTW];
    auto seg_end = [&]() {
      __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(63));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(PROMPT_ABL & 8)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto load_seg = [&](auto curc, auto phc, int t) {
      constexpr int cur = decltype(curc)::value, nxt = cur ^ 1, ph = decltype(phc)::value;
#pragma unroll
      for (int s = 0; s < SP; ++s) {
        const char* pw = lds_w(cur, ph * SP + s);
        const char* px = lds_x(cur, ph * SP + s);
#pragma unroll
        for (int n = 0; n < NTW; ++n) wf[s][n] = frag(pw, n);
        // (one phase per tile: the x fragments of the second k step are read under the first step's MFMAs: x rows belong
        // to this row half alone, while a weight buffer may be rewritten by the other half during the compute segment)
        if (PH == 1 && s == 1) continue;
#pragma unroll
        for (int j = 0; j < MTW; ++j) xf[s][j] = frag(px, j);
      }
      if constexpr (ph == 0) {
        if constexpr (!(PROMPT_ABL & 2)) stage_a(min(t + 1, T - 1), nxt);
        // odd tile 2 j + 1: the raw T16 tile j + 2 (first expanded two tiles later) goes into the set copied out after
        // the previous tile
        if constexpr (cur == 1 && !(PROMPT_ABL & 1)) fetch_raw(rawN, min((t >> 1) + 2, nkt - 1));
      }
      expand(std::integral_constant<int, nxt>{}, std::integral_constant<int, ph * NUW / PH>{},
             std::integral_constant<int, (ph + 1) * NUW / PH>{});
      seg_end();
    };
    auto comp_seg = [&](auto curc, auto phc) {
      constexpr int cur = decltype(curc)::value, ph = decltype(phc)::value;
      __builtin_amdgcn_s_setprio(1);
      if constexpr (PH == 1) {
        const char* px = lds_x(cur, 1);
#pragma unroll
        for (int j = 0; j < MTW; ++j) xf[1][j] = frag(px, j);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < SP; ++s) {
#pragma unroll
        for (int j = 0; j < MTW; ++j)
#pragma unroll
          for (int n = 0; n < NTW; ++n) mfma_or_not(acc[j][n], wf[s][n], xf[s][j]);
        if constexpr (PH == 1) __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ph == PH - 1) {
        __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
        if constexpr (cur == 0) {                      // even tile done: its T16 tile is used up, the next one has landed
#pragma unroll
          for (int i = 0; i < SLOTS; ++i) rawC[i] = rawN[i];
        }
      }
      if constexpr (!(PROMPT_ABL & 8)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    fetch_raw(rawC, 0);
    stage_a(0, 0);
    fetch_raw(rawN, min(1, nkt - 1));
    __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    expand(I0{}, I0{}, std::integral_constant<int, NUW>{});
    __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(63));
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < T; t += 2) {
      load_seg(I0{}, I0{}, t);
      comp_seg(I0{}, I0{});
      if constexpr (PH == 2) {
        load_seg(I0{}, I1{}, t);
        comp_seg(I0{}, I1{});
      }
      load_seg(I1{}, I0{}, t + 1);
      comp_seg(I1{}, I0{});
      if constexpr (PH == 2) {
        load_seg(I1{}, I1{}, t + 1);
        comp_seg(I1{}, I1{});
      }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
  } else {
    // ---- software-pipelined form (4 wavefronts, one per SIMD: no partner to alternate with; none of the configurations
    // in gemm_prompt_launch uses it since the 128 x 128 tile went to 8 wavefronts).  Tile t (LDS buffers t % 2)
    // runs as four phases of HM x NTW MFMAs: (k step 0, row half 0), (0, 1), (1, 0), (1, 1); fragments live in two named
    // register sets per operand (the set of the phase in flight and the set being filled for the next one; static
    // indices: a run-time set index would put them in scratch).  Phases 0 - 2 also expand tile t + 1 into the other weight
    // buffer and read the next phase's fragments; the scheduler is asked to put one MFMA, then a few vector instructions,
    // then an LDS read, and so on, so that expansion and reads ride in the issue slots between the MFMAs.  Phase 3 opens
    // with the tile's only barrier: behind it every wavefront has its last fragments of tile t in registers and tile
    // t + 1 is complete in LDS (own stores: lgkmcnt(0); x tile t + 1 and the raw weights, requested a tile or more
    // earlier: vmcnt(0)), so the phase requests x tile t + 2 into the buffer just released, reads the first fragments of
    // tile t + 1 and runs its MFMAs under those reads.
    constexpr int HM = MTW / 2;
    static_assert(MTW % 2 == 0 && HALF == 0, "a phase covers half of the wavefront's row tiles");
    half8 wfA[NTW], wfB[NTW], xfA[HM], xfB[HM];
    auto read_w = [&](half8 (&wf)[NTW], int buf, int s) {
      const char* p = lds_w(buf, s);
#pragma unroll
      for (int n = 0; n < NTW; ++n) wf[n] = frag(p, n);
    };
    auto read_x = [&](half8 (&xf)[HM], int buf, int s, int half) {
      const char* p = lds_x(buf, s) + half * (HM * 2048);
#pragma unroll
      for (int j = 0; j < HM; ++j) xf[j] = frag(p, j);
    };
    auto mma = [&](const half8 (&wf)[NTW], const half8 (&xf)[HM], auto halfc) {
      constexpr int half = decltype(halfc)::value;
#pragma unroll
      for (int j = 0; j < HM; ++j)
#pragma unroll
        for (int n = 0; n < NTW; ++n) mfma_or_not(acc[half * HM + j][n], wf[n], xf[j]);
    };
    auto interleave = [&](auto nmc, auto nvc, auto nrc, auto nwc) {
      constexpr int nm = decltype(nmc)::value, nv = decltype(nvc)::value, nr = decltype(nrc)::value, nw = decltype(nwc)::value;
#pragma unroll
      for (int k = 0; k < nm; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (k < nr) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (nv > 0) __builtin_amdgcn_sched_group_barrier(0x002, nv, 0);
        if (k >= nm - nw) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    };
    constexpr int PM = HM * NTW;                      // MFMAs per phase
    constexpr int U0 = NUW - 2 * (NUW / 3), U1 = U0 + NUW / 3, U2 = NUW;     // units per phase: 4 -> 2, 1, 1
    constexpr int VPU = 16;                           // vector instructions of one unit's expansion (scheduling hint)
    typedef std::integral_constant<int, PM> CPM;
    auto tile_body = [&](auto curc, int t) {
      constexpr int cur = decltype(curc)::value, nxt = cur ^ 1;
      typedef std::integral_constant<int, nxt> HN;
      // phase 0
      read_x(xfB, cur, 0, 1);
      expand(HN{}, std::integral_constant<int, 0>{}, std::integral_constant<int, U0>{});
      mma(wfA, xfA, I0{});
      interleave(CPM{}, std::integral_constant<int, (U0 * VPU + PM - 1) / PM>{}, std::integral_constant<int, HM>{}, std::integral_constant<int, U0>{});
      __builtin_amdgcn_sched_barrier(0);
      // phase 1
      read_w(wfB, cur, 1);
      read_x(xfA, cur, 1, 0);
      expand(HN{}, std::integral_constant<int, U0>{}, std::integral_constant<int, U1>{});
      mma(wfA, xfB, I1{});
      interleave(CPM{}, std::integral_constant<int, ((U1 - U0) * VPU + PM - 1) / PM>{}, std::integral_constant<int, HM + NTW>{}, std::integral_constant<int, U1 - U0>{});
      __builtin_amdgcn_sched_barrier(0);
      // phase 2
      read_x(xfB, cur, 1, 1);
      expand(HN{}, std::integral_constant<int, U1>{}, std::integral_constant<int, U2>{});
      mma(wfB, xfA, I0{});
      interleave(CPM{}, std::integral_constant<int, ((U2 - U1) * VPU + PM - 1) / PM>{}, std::integral_constant<int, HM>{}, std::integral_constant<int, U2 - U1>{});
      __builtin_amdgcn_sched_barrier(0);
      // phase 3
      __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));
      if constexpr (!(PROMPT_ABL & 8)) __builtin_amdgcn_s_barrier();
      if constexpr (cur == 0) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) rawC[i] = rawN[i];
      }
      if constexpr (!(PROMPT_ABL & 2)) stage_a(min(t + 2, T - 1), cur);
      if constexpr (cur == 1 && !(PROMPT_ABL & 1)) fetch_raw(rawN, min((t >> 1) + 2, nkt - 1));
      read_w(wfA, nxt, 0);
      read_x(xfA, nxt, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma(wfB, xfB, I1{});
      __builtin_amdgcn_sched_barrier(0);
    };
    fetch_raw(rawC, 0);
    stage_a(0, 0);
    fetch_raw(rawN, min(1, nkt - 1));
    __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    expand(I0{}, I0{}, std::integral_constant<int, NUW>{});
    __builtin_amdgcn_sched_barrier(0);
    stage_a(1, 1);
    __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(63));    // own LDS stores
    __builtin_amdgcn_s_barrier();
    read_w(wfA, 0, 0);
    read_x(xfA, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    for (int t = 0; t < T; t += 2) {
      tile_body(I0{}, t);
      tile_body(I1{}, t + 1);
    }
  }
  __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));         // the clamped look-ahead loads of the last tile

  // ---- epilogue.  A lane holds 4 consecutive output columns (n = 16 tile + 4 g + q) of row m = 16 j + r.
  // Two forms: fp32 partial slabs (K splits, or an output the staged form does not cover: the reduce kernel of
  // gemm_lowbit.hip then applies bias / activation / residual), or the staged 16-bit output.
  // (pair form: the wavefronts of the second column half hold the second product of the first half's columns)
  const int wn_out = pair ? wn % (WN / 2) : wn;
  const bool second_half = pair && wn >= WN / 2;
  const int ocol0 = nb * BNO + wn_out * (NTW * 16), orow0 = row0 + wm * (MTW * 16);
  const int64_t mtot = a.M;
  if (!pair && (a.splitk > 1 || a.force_partial)) {
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const int ncol = ocol0 + 16 * n + 4 * g;
      if (ncol >= a.N) continue;
#pragma unroll
      for (int j = 0; j < MTW; ++j) {
        const int64_t orow = orow0 + 16 * j + r;
        if (orow >= row0 + rows) continue;
        float* dst = a.partial + ((int64_t)sp * mtot + orow) * a.N + ncol;
        if (ncol + 4 <= a.N && (a.N & 3) == 0) *(float4v*)dst = acc[j][n];
        else {
#pragma unroll
          for (int q = 0; q < 4; ++q)             // (static register indices: a run-time index would put acc in scratch)
            if (ncol + q < a.N) dst[q] = acc[j][n][q];
        }
      }
    }
    return;
  }
  // staged form (host: 16-bit output, N % 8 == 0, ldo % 8 == 0, 16-byte aligned out / residual): bias added to the fp32
  // accumulators, rounded to the output type, the wavefront's tile transposed through LDS ([16 MTW rows][16 NTW
  // columns], rows padded by 16 bytes), then whole row segments (16 bytes per lane) leave with activation and residual
  // applied (a rolled loop: the activation's code exists once)
  constexpr int RB = 32 * NTW + 16;
  char* stg = smem + wave * (MTW * 16 * RB);
  __builtin_amdgcn_s_barrier();                      // every wavefront's last look-ahead tile has landed and been read
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    const int ncol = ocol0 + 16 * n + 4 * g;
    float bias[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const _Float16* bsrc = (const _Float16*)(second_half ? a.bias2 : a.bias);
    if (bsrc != nullptr && ncol < a.N) {
      const half4 b4 = *(const half4*)(bsrc + ncol);
#pragma unroll
      for (int q = 0; q < 4; ++q) bias[q] = (float)b4[q];
    }
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
      uint2v pk;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = acc[j][n][q] + bias[q];
      if (a.out_type == OUT_F16) {
        pk = __builtin_bit_cast(uint2v, half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]});
      } else {
        pk.x = (uint32_t)f32_to_bf16_bits(v[0]) | ((uint32_t)f32_to_bf16_bits(v[1]) << 16);
        pk.y = (uint32_t)f32_to_bf16_bits(v[2]) | ((uint32_t)f32_to_bf16_bits(v[3]) << 16);
      }
      *(uint2v*)(stg + (16 * j + r) * RB + (16 * n + 4 * g) * 2) = pk;
    }
  }
  __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(63));
  constexpr int CH = 2 * NTW;                        // 16-byte chunks per row
  static_assert((MTW * 16 * CH) % 64 == 0, "row chunks do not divide over the lanes");
  const bool plain = a.act == ACT_NONE && a.residual == nullptr && !pair;
  const char* stg2 = stg + (WN / 2) * (MTW * 16 * RB);     // pair form: the partner wavefront's tile (same rows)
  if (pair) {
    __builtin_amdgcn_s_barrier();                    // the partner's tile is complete in LDS
    if (second_half) return;
  }
#pragma unroll 1
  for (int it = 0; it < MTW * 16 * CH / 64; ++it) {
    const int idx = it * 64 + lane, row = idx / CH, ch = idx % CH;
    const int col = ocol0 + ch * 8;
    if (orow0 + row >= row0 + rows || col >= a.N) continue;
    uint4v pk = *(const uint4v*)(stg + row * RB + ch * 16);
    const int64_t oi = (int64_t)(orow0 + row) * a.ldo + col;
    if (pair) {
      // act(first) * second on the rounded 16-bit values (what a separate activation kernel would read)
      const uint4v pu = *(const uint4v*)(stg2 + row * RB + ch * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo, hi, ulo, uhi;
        if (a.out_type == OUT_F16) {
          const half2v h = u2h(pk[e]), hu = u2h(pu[e]);
          lo = (float)h[0]; hi = (float)h[1]; ulo = (float)hu[0]; uhi = (float)hu[1];
        } else {
          lo = bf16_bits_to_f32((uint16_t)(pk[e] & 0xffffu)); hi = bf16_bits_to_f32((uint16_t)(pk[e] >> 16));
          ulo = bf16_bits_to_f32((uint16_t)(pu[e] & 0xffffu)); uhi = bf16_bits_to_f32((uint16_t)(pu[e] >> 16));
        }
        lo = act_apply(a.act, lo) * ulo;
        hi = act_apply(a.act, hi) * uhi;
        if (a.out_type == OUT_F16) pk[e] = h2u(half2v{(_Float16)lo, (_Float16)hi});
        else pk[e] = (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
      }
    } else if (!plain) {
      uint4v rr = uint4v{0, 0, 0, 0};
      if (a.residual != nullptr) rr = *(const uint4v*)((const uint16_t*)a.residual + oi);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo, hi, rlo, rhi;
        if (a.out_type == OUT_F16) {
          const half2v h = u2h(pk[e]), hr = u2h(rr[e]);
          lo = (float)h[0]; hi = (float)h[1]; rlo = (float)hr[0]; rhi = (float)hr[1];
        } else {
          lo = bf16_bits_to_f32((uint16_t)(pk[e] & 0xffffu)); hi = bf16_bits_to_f32((uint16_t)(pk[e] >> 16));
          rlo = bf16_bits_to_f32((uint16_t)(rr[e] & 0xffffu)); rhi = bf16_bits_to_f32((uint16_t)(rr[e] >> 16));
        }
        lo = act_apply(a.act, lo) + rlo;
        hi = act_apply(a.act, hi) + rhi;
        if (a.out_type == OUT_F16) pk[e] = h2u(half2v{(_Float16)lo, (_Float16)hi});
        else pk[e] = (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
      }
    }
    *(uint4v*)((uint16_t*)a.out + oi) = pk;
  }
}

template <int QT, int WM, int WN, int MTW, int NTW, int PH>
static int gemm_prompt_launch_cfg(const GemmArgs& a, hipStream_t st) {
  constexpr int NW = WM * WN, BM = WM * MTW * 16, BN = WN * NTW * 16;
  const size_t lds = prompt_lds_tiles(BM, BN, NW, MTW, NTW) + (QT == Q_NF4 ? 16 * 32 * 4 : 0);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_prompt_kernel<QT, WM, WN, MTW, NTW, PH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return -101;
    attr_set = true;
  }
  const int bno = a.w2 != nullptr ? BN / 2 : BN;
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + bno - 1) / bno);
  hipLaunchKernelGGL((gemm_prompt_kernel<QT, WM, WN, MTW, NTW, PH>), dim3(tiles * a.splitk), dim3(64 * NW), lds, st, a);
  return HIP_CHECK_LAUNCH();
}

// configurations in the order of gemm_prompt.hip's table: 256 x 256, 256 x 192, 128 x 128 (8 wavefronts each)
#define PROMPT_NCFG 3
template <int QT>
static int gemm_prompt_launch(const GemmArgs& a, int cfg, hipStream_t st) {
  switch (cfg) {
    case 0: return gemm_prompt_launch_cfg<QT, 2, 4, 8, 4, PROMPT_PH0>(a, st);
    case 1: return gemm_prompt_launch_cfg<QT, 2, 4, 8, 3, PROMPT_PH1>(a, st);
    // 128 x 128 also runs on 8 wavefronts as a ping-pong pair of row halves (one phase per 64-deep tile): 43.8 us on
    // the 1024 x 4096 x 4096 product against 54.4 us with 4 wavefronts of 64 x 64 each (and ahead on every other shape).
    // 128 x 256 and 256 x 128 tiles on 8 wavefronts were measured too: within 4 % of the planned choice at best.
    case 2: return gemm_prompt_launch_cfg<QT, 2, 4, 4, 2, 1>(a, st);
    default: return -4;
  }
}

}  // namespace bigdl
