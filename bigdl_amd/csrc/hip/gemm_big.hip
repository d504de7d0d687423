// Host entry of the prompt-sized dequant-fused MFMA GEMM (kernel: gemm_big_kernel.h) and its instantiations.
#include "gemm_big_kernel.h"
#include <stdlib.h>

namespace bigdl {

int gemm_launch_reduce(const GemmArgs& a, hipStream_t stream);      // gemm_lowbit.hip: split-K reduce / epilogue

template <int NT>
static int big_dispatch_nt(const GemmArgs& a, int qtype, int mblocks, hipStream_t st) {
  switch (qtype) {
    case Q_SYM_INT4: return gemm_big_launch<Q_SYM_INT4, NT>(a, mblocks, st);
    case Q_ASYM_INT4: return gemm_big_launch<Q_ASYM_INT4, NT>(a, mblocks, st);
    case Q_WOQ_INT4: return gemm_big_launch<Q_WOQ_INT4, NT>(a, mblocks, st);
    case Q_NF4: return gemm_big_launch<Q_NF4, NT>(a, mblocks, st);
    case Q_FP4: return gemm_big_launch<Q_FP4, NT>(a, mblocks, st);
    default: return -2;
  }
}
static int big_dispatch(const GemmArgs& a, int qtype, int nt, int mblocks, hipStream_t st) {
  return nt == 3 ? big_dispatch_nt<3>(a, qtype, mblocks, st) : big_dispatch_nt<4>(a, qtype, mblocks, st);
}

// Tile width (64 * nt columns, nt = 3 or 4) and K splits for an M x N x K product.  Two workgroups share a CU: 512 slots.
// The launch takes rounds(blocks * splitk / 512) * nt * (k-tiles per split) units of time; a split must keep at least
// 8 k-tiles (1024 deep) so that the fp32 slabs stay small next to the product itself.
struct BigPlan { int nt, splitk, kt_per_split; };
static BigPlan big_plan(int M, int N, int K) {
  const int KT = K >> 7, mblocks = (M + 127) / 128;
  static const int forced_nt = getenv("BIGDL_GEMM_BIG_NT") ? atoi(getenv("BIGDL_GEMM_BIG_NT")) : 0;
  BigPlan best = {4, 1, KT};
  double best_cost = 1e30;
  // (256-column tiles, nt = 4, need more registers than there are with the three-slot weight ring: only when forced)
  for (int nt = 3; nt <= 4; ++nt) {
    if (forced_nt ? nt != forced_nt : nt != 3) continue;
    const int blocks = mblocks * ((N + 64 * nt - 1) / (64 * nt));
    for (int sk = 1; sk <= 8 && (sk == 1 || KT / sk >= 8); ++sk) {
      const int per = (KT + sk - 1) / sk, rsk = (KT + per - 1) / per;
      const int rounds = (blocks * rsk + 511) / 512;
      // k loop + epilogue / slab traffic per round (in k-tile units of one workgroup)
      const double cost = rounds * nt * (per + 2.0) + (rsk > 1 ? 1.5 * rsk : 0.0);
      if (cost < best_cost) { best_cost = cost; best = {nt, rsk, per}; }
    }
  }
  return best;
}

}  // namespace bigdl

using namespace bigdl;

extern "C" int bigdl_gemm_lowbit_big_supported(int qtype, int K) {
  qtype = t16_kernel_qtype(qtype);
  const bool q = qtype == Q_SYM_INT4 || qtype == Q_ASYM_INT4 || qtype == Q_WOQ_INT4 || qtype == Q_NF4 || qtype == Q_FP4;
  return q && K > 0 && (K & 127) == 0;
}

extern "C" int bigdl_gemm_lowbit_big_splitk(int M, int N, int K) {
  return big_plan(M, N, K).splitk;
}

// x fp16 [M, ldx] (ldx % 8 == 0, 16-byte aligned), K % 128 == 0.  workspace: fp32, >= splitk * M * N when splitk > 1.
extern "C" int bigdl_gemm_lowbit_big(const void* x, const void* w, void* out, const void* bias, const void* residual,
                                     float* workspace, int64_t workspace_floats, int M, int N, int K, int ldx, int ldo,
                                     int qtype, int act, int out_type, hipStream_t stream) {
  qtype = t16_kernel_qtype(qtype);
  if (M <= 0 || N <= 0 || K <= 0 || (K & 127) || (ldx & 7) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return -1;
  if (out_type < 0 || out_type > 2) return -3;
  if ((int64_t)M * ldx >= (1ll << 31)) return -6;
  GemmArgs a = {};
  a.x = (const _Float16*)x; a.w = w; a.out = out; a.bias = bias; a.residual = residual; a.partial = workspace;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.act = act; a.out_type = out_type;
  const int mblocks = (M + 127) / 128;
  const BigPlan pl = big_plan(M, N, K);
  a.splitk = pl.splitk; a.kt_per_split = pl.kt_per_split;
  if (a.splitk > 1 && (workspace == nullptr || workspace_floats < (int64_t)a.splitk * M * N)) return -5;
  const int rc = big_dispatch(a, qtype, pl.nt, mblocks, stream);
  if (rc != 0 || a.splitk == 1) return rc;
  return gemm_launch_reduce(a, stream);
}
