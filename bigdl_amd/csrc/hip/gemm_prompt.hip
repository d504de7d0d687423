// Host entry of the prompt-sized dequant-fused MFMA GEMM (kernel: gemm_prompt_kernel.h; one instantiation unit per weight
// format: gemm_prompt_inst_*.hip): tile configuration / K-split planning and the C ABI.
#include "gemm_prompt_kernel.h"
#include <stdlib.h>

namespace bigdl {

int gemm_launch_reduce(const GemmArgs& a, hipStream_t stream);      // gemm_lowbit.hip: split-K reduce / epilogue

#define PROMPT_DECL(name) int gemm_prompt_launch_##name(const GemmArgs& a, int cfg, hipStream_t st);
PROMPT_DECL(sym_int4) PROMPT_DECL(asym_int4) PROMPT_DECL(woq_int4) PROMPT_DECL(fp4) PROMPT_DECL(nf4) PROMPT_DECL(sym_int5)
PROMPT_DECL(asym_int5) PROMPT_DECL(fp6) PROMPT_DECL(sym_int8) PROMPT_DECL(fp8_e4m3) PROMPT_DECL(fp8_e5m2)
#undef PROMPT_DECL

static int prompt_dispatch(const GemmArgs& a, int qtype, int cfg, hipStream_t st) {
  switch (qtype) {
    case Q_SYM_INT4: return gemm_prompt_launch_sym_int4(a, cfg, st);
    case Q_ASYM_INT4: return gemm_prompt_launch_asym_int4(a, cfg, st);
    case Q_WOQ_INT4: return gemm_prompt_launch_woq_int4(a, cfg, st);
    case Q_FP4: return gemm_prompt_launch_fp4(a, cfg, st);
    case Q_NF4: return gemm_prompt_launch_nf4(a, cfg, st);
    case Q_SYM_INT5: return gemm_prompt_launch_sym_int5(a, cfg, st);
    case Q_ASYM_INT5: return gemm_prompt_launch_asym_int5(a, cfg, st);
    case Q_FP6: return gemm_prompt_launch_fp6(a, cfg, st);
    case Q_SYM_INT8: return gemm_prompt_launch_sym_int8(a, cfg, st);
    case Q_FP8_E4M3: return gemm_prompt_launch_fp8_e4m3(a, cfg, st);
    case Q_FP8_E5M2: return gemm_prompt_launch_fp8_e5m2(a, cfg, st);
    default: return -2;
  }
}

// Tile configurations (gemm_prompt_launch<QT> instantiates them in this order) and their relative MFMA efficiency
// (measured: benchmark/gemm_big_bench.py --sweep).  One workgroup per CU for the 8-wavefront tiles.
struct PromptCfg { int bm, bn; double eff; };
static const PromptCfg kPromptCfgs[PROMPT_NCFG] = {{256, 256, 1.0}, {256, 192, 0.79}, {128, 128, 0.65}};

struct PromptPlan { int cfg, splitk, kt_per_split; };
static int g_force_cfg = -1, g_force_sk = 0;         // bigdl_gemm_prompt_force (tests / sweeps)

// Cost of a launch in units of (output element x 128-deep k tile) per CU: rounds of 256 concurrent tiles, every tile
// paying its k tiles + a fixed prologue / epilogue; K splits add the fp32 slab traffic and the reduce launch.
// pair: the gate / up form (two products per output column: blocks cover bn / 2 output columns, no K split)
static PromptPlan prompt_plan(int M, int N, int K, bool pair = false) {
  const int KT = K >> 7;
  static const int env_cfg = getenv("BIGDL_PROMPT_CFG") ? atoi(getenv("BIGDL_PROMPT_CFG")) : -1;
  static const int env_sk = getenv("BIGDL_PROMPT_SPLITK") ? atoi(getenv("BIGDL_PROMPT_SPLITK")) : 0;
  const int forced_cfg = g_force_cfg >= 0 ? g_force_cfg : env_cfg, forced_sk = g_force_sk > 0 ? g_force_sk : env_sk;
  PromptPlan best = {0, 1, KT};
  double best_cost = 1e300;
  for (int c = 0; c < PROMPT_NCFG; ++c) {
    if (forced_cfg >= 0 && c != forced_cfg) continue;
    const PromptCfg& pc = kPromptCfgs[c];
    const int bno = pair ? pc.bn / 2 : pc.bn;
    const int64_t tiles = (int64_t)((M + pc.bm - 1) / pc.bm) * ((N + bno - 1) / bno);
    for (int sk = 1; sk <= (pair ? 1 : 8); ++sk) {
      if (forced_sk ? sk != forced_sk : (sk > 1 && KT / sk < 8)) continue;
      if (sk > KT) continue;
      const int per = (KT + sk - 1) / sk, rsk = (KT + per - 1) / per;
      const int64_t rounds = (tiles * rsk + 255) / 256;
      double cost = (double)rounds * pc.bm * pc.bn / pc.eff * (per + 1.5);
      if (rsk > 1) cost += 0.035 * rsk * (double)M * N + 6e4;
      if (cost < best_cost) { best_cost = cost; best = {c, rsk, per}; }
    }
  }
  return best;
}

// the staged 16-bit epilogue covers this output; otherwise one fp32 slab + the reduce kernel
static bool prompt_staged_ok(int N, int ldo, int out_type, const void* out, const void* residual) {
  return out_type != OUT_F32 && (N & 7) == 0 && (ldo & 7) == 0 && (((uintptr_t)out | (uintptr_t)residual) & 15) == 0;
}

}  // namespace bigdl

using namespace bigdl;

extern "C" int bigdl_gemm_prompt_supported(int qtype, int K) {
  qtype = t16_kernel_qtype(qtype);
  return is_t16(qtype) && K >= 256 && (K & 127) == 0;
}

// fp32 slabs of M x N floats the call below needs as workspace (0: none)
extern "C" int bigdl_gemm_prompt_slabs(int M, int N, int K, int ldo, int out_type, int out_aligned16) {
  const PromptPlan pl = prompt_plan(M, N, K);
  if (pl.splitk > 1) return pl.splitk;
  const bool staged = out_type != OUT_F32 && (N & 7) == 0 && (ldo & 7) == 0 && out_aligned16;
  return staged ? 0 : 1;
}

// cfg >= 0 / splitk > 0 pin the tile configuration / K splits of the following calls (-1 / 0: planned)
extern "C" void bigdl_gemm_prompt_force(int cfg, int splitk) {
  g_force_cfg = cfg < PROMPT_NCFG ? cfg : -1;
  g_force_sk = splitk;
}

extern "C" int bigdl_gemm_prompt_plan(int M, int N, int K, int* cfg, int* splitk) {
  const PromptPlan pl = prompt_plan(M, N, K);
  if (cfg) *cfg = pl.cfg;
  if (splitk) *splitk = pl.splitk;
  return 0;
}

// Gate / up pair in one launch: out[M, N] = act(x W^T + bias) * (x W2^T + bias2), both weights [N, K] of one format,
// N % 16 == 0; 16-bit output with N % 8 == 0, ldo % 8 == 0 and a 16-byte aligned out (the staged epilogue; -7 otherwise).
extern "C" int bigdl_gemm_prompt_pair(const void* x, const void* w, const void* w2, void* out, const void* bias,
                                      const void* bias2, int M, int N, int K, int ldx, int ldo, int qtype, int act,
                                      int out_type, hipStream_t stream) {
  qtype = t16_kernel_qtype(qtype);
  if (M <= 0 || N <= 0 || (N & 15) || K < 256 || (K & 127) || (ldx & 7) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15) ||
      w2 == nullptr || ((uintptr_t)w2 & 15))
    return -1;
  if ((bias != nullptr && ((uintptr_t)bias & 7)) || (bias2 != nullptr && ((uintptr_t)bias2 & 7))) return -1;
  if (out_type != OUT_F16 && out_type != OUT_BF16) return -3;
  if (!prompt_staged_ok(N, ldo, out_type, out, nullptr)) return -7;
  if ((int64_t)M * ldx >= (1ll << 31)) return -6;
  GemmArgs a = {};
  a.x = (const _Float16*)x; a.w = w; a.w2 = w2; a.out = out; a.bias = bias; a.bias2 = bias2;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.act = act; a.out_type = out_type;
  const PromptPlan pl = prompt_plan(M, N, K, true);
  a.splitk = 1; a.kt_per_split = K >> 7;
  return prompt_dispatch(a, qtype, pl.cfg, stream);
}

// x fp16 [M, ldx] (ldx % 8 == 0, 16-byte aligned), K % 128 == 0.  workspace: fp32, >= slabs * M * N floats.
extern "C" int bigdl_gemm_prompt(const void* x, const void* w, void* out, const void* bias, const void* residual,
                                 float* workspace, int64_t workspace_floats, int M, int N, int K, int ldx, int ldo,
                                 int qtype, int act, int out_type, hipStream_t stream) {
  qtype = t16_kernel_qtype(qtype);
  if (M <= 0 || N <= 0 || K < 256 || (K & 127) || (ldx & 7) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return -1;
  if (bias != nullptr && ((uintptr_t)bias & 7)) return -1;
  if (out_type < 0 || out_type > 2) return -3;
  if ((int64_t)M * ldx >= (1ll << 31)) return -6;
  GemmArgs a = {};
  a.x = (const _Float16*)x; a.w = w; a.out = out; a.bias = bias; a.residual = residual; a.partial = workspace;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.act = act; a.out_type = out_type;
  const PromptPlan pl = prompt_plan(M, N, K);
  a.splitk = pl.splitk; a.kt_per_split = pl.kt_per_split;
  a.force_partial = (a.splitk == 1 && !prompt_staged_ok(N, ldo, out_type, out, residual)) ? 1 : 0;
  const bool partial = a.splitk > 1 || a.force_partial;
  if (partial && (workspace == nullptr || workspace_floats < (int64_t)a.splitk * M * N)) return -5;
  const int rc = prompt_dispatch(a, qtype, pl.cfg, stream);
  if (rc != 0 || !partial) return rc;
  return gemm_launch_reduce(a, stream);
}
