// Host entry point of the dequant-fused skinny GEMM (kernel: gemv_kernel.h; one instantiation
// unit per weight format: gemv_inst_*.hip).
#include "gemv_kernel.h"

namespace bigdl {
int gemv_launch_sym_int4(const GemvArgs& a, bool gate_up, hipStream_t st);
int gemv_launch_asym_int4(const GemvArgs& a, bool gate_up, hipStream_t st);
int gemv_launch_woq_int4(const GemvArgs& a, bool gate_up, hipStream_t st);
int gemv_launch_fp4(const GemvArgs& a, bool gate_up, hipStream_t st);
int gemv_launch_nf4(const GemvArgs& a, bool gate_up, hipStream_t st);
int gemv_launch_sym_int5(const GemvArgs& a, bool gate_up, hipStream_t st);
int gemv_launch_asym_int5(const GemvArgs& a, bool gate_up, hipStream_t st);
int gemv_launch_fp6(const GemvArgs& a, bool gate_up, hipStream_t st);
int gemv_launch_sym_int8(const GemvArgs& a, bool gate_up, hipStream_t st);
int gemv_launch_fp8_e4m3(const GemvArgs& a, bool gate_up, hipStream_t st);
int gemv_launch_fp8_e5m2(const GemvArgs& a, bool gate_up, hipStream_t st);
}  // namespace bigdl

using namespace bigdl;

static int dispatch(const GemvArgs& a, int qtype, hipStream_t stream) {
  const bool gu = a.w2 != nullptr;
  switch (qtype) {
    case Q_SYM_INT4: return gemv_launch_sym_int4(a, gu, stream);
    case Q_ASYM_INT4: return gemv_launch_asym_int4(a, gu, stream);
    case Q_WOQ_INT4: return gemv_launch_woq_int4(a, gu, stream);
    case Q_FP4: return gemv_launch_fp4(a, gu, stream);
    case Q_NF4: return gemv_launch_nf4(a, gu, stream);
    case Q_SYM_INT5: return gemv_launch_sym_int5(a, gu, stream);
    case Q_ASYM_INT5: return gemv_launch_asym_int5(a, gu, stream);
    case Q_FP6: return gemv_launch_fp6(a, gu, stream);
    case Q_SYM_INT8: return gemv_launch_sym_int8(a, gu, stream);
    case Q_FP8_E4M3: return gemv_launch_fp8_e4m3(a, gu, stream);
    case Q_FP8_E5M2: return gemv_launch_fp8_e5m2(a, gu, stream);
    default: return -2;
  }
}

// Returns 0 on success; negative = rejected on the host before any launch.
// M <= 64 (<= 32 with w2).  x fp16 with ldx % 8 == 0 and 16-byte aligned rows; K % 32 == 0.
static int gemv_lowbit_impl(const void* x, const void* w, const void* w2, void* out, const void* bias,
                            const void* bias2, const void* residual, const void* gamma, float eps,
                            float gamma_offset, int M, int N, int K, int ldx, int ldo, int qtype, int act,
                            int out_type, const float* att_ws, int att_splits, int att_D, int att_G, int att_Hkv,
                            int att_qlen, int att_rows_alloc, const float* ln_c1, const float* ln_c2,
                            const float* ln2_c1, const float* ln2_c2, hipStream_t stream) {
  qtype = t16_kernel_qtype(qtype);
  if (ln_c1 != nullptr) {
    // LayerNorm prologue: gamma and both per-column vectors (and those of w2), no gemma offset, x staged in LDS
    if (!gamma || !ln_c2 || att_ws || gamma_offset != 0.0f || (w2 != nullptr) != (ln2_c1 != nullptr) ||
        (ln2_c1 != nullptr) != (ln2_c2 != nullptr))
      return -7;
  }
  if (M <= 0 || N <= 0 || K <= 0 || (K & 31) || (ldx & 7) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return -1;
  if (gamma && ((uintptr_t)gamma & 15)) return -1;
  if (w2 && ((uintptr_t)w2 & 15)) return -1;
  if (out_type < 0 || out_type > 2) return -3;
  if (M > (w2 ? 32 : 64)) return -4;
  if (att_ws) {
    if (att_splits <= 0 || att_D <= 0 || (att_D & 7) || att_G <= 0 || att_Hkv <= 0 || att_qlen <= 0 || att_rows_alloc <= 0 ||
        K != att_Hkv * att_G * att_D || M % att_qlen || att_G * att_qlen > att_rows_alloc || gamma || ((uintptr_t)att_ws & 15) ||
        M > 16 || w2 != nullptr)
      return -7;
    // the combine is only implemented in the LDS staging prologue
    if ((size_t)M * ((K + 127) / 128 * 128 + XTAIL + XPAD) * 2 > GEMV_XLDS_LIMIT) return -7;
  }
  GemvArgs a;
  a.x = (const _Float16*)x; a.w = w; a.w2 = w2; a.out = out; a.bias = bias; a.bias2 = bias2;
  a.residual = residual; a.gamma = (const _Float16*)gamma; a.eps = eps; a.gamma_offset = gamma_offset;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.act = act; a.out_type = out_type;
  a.att_ws = att_ws; a.att_splits = att_splits; a.att_D = att_D; a.att_G = att_G; a.att_Hkv = att_Hkv;
  a.att_qlen = att_qlen; a.att_rows_alloc = att_rows_alloc;
  a.ln_c1 = ln_c1; a.ln_c2 = ln_c2; a.ln2_c1 = ln2_c1; a.ln2_c2 = ln2_c2;
  a.slot_expert = nullptr; a.slot_scale = nullptr; a.n_total = 0; a.slot_xdiv = 1; a.n_slots = 0; a.slot_cat = 0;
  return dispatch(a, qtype, stream);
}

extern "C" int bigdl_gemv_lowbit(const void* x, const void* w, const void* w2, void* out, const void* bias,
                                 const void* bias2, const void* residual, const void* gamma, float eps,
                                 float gamma_offset, int M, int N, int K, int ldx, int ldo, int qtype, int act,
                                 int out_type, const float* att_ws, int att_splits, int att_D, int att_G, int att_Hkv,
                                 int att_qlen, int att_rows_alloc, hipStream_t stream) {
  return gemv_lowbit_impl(x, w, w2, out, bias, bias2, residual, gamma, eps, gamma_offset, M, N, K, ldx, ldo, qtype, act,
                          out_type, att_ws, att_splits, att_D, att_G, att_Hkv, att_qlen, att_rows_alloc, nullptr, nullptr,
                          nullptr, nullptr, stream);
}

// y = act(LayerNorm(x; gamma, beta) @ W^T + bias) ...: gamma is the LayerNorm weight, ln_c1[n] = sum_k W[n,k] gamma[k]
// and ln_c2[n] = sum_k W[n,k] beta[k] (fp32 [N], computed once per weight; ln2_* for w2).  Only for row counts whose
// x fits the LDS staging budget (returns -7 otherwise).
extern "C" int bigdl_gemv_lowbit_ln(const void* x, const void* w, const void* w2, void* out, const void* bias,
                                    const void* bias2, const void* residual, const void* gamma, float eps, int M, int N,
                                    int K, int ldx, int ldo, int qtype, int act, int out_type, const float* ln_c1,
                                    const float* ln_c2, const float* ln2_c1, const float* ln2_c2, hipStream_t stream) {
  if (!ln_c1) return -1;
  return gemv_lowbit_impl(x, w, w2, out, bias, bias2, residual, gamma, eps, 0.0f, M, N, K, ldx, ldo, qtype, act, out_type,
                          nullptr, 0, 0, 0, 0, 1, 0, ln_c1, ln_c2, ln2_c1, ln2_c2, stream);
}

// Routed-expert products for decode: for every slot s < n_slots (token-major, `slot_xdiv` consecutive slots share
// one x row; pass 1 when every slot has its own row):
//   out[s, :] = slot_scale[s] * f(x[s / slot_xdiv, :] . dequant(W[slot_expert[s]])^T)
// w (and w2 for the fused gate/up + activation form) are T16 buffers of the experts' [N, K] matrices stacked by rows
// ([n_experts * N, K], N % 16 == 0).  slot_expert (int32, device) entries must be < n_experts; negative = not on
// this rank (zero row).
extern "C" int bigdl_gemv_lowbit_moe(const void* x, const void* w, const void* w2, void* out, const int* slot_expert,
                                     const float* slot_scale, const void* gamma, float eps, float gamma_offset,
                                     int n_slots, int slot_xdiv, int n_experts, int N, int K, int ldx, int ldo,
                                     int qtype, int act, int out_type, hipStream_t stream) {
  qtype = t16_kernel_qtype(qtype);
  if (n_slots <= 0 || n_slots > 65535 || slot_xdiv <= 0 || N <= 0 || K <= 0 || (K & 31) || (ldx & 7) || ((uintptr_t)x & 15) ||
      ((uintptr_t)w & 15) || (N & 15) || n_experts <= 0 || slot_expert == nullptr)
    return -1;
  if (gamma && ((uintptr_t)gamma & 15)) return -1;
  if (w2 && ((uintptr_t)w2 & 15)) return -1;
  if (out_type < 0 || out_type > 2) return -3;
  GemvArgs a;
  a.x = (const _Float16*)x; a.w = w; a.w2 = w2; a.out = out; a.bias = nullptr; a.bias2 = nullptr;
  a.residual = nullptr; a.gamma = (const _Float16*)gamma; a.eps = eps; a.gamma_offset = gamma_offset;
  a.M = 1; a.N = N; a.K = K; a.ldx = ldx; a.ldo = ldo; a.act = act; a.out_type = out_type;
  a.att_ws = nullptr; a.att_splits = 0; a.att_D = 0; a.att_G = 0; a.att_Hkv = 0; a.att_qlen = 1; a.att_rows_alloc = 0;
  a.ln_c1 = nullptr; a.ln_c2 = nullptr; a.ln2_c1 = nullptr; a.ln2_c2 = nullptr;
  a.slot_expert = slot_expert; a.slot_scale = slot_scale; a.n_total = n_experts * N; a.slot_xdiv = slot_xdiv;
  a.n_slots = n_slots; a.slot_cat = 0;
  return dispatch(a, qtype, stream);
}

// The second routed product of a decode step with the sum over a token's experts and the residual add inside the launch:
//   out[t, :] = residual[t, :] + sum_j x[t * top_k + j, :] . dequant(W[slot_expert[t * top_k + j]])^T
// x [tokens * top_k, K] fp16 contiguous rows (already scaled by the routing weights; zero rows for negative ids), w the
// experts' [N, K] matrices stacked by rows as above.  K % 128 == 0, top_k <= 8, one workgroup per 16 output columns and
// token walks the top_k weight tile rows back to back (a product over the concatenated reduction axis).  out may alias
// residual.
extern "C" int bigdl_gemv_lowbit_moe_sum(const void* x, const void* w, void* out, const void* residual, const int* slot_expert,
                                         int tokens, int top_k, int n_experts, int N, int K, int ldo, int qtype, int out_type,
                                         hipStream_t stream) {
  qtype = t16_kernel_qtype(qtype);
  if (tokens <= 0 || tokens > 65535 || top_k <= 0 || top_k > 8 || N <= 0 || K <= 0 || (K & 127) || (N & 15) ||
      ((uintptr_t)x & 15) || ((uintptr_t)w & 15) || n_experts <= 0 || slot_expert == nullptr)
    return -1;
  if (out_type < 0 || out_type > 2) return -3;
  if ((size_t)(top_k * K + XTAIL + XPAD) * 2 > GEMV_XLDS_LIMIT_FEW_ROWS) return -7;
  GemvArgs a;
  a.x = (const _Float16*)x; a.w = w; a.w2 = nullptr; a.out = out; a.bias = nullptr; a.bias2 = nullptr;
  a.residual = residual; a.gamma = nullptr; a.eps = 0.0f; a.gamma_offset = 0.0f;
  a.M = 1; a.N = N; a.K = top_k * K; a.ldx = top_k * K; a.ldo = ldo; a.act = 0; a.out_type = out_type;
  a.att_ws = nullptr; a.att_splits = 0; a.att_D = 0; a.att_G = 0; a.att_Hkv = 0; a.att_qlen = 1; a.att_rows_alloc = 0;
  a.ln_c1 = nullptr; a.ln_c2 = nullptr; a.ln2_c1 = nullptr; a.ln2_c2 = nullptr;
  a.slot_expert = slot_expert; a.slot_scale = nullptr; a.n_total = n_experts * N; a.slot_xdiv = 1;
  a.n_slots = tokens; a.slot_cat = top_k;
  return dispatch(a, qtype, stream);
}

// Row counts [lo, hi] for which the skinny products take the CU-resident form (0, 0: never)
extern "C" void bigdl_gemv_cu_rows(int lo, int hi) {
  gemv_cu_rows_lo() = lo;
  gemv_cu_rows_hi() = hi;
}
