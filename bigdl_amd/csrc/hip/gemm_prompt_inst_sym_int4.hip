// sym_int4 instantiations of the prompt-sized fused GEMM (one unit per format: parallel compilation)
#include "gemm_prompt_kernel.h"
namespace bigdl {
int gemm_prompt_launch_sym_int4(const GemmArgs& a, int cfg, hipStream_t st) { return gemm_prompt_launch<Q_SYM_INT4>(a, cfg, st); }
}  // namespace bigdl
