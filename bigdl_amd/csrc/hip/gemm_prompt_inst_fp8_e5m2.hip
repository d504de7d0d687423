// fp8_e5m2 instantiations of the prompt-sized fused GEMM (one unit per format: parallel compilation)
#include "gemm_prompt_kernel.h"
namespace bigdl {
int gemm_prompt_launch_fp8_e5m2(const GemmArgs& a, int cfg, hipStream_t st) { return gemm_prompt_launch<Q_FP8_E5M2>(a, cfg, st); }
}  // namespace bigdl
