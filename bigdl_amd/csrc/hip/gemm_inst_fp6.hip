// Instantiation unit of the fused low-bit MFMA GEMM for one weight format (one file per format so that the
// formats compile in parallel).
#include "gemm_kernel.h"
namespace bigdl {
int gemm_launch_fp6(const GemmArgs& a, int bm, int bn, int mblocks, hipStream_t st) { return gemm_launch_qt<Q_FP6>(a, bm, bn, mblocks, st); }
}  // namespace bigdl
