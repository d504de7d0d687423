// Instantiation unit of the fused low-bit MFMA GEMM for one weight format (one file per format so that the
// formats compile in parallel).
#include "gemm_kernel.h"
namespace bigdl {
int gemm_launch_sym_int4(const GemmArgs& a, int bm, int bn, int mblocks, hipStream_t st) { return gemm_launch_qt<Q_SYM_INT4>(a, bm, bn, mblocks, st); }
}  // namespace bigdl
