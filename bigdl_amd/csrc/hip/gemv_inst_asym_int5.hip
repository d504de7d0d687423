// Instantiation unit of the fused low-bit GEMV for one weight format (one file per format so
// that the formats compile in parallel).
#include "gemv_kernel.h"
namespace bigdl {
int gemv_launch_asym_int5(const GemvArgs& a, bool gate_up, hipStream_t st) { return launch_qt<Q_ASYM_INT5>(a, gate_up, st); }
}  // namespace bigdl
