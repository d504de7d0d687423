// How fast can ONE compute unit pull a stream out of HBM, as a function of how many CUs pull at once?
// G workgroups (512 threads, one per CU up to 256) each read their own contiguous share of a 64 MB buffer with 16-byte
// loads, UNROLL loads in flight per lane.  Prints GB/s per workgroup and in total.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
template <int UNROLL>
__global__ __launch_bounds__(512) void pull(const uint4v* src, unsigned* out, size_t vec_per_wg) {
  const uint4v* p = src + (size_t)blockIdx.x * vec_per_wg;
  uint4v acc = {0, 0, 0, 0};
  for (size_t i = threadIdx.x; i < vec_per_wg; i += 512 * UNROLL) {
    uint4v v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + i + 512 * u);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[blockIdx.x] = 1;
}
int main() {
  const size_t total = 64u << 20;
  uint4v* src; unsigned* out;
  hipMalloc(&src, 4 * total); hipMalloc(&out, 4096);
  hipMemset(src, 1, 4 * total);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {16, 32, 64, 96, 128, 192, 256, 512, 1024};
  for (int g : grids) {
    const size_t vec_per_wg = total / 16 / g / (512 * 8) * (512 * 8);
    float best = 1e9;
    for (int rep = 0; rep < 8; ++rep) {
      const uint4v* base = src + (size_t)(rep & 3) * (total / 16);      // rotate over 256 MB: not cache resident
      hipEventRecord(e0);
      hipLaunchKernelGGL(pull<8>, dim3(g), dim3(512), 0, 0, base, out, vec_per_wg);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 2 && ms < best) best = ms;
    }
    const double bytes = (double)vec_per_wg * 16 * g;
    printf("{\"workgroups\": %d, \"kb_per_wg\": %.0f, \"us\": %.2f, \"gbps_total\": %.0f, \"gbps_per_wg\": %.1f}\n", g,
           vec_per_wg * 16 / 1024.0, best * 1e3, bytes / best / 1e6, bytes / best / 1e6 / g);
  }
  return 0;
}
