// Sustained HBM read rate of plain streaming kernels on one GPU: 1 GiB read once per launch, swept over workgroup
// count, workgroup size and 16-byte loads in flight per lane; non-temporal vs plain loads.  What a decode GEMV could
// reach if it were nothing but its weight stream.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ void pull(const uint4v* __restrict__ src, unsigned* out, size_t nvec) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  uint4v acc = {0, 0, 0, 0};
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < nvec; i += UNROLL * stride) {
    uint4v v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}
template <int UNROLL, bool NT>
float run(const uint4v* src, unsigned* out, size_t nvec, int grid, int block) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((pull<UNROLL, NT>), dim3(grid), dim3(block), 0, 0, src, out, nvec);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best;
}
int main() {
  const size_t bytes = 1ull << 30, nvec = bytes / 16;
  uint4v* src; unsigned* out;
  (void)hipMalloc(&src, bytes); (void)hipMalloc(&out, 64);
  (void)hipMemset(src, 1, bytes);
  const int grids[] = {512, 1024, 2048, 4096, 8192};
  const int blocks[] = {256, 512, 1024};
  for (int b : blocks)
    for (int g : grids) {
      const float t4 = run<4, true>(src, out, nvec, g, b), t8 = run<8, true>(src, out, nvec, g, b), t16 = run<16, true>(src, out, nvec, g, b);
      const float p8 = run<8, false>(src, out, nvec, g, b);
      printf("{\"block\": %d, \"grid\": %d, \"TBps_nt_u4\": %.2f, \"TBps_nt_u8\": %.2f, \"TBps_nt_u16\": %.2f, \"TBps_plain_u8\": %.2f}\n", b, g,
             bytes / t4 / 1e9, bytes / t8 / 1e9, bytes / t16 / 1e9, bytes / p8 / 1e9);
    }
  return 0;
}
