// Cost of a device-wide barrier between resident workgroups on one MI355X (monotonic counter, agent-scope
// release/acquire), with and without a small payload handed across it.  Spins are bounded: a workgroup that does not
// see the barrier complete within ~2^22 polls sets an abort flag and every workgroup leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned* abort_flag, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
    }
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(512) void barrier_kernel(unsigned* counter, unsigned* abort_flag, float* data, int iters, int payload) {
  const unsigned nwg = gridDim.x;
  float acc = 0.0f;
  for (int i = 0; i < iters; ++i) {
    if (payload) {   // every workgroup writes 16 floats, reads 16 floats written by another workgroup after the barrier
      if (threadIdx.x < 16) data[(i & 1) * nwg * 16 + blockIdx.x * 16 + threadIdx.x] = (float)(i + blockIdx.x);
    }
    if (!grid_barrier(counter, abort_flag, nwg * (unsigned)(i + 1))) return;
    if (payload && threadIdx.x < 16)
      acc += __builtin_nontemporal_load(&data[(i & 1) * nwg * 16 + ((blockIdx.x + 17) % nwg) * 16 + threadIdx.x]);
  }
  if (payload && threadIdx.x < 16) data[2 * nwg * 16 + blockIdx.x * 16 + threadIdx.x] = acc;
}

int main() {
  unsigned *counter, *abort_flag;
  float* data;
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&abort_flag, 4)); CK(hipMalloc(&data, 3 * 1024 * 16 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 200;
  for (int nwg : {64, 256, 512}) {
    for (int threads : {256, 512}) {
      if (nwg == 512 && threads == 512) continue;   // keep every workgroup resident: <= 2048 threads per CU
      for (int payload : {0, 1}) {
        float best = 1e9f;
        unsigned ab = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemset(counter, 0, 4)); CK(hipMemset(abort_flag, 0, 4));
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(barrier_kernel, dim3(nwg), dim3(threads), 0, 0, counter, abort_flag, data, iters, payload);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
          CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
          if (ab) break;
        }
        printf("{\"wgs\": %d, \"threads\": %d, \"payload\": %d, \"us_per_barrier\": %.2f, \"aborted\": %u}\n", nwg, threads,
               payload, best * 1e3 / iters, ab);
        if (ab) return 1;
      }
    }
  }
  return 0;
}
