// One-shot all-reduce for the latency-bound collectives of a tensor-parallel decode step (fp16 [tokens, hidden],
// 8-64 KiB, two per layer), written for xGMI: every GPU stores its vector straight into a slot of every peer's
// buffer (all 7 links busy at once, one hop), publishes a flag, waits for the peers' flags on its own buffer and
// sums the slots locally in rank order (bitwise identical result on every rank).  A ring all-reduce would use two of
// the seven links and 2 (W - 1) dependent hops for the same bytes.
//
// The reference has no counterpart in its tree: its tensor parallelism calls oneCCL's all_reduce after every
// row-parallel linear (python/llm/src/ipex_llm/transformers/low_bit_linear.py:675-682) and ships patched oneCCL
// builds for latency (docker/llm/serving/xpu/docker/ccl_torch.patch).
//
// Protocol (per rank: a fine-grained, uncached device buffer mapped into every peer through hipIpc):
//   data  [2 sets][W source slots][max_bytes]      flags [2 sets][W] (one 64-byte line each)
//   call e (epoch counter in local device memory, so hipGraph replays advance it): set = e & 1
//     1. store my vector into slot[me] of every peer (and my own buffer), system-scope fence
//     2. release-store e into flag[set][me] of every peer
//     3. spin (bounded) until flag[set][s] == e for every source s on my buffer
//     4. out = sum over s of slot[s]
//   Two sets are enough: a rank can only start call e + 2 after it finished e + 1, which needs every peer's flag for
//   e + 1, which a peer publishes only after it finished reading the slots of call e.
// The spin is bounded (about a second): on expiry the kernel raises the error word and leaves, so a peer that died
// cannot hang the GPU.
#include "common.h"
#include <string.h>

namespace bigdl {

constexpr int AR_MAX_WORLD = 16;
constexpr int AR_FLAG_STRIDE = 16;          // uint32 per flag: one 64-byte line each
constexpr int AR_THREADS = 1024;

struct ArArgs {
  uint8_t* peers[AR_MAX_WORLD];   // comm buffer of every rank, rank order (own buffer at [rank])
  const _Float16* in;
  _Float16* out;
  uint32_t* epoch;                // local device counter
  int* error;                     // local device error word
  int64_t max_bytes;              // slot size
  int n;                          // elements (multiple of 8)
  int rank, world;
};

__device__ __forceinline__ uint8_t* ar_slot(uint8_t* base, int64_t max_bytes, int world, int set, int src) {
  return base + ((int64_t)set * world + src) * max_bytes;
}
__device__ __forceinline__ uint32_t* ar_flag(uint8_t* base, int64_t max_bytes, int world, int set, int src) {
  uint32_t* flags = (uint32_t*)(base + 2 * (int64_t)world * max_bytes);
  return flags + ((int64_t)set * world + src) * AR_FLAG_STRIDE;
}

__global__ __launch_bounds__(AR_THREADS) void allreduce_oneshot_kernel(ArArgs a) {
  __shared__ uint32_t ep_s;
  if (threadIdx.x == 0) {
    const uint32_t e = *a.epoch + 1;
    *a.epoch = e;
    ep_s = e;
  }
  __syncthreads();
  const uint32_t ep = ep_s;
  const int set = ep & 1;
  const int nvec = a.n >> 3;
  // 1. push
  for (int v = threadIdx.x; v < nvec; v += AR_THREADS) {
    const uint4v val = *(const uint4v*)(a.in + 8 * (int64_t)v);
    for (int p = 0; p < a.world; ++p)
      *(uint4v*)(ar_slot(a.peers[p], a.max_bytes, a.world, set, a.rank) + 16 * (int64_t)v) = val;
  }
  __threadfence_system();
  __syncthreads();
  // 2. publish, 3. wait
  if (threadIdx.x < a.world) {
    __hip_atomic_store(ar_flag(a.peers[threadIdx.x], a.max_bytes, a.world, set, a.rank), ep, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t* mine = ar_flag(a.peers[a.rank], a.max_bytes, a.world, set, threadIdx.x);
    long long spins = 0;
    while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != ep) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > 20000000ll) {          // about a second: give up instead of hanging the queue
        *a.error = 1;
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  // 4. reduce in rank order
  uint8_t* self = a.peers[a.rank];
  for (int v = threadIdx.x; v < nvec; v += AR_THREADS) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < a.world; ++s) {
      const half8 h = *(const half8*)(ar_slot(self, a.max_bytes, a.world, set, s) + 16 * (int64_t)v);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += (float)h[q];
    }
    half8 o;
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = (_Float16)acc[q];
    *(half8*)(a.out + 8 * (int64_t)v) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// Two-shot all-reduce for the large messages of a tensor-parallel prompt pass (fp16 [tokens, hidden], 1-64 MiB):
// reduce-scatter then all-gather, both as direct stores to every peer, so each GPU moves 2 (W-1)/W of the message
// over seven links at once instead of pushing all of it W-1 times (one-shot) or around a two-link ring.
//   rank r owns chunk r (1/W of the vector).
//   1. every rank stores chunk p of its input into scatter[me] of peer p; flag; wait for all sources
//   2. every rank sums the W copies of its chunk (rank order: identical bits everywhere) and stores the sum into
//      result[chunk me] of every peer; flag; wait for all chunks
//   3. out = result
// AR2_WGS workgroups work on disjoint slices of every chunk; workgroup b only ever synchronizes with workgroup b of
// the peers (own flag lines, own epoch counter), so there is no grid-wide barrier.  The grid (64 workgroups) is far
// below what the GPU keeps resident, and the spin is bounded as above.
// Buffer per rank: scatter [2 sets][W][max_bytes / W] | result [2 sets][max_bytes] | flags [2 phases][2 sets][W][AR2_WGS]
// ---------------------------------------------------------------------------------------------
constexpr int AR2_WGS = 64;
constexpr int AR2_THREADS = 512;

struct Ar2Args {
  uint8_t* peers[AR_MAX_WORLD];
  const _Float16* in;
  _Float16* out;
  uint32_t* epoch;                // [AR2_WGS] local device counters
  int* error;
  int64_t max_bytes;              // multiple of 64 * world
  int n;                          // elements (multiple of 8)
  int rank, world;
};

__device__ __forceinline__ uint32_t* ar2_flag(uint8_t* base, int64_t max_bytes, int world, int phase, int set, int src, int wg) {
  uint32_t* flags = (uint32_t*)(base + 4 * max_bytes);
  return flags + ((((int64_t)phase * 2 + set) * world + src) * AR2_WGS + wg) * AR_FLAG_STRIDE;
}

__device__ __forceinline__ void ar2_sync(const Ar2Args& a, int phase, int set, uint32_t ep) {
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x < a.world) {
    __hip_atomic_store(ar2_flag(a.peers[threadIdx.x], a.max_bytes, a.world, phase, set, a.rank, blockIdx.x), ep,
                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t* mine = ar2_flag(a.peers[a.rank], a.max_bytes, a.world, phase, set, threadIdx.x, blockIdx.x);
    long long spins = 0;
    while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != ep) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > 20000000ll) {
        *a.error = 1;
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

__global__ __launch_bounds__(AR2_THREADS) void allreduce_twoshot_kernel(Ar2Args a) {
  __shared__ uint32_t ep_s;
  if (threadIdx.x == 0) {
    const uint32_t e = a.epoch[blockIdx.x] + 1;
    a.epoch[blockIdx.x] = e;
    ep_s = e;
  }
  __syncthreads();
  const uint32_t ep = ep_s;
  const int set = ep & 1;
  const int W = a.world;
  const int64_t nvec = a.n >> 3;
  const int64_t cvec = (nvec + W - 1) / W;                      // vectors per chunk
  const int64_t svec = (cvec + AR2_WGS - 1) / AR2_WGS;          // vectors per workgroup slice of a chunk
  const int64_t v0 = (int64_t)blockIdx.x * svec, v1 = min(cvec, v0 + svec);
  const int64_t chunk_cap = a.max_bytes / W;
  auto scatter = [&](uint8_t* base, int src) { return base + ((int64_t)set * W + src) * chunk_cap; };
  auto result = [&](uint8_t* base) { return base + 2 * a.max_bytes + (int64_t)set * a.max_bytes; };

  // 1. chunk p of my input -> scatter[me] on peer p
  for (int p = 0; p < W; ++p) {
    uint8_t* dst = scatter(a.peers[p], a.rank);
    for (int64_t v = v0 + threadIdx.x; v < v1; v += AR2_THREADS) {
      const int64_t gv = (int64_t)p * cvec + v;
      if (gv < nvec) *(uint4v*)(dst + 16 * v) = *(const uint4v*)(a.in + 8 * gv);
    }
  }
  ar2_sync(a, 0, set, ep);
  // 2. reduce my chunk in rank order, store the sum into every peer's result
  uint8_t* self = a.peers[a.rank];
  for (int64_t v = v0 + threadIdx.x; v < v1; v += AR2_THREADS) {
    const int64_t gv = (int64_t)a.rank * cvec + v;
    if (gv >= nvec) continue;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < W; ++s) {
      const half8 h = *(const half8*)(scatter(self, s) + 16 * v);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += (float)h[q];
    }
    half8 o;
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = (_Float16)acc[q];
    for (int p = 0; p < W; ++p) *(half8*)(result(a.peers[p]) + 16 * gv) = o;
  }
  ar2_sync(a, 1, set, ep);
  // 3. result -> out (every chunk's slice of this workgroup)
  const uint8_t* res = result(self);
  for (int c = 0; c < W; ++c)
    for (int64_t v = v0 + threadIdx.x; v < v1; v += AR2_THREADS) {
      const int64_t gv = (int64_t)c * cvec + v;
      if (gv < nvec) *(uint4v*)(a.out + 8 * gv) = *(const uint4v*)(res + 16 * gv);
    }
}

}  // namespace bigdl

using namespace bigdl;

extern "C" int64_t bigdl_ar_buffer_bytes(int world, int64_t max_bytes) {
  return 2 * (int64_t)world * max_bytes + 2 * (int64_t)world * AR_FLAG_STRIDE * 4;
}

// Fine-grained uncached device memory (stores from peers become visible inside a running kernel), zero-filled.
extern "C" int bigdl_ar_alloc(int64_t bytes, void** ptr) {
  if (hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocUncached) != hipSuccess) return -1;
  if (hipMemset(*ptr, 0, (size_t)bytes) != hipSuccess) return -2;
  if (hipDeviceSynchronize() != hipSuccess) return -3;
  return 0;
}
extern "C" int bigdl_ar_free(void* ptr) { return hipFree(ptr) == hipSuccess ? 0 : -1; }
extern "C" int bigdl_ar_ipc_handle(void* ptr, void* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  return hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, ptr) == hipSuccess ? 0 : -1;
}
extern "C" int bigdl_ar_ipc_open(const void* handle64, void** ptr) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  return hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? 0 : -1;
}
extern "C" int bigdl_ar_ipc_close(void* ptr) { return hipIpcCloseMemHandle(ptr) == hipSuccess ? 0 : -1; }

// peers: host array of `world` buffer addresses in rank order.  in / out: fp16, n % 8 == 0, n * 2 <= max_bytes,
// 16-byte aligned; in == out is allowed.
extern "C" int bigdl_allreduce_oneshot(const int64_t* peers, int rank, int world, const void* in, void* out, int n,
                                       int64_t max_bytes, void* epoch, void* error, hipStream_t stream) {
  if (world < 2 || world > AR_MAX_WORLD || rank < 0 || rank >= world || n <= 0 || (n & 7) || (int64_t)n * 2 > max_bytes ||
      (max_bytes & 63) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15) || epoch == nullptr || error == nullptr)
    return -1;
  ArArgs a;
  for (int i = 0; i < AR_MAX_WORLD; ++i) a.peers[i] = i < world ? (uint8_t*)peers[i] : nullptr;
  for (int i = 0; i < world; ++i)
    if (a.peers[i] == nullptr) return -1;
  a.in = (const _Float16*)in; a.out = (_Float16*)out; a.epoch = (uint32_t*)epoch; a.error = (int*)error;
  a.max_bytes = max_bytes; a.n = n; a.rank = rank; a.world = world;
  hipLaunchKernelGGL(allreduce_oneshot_kernel, dim3(1), dim3(AR_THREADS), 0, stream, a);
  return HIP_CHECK_LAUNCH();
}

extern "C" int64_t bigdl_ar2_buffer_bytes(int world, int64_t max_bytes) {
  return 4 * max_bytes + 2 * 2 * (int64_t)world * AR2_WGS * AR_FLAG_STRIDE * 4;
}
extern "C" int bigdl_ar2_epoch_words() { return AR2_WGS; }

// Two-shot all-reduce (see the kernel).  max_bytes: multiple of 64 * world; epoch: AR2_WGS zero-initialized uint32.
extern "C" int bigdl_allreduce_twoshot(const int64_t* peers, int rank, int world, const void* in, void* out, int64_t n,
                                       int64_t max_bytes, void* epoch, void* error, hipStream_t stream) {
  if (world < 2 || world > AR_MAX_WORLD || rank < 0 || rank >= world || n <= 0 || (n & 7) || n * 2 > max_bytes ||
      n > 0x7fffffff || (max_bytes % (64 * world)) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15) || epoch == nullptr ||
      error == nullptr)
    return -1;
  // chunk capacity must hold ceil(nvec / world) vectors
  if (((n / 8 + world - 1) / world) * 16 > max_bytes / world) return -1;
  Ar2Args a;
  for (int i = 0; i < AR_MAX_WORLD; ++i) a.peers[i] = i < world ? (uint8_t*)peers[i] : nullptr;
  for (int i = 0; i < world; ++i)
    if (a.peers[i] == nullptr) return -1;
  a.in = (const _Float16*)in; a.out = (_Float16*)out; a.epoch = (uint32_t*)epoch; a.error = (int*)error;
  a.max_bytes = max_bytes; a.n = (int)n; a.rank = rank; a.world = world;
  hipLaunchKernelGGL(allreduce_twoshot_kernel, dim3(AR2_WGS), dim3(AR2_THREADS), 0, stream, a);
  return HIP_CHECK_LAUNCH();
}
