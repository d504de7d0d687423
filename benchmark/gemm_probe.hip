// The fused low-bit GEMM at decode batch sizes (17 .. 256 rows) on the Llama-2-7B projection shapes, timed the way a
// decode step runs it: a hipGraph over L launches with DISTINCT weights (3.5 GB in total: nothing stays in a cache
// between launches), split-K reduce included.  The kernel comes straight from the library sources, so -D knobs and
// template variants of gemm_kernel.h can be compared build against build:
//   hipcc -O3 --offload-arch=gfx950 -I bigdl_amd/csrc/hip benchmark/gemm_probe.hip -o gemm_probe
//   ./gemm_probe <tag> [rows] [unused] [target workgroups]
// -DGEMM_STAMP=1 prints the phase timeline of the wavefronts, -DGEMM_ABL=<bits> removes parts of the k loop (gemm_kernel.h).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <functional>
#include <algorithm>
#include "gemm_kernel.h"

using namespace bigdl;

#ifndef PROBE_NT
#define PROBE_NT 2          // 16-column weight tiles per wavefront: column block = 64 * PROBE_NT
#endif
#ifndef PROBE_NW
#define PROBE_NW 4          // wavefronts per workgroup that multiply
#endif
#define PROBE_BN (16 * PROBE_NW * PROBE_NT)
#ifndef PROBE_LW
#define PROBE_LW 0          // 1: x tiles brought in by a loader wavefront (gemm_kernel.h, LW)
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static size_t t16_bytes(int n, int k) {   // sym_int4: 4-bit codes + fp16 scale per 32
  const size_t np = (n + 15) / 16 * 16, kp = (k + 127) / 128 * 128;
  return np * kp / 2 + np * (kp / 32) * 2;
}
__global__ void fill_codes(uint32_t* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = x;
  }
}
__global__ void fill_half(_Float16* p, size_t n, float lo, float hi, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (_Float16)(lo + (hi - lo) * (float)(x & 0xffff) / 65535.0f);
  }
}
static void* make_weight(int n, int k, uint32_t seed) {
  const size_t nb = t16_bytes(n, k);
  const size_t np = (n + 15) / 16 * 16, kp = (k + 127) / 128 * 128;
  void* w;
  CK(hipMalloc(&w, nb));
  fill_codes<<<1024, 256>>>((uint32_t*)w, np * kp / 8, seed);
  fill_half<<<256, 256>>>((_Float16*)((char*)w + np * kp / 2), np * (kp / 32), 0.001f, 0.004f, seed + 1);
  return w;
}

__global__ __launch_bounds__(256) void reduce_kernel(GemmArgs a) {
  const int64_t total = (int64_t)a.M * a.N;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float v = 0.0f;
  for (int s = 0; s < a.splitk; ++s) v += a.partial[(int64_t)s * total + i];
  const int64_t m = i / a.N;
  ((_Float16*)a.out)[m * a.ldo + (i - m * a.N)] = (_Float16)v;
}

int main(int argc, char** argv) {
  const char* tag = argc > 1 ? argv[1] : "probe";
  const int M = argc > 2 ? atoi(argv[2]) : 64;
  const int kg = argc > 3 ? atoi(argv[3]) : 1;
  const int target = argc > 4 ? atoi(argv[4]) : 256;
  const int L = 32;
  struct Shape { const char* name; int n, k; } shapes[] = {{"qkv", 12288, 4096}, {"o", 4096, 4096}, {"gate", 11008, 4096}, {"down", 4096, 11008}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  _Float16 *x, *out, *ref;
  float* ws;
  CK(hipMalloc(&x, (size_t)M * 11008 * 2));
  CK(hipMalloc(&out, (size_t)M * 12288 * 2));
  CK(hipMalloc(&ref, (size_t)M * 12288 * 2));
  CK(hipMalloc(&ws, (size_t)64 * M * 12288 * 4));
  fill_half<<<256, 256>>>(x, (size_t)M * 11008, -1.0f, 1.0f, 7);
  double total_us = 0.0;
#if GEMM_STAMP
  unsigned long long* sbuf;
  const size_t nst = (size_t)2048 * 16 * 32;
  CK(hipMalloc(&sbuf, nst * 8));
  CK(hipMemset(sbuf, 0, nst * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), &sbuf, sizeof(sbuf)));
  CK(hipDeviceSynchronize());
#endif
  for (auto& sh : shapes) {
    std::vector<void*> w(L);
    for (int l = 0; l < L; ++l) w[l] = make_weight(sh.n, sh.k, 13 * l + 5);
    CK(hipDeviceSynchronize());
    const int KT = (sh.k + 127) / 128;
    auto args = [&](int l, int sk_mult_of) {
      GemmArgs a = {};
      a.x = x; a.w = w[l]; a.out = out; a.partial = ws; a.M = M; a.N = sh.n; a.K = sh.k; a.ldx = sh.k; a.ldo = sh.n; a.out_type = OUT_F16;
      const int mblocks = (M + 63) / 64;
      const int blocks = ((sh.n + PROBE_BN - 1) / PROBE_BN) * mblocks;
      int sk = 1;
      if (blocks < target * 3 / 4) sk = (target + blocks - 1) / blocks;
      if (getenv("PROBE_SK_FLOOR")) sk = std::max(1, target / blocks);      // never more workgroups than the target
      if (sk > KT / 2) sk = KT / 2 > 0 ? KT / 2 : 1;
      int per = (KT + sk - 1) / sk;
      per = (per + sk_mult_of - 1) / sk_mult_of * sk_mult_of;
      a.kt_per_split = per;
      a.splitk = (KT + per - 1) / per;
      return a;
    };
    auto launch = [&](int l, int use_kg) {
      GemmArgs a = args(l, 1);
      const int mblocks = (M + 63) / 64;
      int rc;
      (void)use_kg;
      rc = gemm_launch_mt<Q_SYM_INT4, 4, PROBE_NT, false, PROBE_NW, (PROBE_LW != 0)>(a, mblocks, st);
      if (rc != 0) { printf("launch failed %d\n", rc); exit(1); }
      if (a.splitk > 1) hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)(((int64_t)M * sh.n + 255) / 256)), dim3(256), 0, st, a);
      return a;
    };
    // numerics: the variant against the production form on the same weight
    GemmArgs a1 = launch(0, 1);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(ref, out, (size_t)M * sh.n * 2, hipMemcpyDeviceToDevice));
    GemmArgs a2 = launch(0, kg);
    CK(hipStreamSynchronize(st));
    std::vector<_Float16> h1((size_t)M * sh.n), h2((size_t)M * sh.n);
    CK(hipMemcpy(h1.data(), ref, h1.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), out, h2.size() * 2, hipMemcpyDeviceToHost));
    double dmax = 0, vmax = 0;
    for (size_t i = 0; i < h1.size(); ++i) { dmax = std::max(dmax, std::fabs((double)h1[i] - (double)h2[i])); vmax = std::max(vmax, std::fabs((double)h1[i])); }
#if GEMM_STAMP
    {
      for (int rep = 0; rep < 3; ++rep) {
        launch(5 + rep, kg);               // a predecessor, as in the decode chain
        CK(hipMemsetAsync(sbuf, 0, nst * 8, st));
        launch(9 + rep, kg);
        CK(hipStreamSynchronize(st));
      }
      std::vector<unsigned long long> hs(nst);
      CK(hipMemcpy(hs.data(), sbuf, nst * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull;
      const int nwv = a2.splitk * ((sh.n + PROBE_BN - 1) / PROBE_BN) * 16;
      for (int w = 0; w < nwv; ++w) if (hs[(size_t)w * 32] && hs[(size_t)w * 32] < t0) t0 = hs[(size_t)w * 32];
      printf("{\"tag\": \"%s\", \"shape\": \"%s\", \"stamps_us [min, median, p90, max]\": {", tag, sh.name);
      for (int i = 0; i < 32; ++i) {
        std::vector<double> v;
        for (int w = 0; w < nwv; ++w) if (hs[(size_t)w * 32 + i]) v.push_back((double)(hs[(size_t)w * 32 + i] - t0) * 0.01);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("\"%d\": [%.2f, %.2f, %.2f, %.2f], ", i, v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back());
      }
      printf("\"waves\": %d}}\n", nwv);
    }
#endif
#if GEMM_ABL & 32
    {   // where the workgroups of one launch ran: distinct (XCC, SE, CU) and the largest number sharing a CU
      GemmArgs aw = launch(3, kg);
      CK(hipStreamSynchronize(st));
      const int nwg = ((sh.n + PROBE_BN - 1) / PROBE_BN) * aw.splitk * ((M + 63) / 64);
      std::vector<uint32_t> hw(2 * (size_t)nwg);
      CK(hipMemcpy(hw.data(), ws, hw.size() * 4, hipMemcpyDeviceToHost));
      std::vector<int> cnt(1 << 16, 0);
      int distinct = 0, worst = 0;
      for (int i = 0; i < nwg; ++i) {
        const uint32_t h = hw[2 * i], x = hw[2 * i + 1] & 0xf;
        const int cu = (h >> 8) & 0xf, sh_ = (h >> 12) & 0x1, se = (h >> 13) & 0x7;
        const int key = (int)(x << 8 | se << 5 | sh_ << 4 | cu);
        if (cnt[key]++ == 0) ++distinct;
        worst = std::max(worst, cnt[key]);
      }
      printf("{\"tag\": \"%s\", \"shape\": \"%s\", \"workgroups\": %d, \"distinct_cus\": %d, \"max_per_cu\": %d, \"hw_id0\": \"%08x\", \"xcc0\": \"%08x\"}\n",
             tag, sh.name, nwg, distinct, worst, hw[0], hw[1]);
    }
#endif
    // timing
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; ++l) launch(l, kg);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    const double us = best * 1e3 / 5 / L;
    total_us += us * (strcmp(sh.name, "gate") == 0 ? 2 : 1);
    printf("{\"tag\": \"%s\", \"shape\": \"%s\", \"M\": %d, \"kg\": %d, \"splitk\": %d, \"kt_per_split\": %d, \"workgroups\": %d, \"us\": %.2f, "
           "\"weight_TBps\": %.2f, \"maxdiff_vs_kg1\": %.4g, \"max_abs\": %.4g, \"kg1_splitk\": %d}\n", tag, sh.name, M, kg, a2.splitk,
           a2.kt_per_split, ((sh.n + PROBE_BN - 1) / PROBE_BN) * a2.splitk * ((M + 63) / 64), us, t16_bytes(sh.n, sh.k) / us * 1e-6, dmax, vmax, a1.splitk);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    for (int l = 0; l < L; ++l) CK(hipFree(w[l]));
  }
  printf("{\"tag\": \"%s\", \"M\": %d, \"kg\": %d, \"layer_gemms_us\": %.2f}\n", tag, M, kg, total_us);
  return 0;
}
