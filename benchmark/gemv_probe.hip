// Where does the time of one decode GEMV go?  Builds the production kernel (GEMV_PROBE=0) and three truncated
// variants from the same header and times back-to-back launches over rotating weight copies.
//   hipcc -O3 --offload-arch=gfx950 -DGEMV_PROBE=<n> -I bigdl_amd/csrc/hip benchmark/gemv_probe.hip -o probe<n>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemv_kernel.h"

using namespace bigdl;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static size_t t16_bytes(int n, int k) {   // sym_int4: 4-bit codes + fp16 scale per 32
  const size_t np = (n + 15) / 16 * 16, kp = (k + 127) / 128 * 128;
  return np * kp / 2 + np * (kp / 32) * 2;
}

int main() {
  const int M = getenv("PROBE_M") ? atoi(getenv("PROBE_M")) : 1;          // rows of x (1..16)
  const bool with_norm = getenv("PROBE_NORM") != nullptr;                  // fused RMSNorm prologue as in the decode step
  // bytes of weight copies the launches rotate over: 600 MB defeats every cache, ~150 MB stays in the memory-side
  // cache, ~16 MB stays in the L2s
  const double footprint = (getenv("PROBE_FOOTPRINT_MB") ? atof(getenv("PROBE_FOOTPRINT_MB")) : 600.0) * 1e6;
  struct Shape { const char* name; int n, k; bool gate_up; } shapes[] = {
      {"o", 4096, 4096, false}, {"qkv", 12288, 4096, false}, {"down", 4096, 11008, false}, {"gate_up", 11008, 4096, true}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (auto& sh : shapes) {
    const size_t nb = t16_bytes(sh.n, sh.k);
    const int copies = (int)(footprint / nb) < 2 ? 2 : (int)(footprint / nb);
    std::vector<void*> ws(copies);
    for (auto& w : ws) { CK(hipMalloc(&w, nb)); CK(hipMemset(w, 0x11, nb)); }
    _Float16 *x, *out;
    CK(hipMalloc(&x, (size_t)M * sh.k * 2)); CK(hipMemset(x, 0, (size_t)M * sh.k * 2));
    CK(hipMalloc(&out, (size_t)M * sh.n * 2));
    _Float16* gamma;
    CK(hipMalloc(&gamma, sh.k * 2)); CK(hipMemset(gamma, 0x3c, sh.k * 2));
    GemvArgs a{};
    a.gamma = with_norm ? gamma : nullptr;
    a.x = x; a.out = out; a.M = M; a.N = sh.n; a.K = sh.k; a.ldx = sh.k; a.ldo = sh.n; a.out_type = OUT_F16; a.att_qlen = 1;
    a.slot_xdiv = 1; a.eps = 1e-5f;
    auto launch = [&](int i) {
      a.w = ws[i % copies];
      a.w2 = sh.gate_up ? ws[(i + 1) % copies] : nullptr;
      a.act = sh.gate_up ? 1 : 0;
      if (launch_qt<Q_SYM_INT4>(a, sh.gate_up, st) != 0) { printf("launch failed\n"); exit(1); }
    };
    for (int i = 0; i < 20; ++i) launch(i);
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) launch(i);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("{\"probe\": %d, \"M\": %d, \"norm\": %d, \"shape\": \"%s\", \"us\": %.2f, \"MB\": %.1f, \"copies\": %d}\n", GEMV_PROBE, M,
           (int)with_norm, sh.name, best * 1e3 / iters, nb * (sh.gate_up ? 2 : 1) / 1e6, copies);
    for (auto& w : ws) CK(hipFree(w));
    CK(hipFree(x)); CK(hipFree(out));
  }
  return 0;
}
