// One decode layer of Llama-2-7B (sym_int4 weights, e5m2 KV cache, batch 1) as the production kernel chain
//   qkv GEMV (RMSNorm fused) -> split-KV attention (RoPE + append fused) -> o_proj GEMV (split combine + residual)
//   -> gate/up GEMV (RMSNorm + SiLU*mul) -> down GEMV (residual)
// captured in a hipGraph over L layers with distinct weights and caches (3.9 GB: no cache keeps anything between
// replays), plus one graph per kernel type to attribute the time.  The kernels come straight from the library
// sources, so -D knobs of gemv_kernel.h / attn_decode.hip can be compared build against build:
//   hipcc -O3 --offload-arch=gfx950 [-DGEMV_NW_MAIN=4 ...] -I bigdl_amd/csrc/hip benchmark/layer_probe.hip -o layer_probe
//   ./layer_probe <tag> [ctx]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <functional>
#include <algorithm>
#include "gemv_kernel.h"
#include "attn_decode.hip"

using namespace bigdl;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static size_t t16_bytes(int n, int k) {   // sym_int4: 4-bit codes + fp16 scale per 32
  const size_t np = (n + 15) / 16 * 16, kp = (k + 127) / 128 * 128;
  return np * kp / 2 + np * (kp / 32) * 2;
}

__global__ void fill_codes(uint32_t* p, size_t n, uint32_t seed, uint32_t mask = 0xffffffffu) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = x & mask;
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

// plain streaming read of n_vec 16-byte vectors: every thread keeps UNR loads in flight, grid-stride; the xor of
// everything goes to out (keeps the loads alive).  The floor for "read these bytes once" inside a kernel chain.
template <int UNR, bool NT>
__global__ void stream_read(const uint4v* __restrict__ p, size_t n_vec, uint32_t* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4v acc = {0, 0, 0, 0};
  for (; i + (UNR - 1) * stride < n_vec; i += UNR * stride) {
    uint4v v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc ^= v[u];
  }
  for (; i < n_vec; i += stride) acc ^= p[i];
  const uint32_t x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (x == 0x12345678u) out[0] = x;
}

int main(int argc, char** argv) {
  const char* tag = argc > 1 ? argv[1] : "probe";
  const int ctx = argc > 2 ? atoi(argv[2]) : 1100;
  const int L = 32, H = 4096, I = 11008, Hq = 32, Hkv = 32, D = 128, QKV = 3 * H;
  const int Lmax = (ctx + 256 + 255) / 256 * 256;
  const int splits = 8;
  // PROBE_SAME=1: every layer reads layer 0's weights and cache (what the kernels cost when their bytes are cache-resident)
  const bool same = getenv("PROBE_SAME") != nullptr;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  std::vector<void*> wqkv(L), wo(L), wg(L), wu(L), wd(L), kc(L), vc(L);
  for (int l = 0; l < L; ++l) {
    wqkv[l] = make_weight(QKV, H, 11 * l + 1);
    wo[l] = make_weight(H, H, 11 * l + 2);
    wg[l] = make_weight(I, H, 11 * l + 3);
    wu[l] = make_weight(I, H, 11 * l + 4);
    wd[l] = make_weight(H, I, 11 * l + 5);
    const size_t cb = (size_t)Hkv * Lmax * D;
    CK(hipMalloc(&kc[l], cb)); CK(hipMalloc(&vc[l], cb));
    // e5m2 bytes with the top exponent bit cleared: finite, |x| < 2
    fill_codes<<<512, 256>>>((uint32_t*)kc[l], cb / 4, 7 * l + 3, 0xbfbfbfbfu);
    fill_codes<<<512, 256>>>((uint32_t*)vc[l], cb / 4, 7 * l + 4, 0xbfbfbfbfu);
  }
  _Float16 *h, *qkv, *attn, *act, *gamma;
  CK(hipMalloc(&h, H * 2)); CK(hipMalloc(&qkv, QKV * 2)); CK(hipMalloc(&attn, H * 2)); CK(hipMalloc(&act, I * 2));
  CK(hipMalloc(&gamma, H * 2));
  fill_half<<<16, 256>>>(h, H, -1.0f, 1.0f, 99);
  fill_half<<<16, 256>>>(gamma, H, 0.9f, 1.1f, 98);
  float* rope;
  const int max_pos = 4096;
  CK(hipMalloc(&rope, (size_t)max_pos * (D / 2) * 2 * 4));
  {
    std::vector<float> t((size_t)max_pos * (D / 2) * 2);
    for (int p = 0; p < max_pos; ++p)
      for (int i = 0; i < D / 2; ++i) {
        const double f = p * pow(10000.0, -2.0 * i / D);
        t[((size_t)p * (D / 2) + i) * 2] = (float)cos(f);
        t[((size_t)p * (D / 2) + i) * 2 + 1] = (float)sin(f);
      }
    CK(hipMemcpy(rope, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  }
  int* past;
  CK(hipMalloc(&past, 4));
  CK(hipMemcpy(past, &ctx, 4, hipMemcpyHostToDevice));
  float* ws;
  const int64_t wsf = bigdl_attn_decode_ws_floats(1, Hq, Hkv, 1, D, splits);
  CK(hipMalloc(&ws, (wsf + 16) * 4)); CK(hipMemset(ws, 0, (wsf + 16) * 4));
  int* counters;
  CK(hipMalloc(&counters, 4 * 64)); CK(hipMemset(counters, 0, 4 * 64));
  CK(hipDeviceSynchronize());

  auto base_args = [&]() {
    GemvArgs a{};
    a.M = 1; a.out_type = OUT_F16; a.att_qlen = 1; a.slot_xdiv = 1; a.eps = 1e-5f;
    return a;
  };
  auto k_qkv = [&](int l) {
    GemvArgs a = base_args();
    a.x = h; a.w = wqkv[l]; a.out = qkv; a.gamma = gamma; a.N = QKV; a.K = H; a.ldx = H; a.ldo = QKV;
    if (launch_qt<Q_SYM_INT4>(a, false, st) != 0) { printf("qkv launch failed\n"); exit(1); }
  };
  auto k_attn = [&](int l) {
    const int rc = bigdl_attn_decode(qkv, kc[l], vc[l], attn, ws, counters, past, nullptr, 0, nullptr, rope, nullptr, max_pos, D, 0,
                                     0.088388f, 0.0f, 1, Hq, Hkv, 1, 0, 1, QKV, Lmax, D, splits, 0, 1, 0, 1, nullptr, nullptr, st);
    if (rc != 0) { printf("attn launch failed %d\n", rc); exit(1); }
  };
  auto k_o = [&](int l) {
    GemvArgs a = base_args();
    a.x = attn; a.w = wo[l]; a.out = h; a.residual = h; a.N = H; a.K = H; a.ldx = H; a.ldo = H;
    a.att_ws = ws; a.att_splits = splits; a.att_D = D; a.att_G = 1; a.att_Hkv = Hkv; a.att_rows_alloc = 16;
    if (launch_qt<Q_SYM_INT4>(a, false, st) != 0) { printf("o launch failed\n"); exit(1); }
  };
  auto k_gu = [&](int l) {
    GemvArgs a = base_args();
    a.x = h; a.w = wg[l]; a.w2 = wu[l]; a.out = act; a.gamma = gamma; a.N = I; a.K = H; a.ldx = H; a.ldo = I; a.act = ACT_SILU;
    if (launch_qt<Q_SYM_INT4>(a, true, st) != 0) { printf("gate_up launch failed\n"); exit(1); }
  };
  auto k_down = [&](int l) {
    GemvArgs a = base_args();
    a.x = act; a.w = wd[l]; a.out = h; a.residual = h; a.N = H; a.K = I; a.ldx = I; a.ldo = H;
    if (launch_qt<Q_SYM_INT4>(a, false, st) != 0) { printf("down launch failed\n"); exit(1); }
  };
  auto time_graph = [&](const std::function<void(int)>& body, int nodes_per_layer) {
    // warm-up outside capture (module load), then capture L layers and replay
    body(0);
    CK(hipStreamSynchronize(st));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; ++l) body(same ? 0 : l);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    const int reps = 10;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best * 1e3f / reps / L;      // us per layer
  };
#if GEMV_STAMP
  {
    // phase timeline of one cold launch per kernel type (preceded by another kernel, as in the decode chain)
    unsigned long long* sbuf;
    const size_t nst = (size_t)1024 * 16 * 8;
    CK(hipMalloc(&sbuf, nst * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemv_stamps), &sbuf, sizeof(sbuf)));
    struct { const char* name; std::function<void(int)> pre, run; int wgs, nw; } ks[] = {
        {"qkv", k_down, k_qkv, QKV / 16, 8}, {"gate_up", k_o, k_gu, I / 16, 8}, {"down", k_gu, k_down, H / 16, 16}, {"o", k_attn, k_o, H / 16, 8}};
    for (auto& k : ks) {
      std::vector<unsigned long long> hs(nst);
      for (int rep = 0; rep < 3; ++rep) {
        k.pre(3 + rep);
        CK(hipMemsetAsync(sbuf, 0, nst * 8, st));
        k.run(7 + rep);
        CK(hipStreamSynchronize(st));
      }
      CK(hipMemcpy(hs.data(), sbuf, nst * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull;
      const int nwv = k.wgs * k.nw;
      for (int w = 0; w < nwv; ++w) if (hs[w * 8] && hs[w * 8] < t0) t0 = hs[w * 8];
      printf("{\"tag\": \"%s\", \"kernel\": \"%s\", \"waves\": %d", tag, k.name, nwv);
      const char* names[6] = {"entry", "issued", "staged", "barrier", "loop_done", "exit"};
      for (int i = 0; i < 6; ++i) {
        std::vector<double> v;
        for (int w = 0; w < nwv; ++w) if (hs[w * 8 + i]) v.push_back((double)(hs[w * 8 + i] - t0) * 0.01);
        std::sort(v.begin(), v.end());
        if (v.empty()) continue;
        printf(", \"%s\": [%.2f, %.2f, %.2f, %.2f]", names[i], v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back());
      }
      printf("}\n");
    }
    return 0;
  }
#endif
  if (getenv("PROBE_STREAM")) {
    uint32_t* sink;
    CK(hipMalloc(&sink, 64));
    struct Cfg { const char* name; int grid, block, unr; bool nt; } cfgs[] = {
        {"768x512_u4_nt", 768, 512, 4, true}, {"768x512_u4", 768, 512, 4, false}, {"1536x256_u4_nt", 1536, 256, 4, true},
        {"2048x256_u2_nt", 2048, 256, 2, true}, {"512x512_u8_nt", 512, 512, 8, true}, {"256x1024_u4_nt", 256, 1024, 4, true},
        {"1024x256_u8_nt", 1024, 256, 8, true}, {"256x256_u8_nt", 256, 256, 8, true}};
    struct Sz { const char* name; std::vector<void*>* w; size_t bytes; } szs[] = {
        {"o_9MB", &wo, t16_bytes(H, H)}, {"qkv_28MB", &wqkv, t16_bytes(QKV, H)}, {"down_25MB", &wd, t16_bytes(H, I)}};
    for (auto& z : szs)
      for (auto& c : cfgs) {
        const size_t nv = z.bytes / 16;
        auto body = [&](int l) {
          const uint4v* p = (const uint4v*)(*z.w)[l];
          if (c.unr == 2) hipLaunchKernelGGL((stream_read<2, true>), dim3(c.grid), dim3(c.block), 0, st, p, nv, sink);
          else if (c.unr == 8) hipLaunchKernelGGL((stream_read<8, true>), dim3(c.grid), dim3(c.block), 0, st, p, nv, sink);
          else if (c.nt) hipLaunchKernelGGL((stream_read<4, true>), dim3(c.grid), dim3(c.block), 0, st, p, nv, sink);
          else hipLaunchKernelGGL((stream_read<4, false>), dim3(c.grid), dim3(c.block), 0, st, p, nv, sink);
        };
        const float t = time_graph(body, 1);
        printf("{\"stream\": \"%s\", \"cfg\": \"%s\", \"us\": %.2f, \"TBps_minus_1.78\": %.2f}\n", z.name, c.name, t,
               z.bytes / (t - 1.78) * 1e-6);
      }
    return 0;
  }
  if (getenv("PROBE_TWO_STREAMS")) {
    // optimistic bound for cross-kernel overlap: the chain on two alternating streams with NO dependency between
    // neighbours (results are meaningless): kernel i + 1 streams its weights while kernel i drains
    hipStream_t s2;
    CK(hipStreamCreate(&s2));
    hipStream_t s1 = st;
    auto chain = [&](bool with_attn) {
      int n = 0;
      for (int l = 0; l < L; ++l) {
        st = (n++ & 1) ? s2 : s1; k_qkv(l);
        if (with_attn) { st = (n++ & 1) ? s2 : s1; k_attn(l); }
        st = (n++ & 1) ? s2 : s1; k_o(l);
        st = (n++ & 1) ? s2 : s1; k_gu(l);
        st = (n++ & 1) ? s2 : s1; k_down(l);
      }
      st = s1;
    };
    for (int with_attn = 0; with_attn < 2; ++with_attn) {
      hipEvent_t e0, e1, ej;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ej));
      float best = 1e9f;
      for (int r = 0; r < 6; ++r) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s1));
        CK(hipStreamWaitEvent(s2, e0, 0));
        chain(with_attn);
        CK(hipEventRecord(ej, s2));
        CK(hipStreamWaitEvent(s1, ej, 0));
        CK(hipEventRecord(e1, s1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("{\"tag\": \"%s\", \"two_streams\": 1, \"with_attn\": %d, \"layer_us\": %.2f}\n", tag, with_attn, best * 1e3f / L);
    }
    // the same chain on one stream, eager, for reference
    for (int with_attn = 0; with_attn < 2; ++with_attn) {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float best = 1e9f;
      for (int r = 0; r < 6; ++r) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s1));
        for (int l = 0; l < L; ++l) { k_qkv(l); if (with_attn) k_attn(l); k_o(l); k_gu(l); k_down(l); }
        CK(hipEventRecord(e1, s1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("{\"tag\": \"%s\", \"two_streams\": 0, \"with_attn\": %d, \"layer_us\": %.2f}\n", tag, with_attn, best * 1e3f / L);
    }
    return 0;
  }
  const float t_layer = time_graph([&](int l) { k_qkv(l); k_attn(l); k_o(l); k_gu(l); k_down(l); }, 5);
  const float t_qkv = time_graph([&](int l) { k_qkv(l); }, 1);
  const float t_attn = time_graph([&](int l) { k_attn(l); }, 1);
  const float t_o = time_graph([&](int l) { k_o(l); }, 1);
  const float t_gu = time_graph([&](int l) { k_gu(l); }, 1);
  const float t_down = time_graph([&](int l) { k_down(l); }, 1);
  printf("{\"tag\": \"%s\", \"ctx\": %d, \"layer_us\": %.2f, \"qkv\": %.2f, \"attn\": %.2f, \"o\": %.2f, \"gate_up\": %.2f, \"down\": %.2f, "
         "\"sum\": %.2f}\n", tag, ctx, t_layer, t_qkv, t_attn, t_o, t_gu, t_down, t_qkv + t_attn + t_o + t_gu + t_down);
  return 0;
}
