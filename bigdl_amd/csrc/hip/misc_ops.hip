// Memory-bound helper kernels for gfx950: RMSNorm / LayerNorm (+fused residual add), SiLU/GELU*mul,
// row softmax, RoPE on strided [B,H,S,D] views, fp8 KV quantize/dequantize, repetition penalty,
// argmax.  They replace the reference's closed xe_addons.{rms_norm, layer_norm, mlp_silu_mul_inplaced,
// attn_softmax_inplaced, rotary_*_inplaced, quantize_key_value, dequantize_key_value,
// repetition_penalty_logits_process_inplaced} (signatures: python/llm/src/ipex_llm/transformers/
// xpu_ops.py:64-162).  All loads/stores are 16 bytes per lane where the shape allows.
#include "common.h"

namespace bigdl {

enum DType : int { DT_F16 = 0, DT_BF16 = 1, DT_F32 = 2 };

template <int DT> struct Elem;
template <> struct Elem<DT_F16> {
  using T = _Float16;
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return (float)((const T*)p)[i]; }
  static __device__ __forceinline__ void st(void* p, int64_t i, float v) { ((T*)p)[i] = (T)v; }
};
template <> struct Elem<DT_BF16> {
  using T = uint16_t;
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return bf16_bits_to_f32(((const T*)p)[i]); }
  static __device__ __forceinline__ void st(void* p, int64_t i, float v) { ((T*)p)[i] = f32_to_bf16_bits(v); }
};
template <> struct Elem<DT_F32> {
  using T = float;
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return ((const T*)p)[i]; }
  static __device__ __forceinline__ void st(void* p, int64_t i, float v) { ((T*)p)[i] = v; }
};

// 8 consecutive elements <-> 8 floats (16-byte access for 2-byte types)
template <int DT>
__device__ __forceinline__ void load8(const void* p, int64_t i, float* v) {
  if constexpr (DT == DT_F16) {
    const half8 h = *(const half8*)((const _Float16*)p + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
  } else if constexpr (DT == DT_BF16) {
    const uint4v u = *(const uint4v*)((const uint16_t*)p + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = __uint_as_float(u[j] << 16);
      v[2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u);
    }
  } else {
    const float4v a = *(const float4v*)((const float*)p + i), b = *(const float4v*)((const float*)p + i + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
  }
}
template <int DT>
__device__ __forceinline__ void store8(void* p, int64_t i, const float* v) {
  if constexpr (DT == DT_F16) {
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
    *(half8*)((_Float16*)p + i) = h;
  } else if constexpr (DT == DT_BF16) {
    uint4v u;
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = (uint32_t)f32_to_bf16_bits(v[2 * j]) | ((uint32_t)f32_to_bf16_bits(v[2 * j + 1]) << 16);
    *(uint4v*)((uint16_t*)p + i) = u;
  } else {
    float4v a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = v[j]; b[j] = v[4 + j]; }
    *(float4v*)((float*)p + i) = a;
    *(float4v*)((float*)p + i + 4) = b;
  }
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.0f;
  for (int w = 0; w < nw; ++w) t += sh[w];
  __syncthreads();
  return t;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int w = 0; w < nw; ++w) t = fmaxf(t, sh[w]);
  __syncthreads();
  return t;
}

// ---- RMSNorm: out = (x [+ residual]) * rsqrt(mean(.^2) + eps) * (w + offset); optionally writes the
// summed residual stream back (res_out) -- one row per workgroup, row kept in registers ----
template <int DT, int VPT>  // VPT: 8-element vectors per thread
__global__ __launch_bounds__(256) void rms_norm_kernel(const void* x, const void* residual, const void* w, void* out,
                                                       void* res_out, int H, int64_t ldx, float eps, float offset) {
  __shared__ float sh[8];
  const int64_t row = blockIdx.x;
  float v[VPT][8];
  float ss = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (threadIdx.x + i * 256) * 8;
    if (c < H) {
      load8<DT>(x, row * ldx + c, v[i]);
      if (residual) {
        float r[8];
        load8<DT>(residual, row * (int64_t)H + c, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] += r[j];
      }
      if (res_out) store8<DT>(res_out, row * (int64_t)H + c, v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[i][j] * v[i][j];
    }
  }
  const float rs = rsqrtf(block_sum(ss, sh) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (threadIdx.x + i * 256) * 8;
    if (c < H) {
      float g[8];
      load8<DT>(w, c, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = v[i][j] * rs * (g[j] + offset);
      store8<DT>(out, row * (int64_t)H + c, v[i]);
    }
  }
}

template <int DT, int VPT>
__global__ __launch_bounds__(256) void layer_norm_kernel(const void* x, const void* w, const void* b, void* out, int H,
                                                         int64_t ldx, float eps) {
  __shared__ float sh[8];
  const int64_t row = blockIdx.x;
  float v[VPT][8];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (threadIdx.x + i * 256) * 8;
    if (c < H) {
      load8<DT>(x, row * ldx + c, v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j];
    }
  }
  const float mean = block_sum(s, sh) / (float)H;
  float ss = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (threadIdx.x + i * 256) * 8;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; ss += d * d; }
    }
  }
  const float rs = rsqrtf(block_sum(ss, sh) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (threadIdx.x + i * 256) * 8;
    if (c < H) {
      float g[8], bb[8];
      load8<DT>(w, c, g);
      if (b) load8<DT>(b, c, bb);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = (v[i][j] - mean) * rs * g[j] + (b ? bb[j] : 0.0f);
      store8<DT>(out, row * (int64_t)H + c, v[i]);
    }
  }
}

// gate[i] = act(gate[i]) * up[i]   (gate/up rows may be strided views of one fused buffer)
template <int DT>
__global__ __launch_bounds__(256) void act_mul_kernel(void* gate, const void* up, void* out, int64_t rows, int H,
                                                      int64_t ldg, int64_t ldu, int64_t ldo, int act) {
  const int64_t nvec = rows * (H / 8);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / (H / 8);
    const int c = (int)(i % (H / 8)) * 8;
    float g[8], u[8];
    load8<DT>(gate, r * ldg + c, g);
    load8<DT>(up, r * ldu + c, u);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = act_apply(act, g[j]) * u[j];
    store8<DT>(out, r * ldo + c, g);
  }
}

// backward of out = silu(gate) * up: dgate = g * up * s * (1 + gate * (1 - s)), dup = g * gate * s with s = sigmoid(gate)
// (training path: transformers/fused_train_ops.py)
template <int DT>
__global__ __launch_bounds__(256) void silu_mul_bwd_kernel(const void* gate, const void* up, const void* grad, void* dgate,
                                                           void* dup, int64_t rows, int H, int64_t ldg, int64_t ldu,
                                                           int64_t ldgr) {
  const int64_t nvec = rows * (H / 8);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / (H / 8);
    const int c = (int)(i % (H / 8)) * 8;
    float a[8], u[8], g[8], da[8], du[8];
    load8<DT>(gate, r * ldg + c, a);
    load8<DT>(up, r * ldu + c, u);
    load8<DT>(grad, r * ldgr + c, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float sg = 1.0f / (1.0f + __expf(-a[j]));
      da[j] = g[j] * u[j] * sg * (1.0f + a[j] * (1.0f - sg));
      du[j] = g[j] * a[j] * sg;
    }
    store8<DT>(dgate, r * (int64_t)H + c, da);
    store8<DT>(dup, r * (int64_t)H + c, du);
  }
}

// backward of RMSNorm w.r.t. its input, one row per workgroup: with w' = w + offset, rs = rsqrt(mean(x^2) + eps),
// dx = rs * (g w') - x * rs^3 * sum(g w' x) / H
template <int DT, int VPT>
__global__ __launch_bounds__(256) void rms_norm_bwd_kernel(const void* x, const void* w, const void* grad, void* dx, int H,
                                                           int64_t ldx, int64_t ldg, float eps, float offset) {
  __shared__ float sh[8];
  const int64_t row = blockIdx.x;
  float xv[VPT][8], gw[VPT][8];
  float ss = 0.0f, dot = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (threadIdx.x + i * 256) * 8;
    if (c < H) {
      float wv[8];
      load8<DT>(x, row * ldx + c, xv[i]);
      load8<DT>(grad, row * ldg + c, gw[i]);
      load8<DT>(w, c, wv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        gw[i][j] *= wv[j] + offset;
        ss += xv[i][j] * xv[i][j];
        dot += gw[i][j] * xv[i][j];
      }
    }
  }
  ss = block_sum(ss, sh);
  dot = block_sum(dot, sh);
  const float rs = rsqrtf(ss / (float)H + eps);
  const float k = rs * rs * rs * dot / (float)H;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (threadIdx.x + i * 256) * 8;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) gw[i][j] = rs * gw[i][j] - xv[i][j] * k;
      store8<DT>(dx, row * (int64_t)H + c, gw[i]);
    }
  }
}

// in-place softmax over the last dim, one row per workgroup
template <int DT>
__global__ __launch_bounds__(256) void softmax_kernel(void* x, int L, float scale) {
  __shared__ float sh[8];
  const int64_t row = blockIdx.x;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < L; c += 256) mx = fmaxf(mx, Elem<DT>::ld(x, row * L + c) * scale);
  mx = block_max(mx, sh);
  float s = 0.0f;
  for (int c = threadIdx.x; c < L; c += 256) s += __expf(Elem<DT>::ld(x, row * L + c) * scale - mx);
  s = block_sum(s, sh);
  const float inv = 1.0f / s;
  for (int c = threadIdx.x; c < L; c += 256)
    Elem<DT>::st(x, row * L + c, __expf(Elem<DT>::ld(x, row * L + c) * scale - mx) * inv);
}

// RoPE in place on a strided [B, H, S, D] view.  positions [B, S] (int64).  Either inv_freq
// (cos/sin computed on the fly) or cos/sin tables [B or 1, S, rot or rot/2] are given.
struct RopeArgs {
  void* x;
  int64_t sb, sh, ss;     // element strides of x (last dim contiguous)
  const int64_t* pos;     // [B, S]
  const float* inv_freq;  // [rot/2]
  const void* cos_t;      // tables (same dtype as x), row stride tab_ld, batch stride tab_sb
  const void* sin_t;
  int64_t tab_sb, tab_ld;
  int tab_full;           // 1: table has rot entries (duplicated halves), 0: rot/2 entries
  int B, H, S, D, rot;
  int interleaved;
  float scaling;
};

template <int DT>
__global__ __launch_bounds__(64) void rope_kernel(RopeArgs a) {
  const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int64_t base = b * a.sb + h * a.sh + s * a.ss;
  const int half = a.rot / 2;
  for (int i = threadIdx.x; i < half; i += 64) {
    const int i0 = a.interleaved ? 2 * i : i, i1 = a.interleaved ? 2 * i + 1 : i + half;
    float cs, sn;
    if (a.inv_freq) {
      const float p = (float)a.pos[(int64_t)b * a.S + s];
      sincosf(p * a.inv_freq[i], &sn, &cs);
      cs *= a.scaling;
      sn *= a.scaling;
    } else {
      const int ti = a.tab_full ? i0 : i;
      const int64_t to = b * a.tab_sb + (int64_t)s * a.tab_ld + ti;
      cs = Elem<DT>::ld(a.cos_t, to);
      sn = Elem<DT>::ld(a.sin_t, to);
    }
    const float x0 = Elem<DT>::ld(a.x, base + i0), x1 = Elem<DT>::ld(a.x, base + i1);
    Elem<DT>::st(a.x, base + i0, x0 * cs - x1 * sn);
    Elem<DT>::st(a.x, base + i1, x1 * cs + x0 * sn);
  }
}

__device__ __forceinline__ uint8_t f32_to_e5m2_sat(float f) {
  union { _Float16 h; uint16_t u; } c;
  c.h = (_Float16)fminf(fmaxf(f, -57344.0f), 57344.0f);
  uint32_t u = c.u;
  const uint32_t sign = u & 0x8000u;
  uint32_t mag = u & 0x7fffu;
  mag += 0x7fu + ((mag >> 8) & 1u);
  if (mag >= 0x7c00u) mag = 0x7b00u;
  return (uint8_t)((sign | mag) >> 8);
}

// generic contiguous fp8(e5m2) <-> fp16/bf16/fp32 conversion (KV cache for the module-level path)
template <int DT>
__global__ __launch_bounds__(256) void to_e5m2_kernel(const void* src, uint8_t* dst, int64_t rows, int D,
                                                      int64_t src_ld, int64_t dst_ld) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows * D; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    const int c = (int)(i % D);
    dst[r * dst_ld + c] = f32_to_e5m2_sat(Elem<DT>::ld(src, r * src_ld + c));
  }
}
template <int DT>
__global__ __launch_bounds__(256) void from_e5m2_kernel(const uint8_t* src, void* dst, int64_t rows, int D,
                                                        int64_t src_ld, int64_t dst_ld) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows * D; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    const int c = (int)(i % D);
    union { uint16_t u; _Float16 h; } cv;
    cv.u = (uint16_t)src[r * src_ld + c] << 8;
    Elem<DT>::st(dst, r * dst_ld + c, (float)cv.h);
  }
}

// scores[b, id] = s < 0 ? s * penalty : s / penalty  for every id in input_ids[b, :]
template <int DT>
__global__ __launch_bounds__(256) void rep_penalty_kernel(void* scores, const int64_t* ids, int B, int S, int V,
                                                          float penalty) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < S; i += 256) {
    const int64_t id = ids[(int64_t)b * S + i];
    if (id < 0 || id >= V) continue;
    // several positions may hold the same id: apply the penalty once, by the first occurrence
    bool first = true;
    for (int j = 0; j < i; ++j) if (ids[(int64_t)b * S + j] == id) { first = false; break; }
    if (!first) continue;
    const float s = Elem<DT>::ld(scores, (int64_t)b * V + id);
    Elem<DT>::st(scores, (int64_t)b * V + id, s < 0.0f ? s * penalty : s / penalty);
  }
}

// argmax over the last dim; ties -> lowest index.  A row is split over gridDim.y workgroups (a single workgroup reads a
// 32000-entry fp32 row at the rate of one CU: 6 us); each writes its (value, index) candidate, the last one to arrive
// (agent-scope ticket) picks the winner.  The same launch can add `inc_alpha * inc_src[i]` to `inc_dst[i]` for
// i < inc_n: the decode graph advances its per-sequence cache lengths here instead of in a launch of its own.
struct ArgmaxArgs {
  const void* x; int64_t* out; float* part_v; int* part_i; int* tickets;
  int* inc_dst; const int* inc_src; int inc_n, inc_alpha;
  int V; int64_t ld;
};

template <int DT>
__global__ __launch_bounds__(1024) void argmax_kernel(ArgmaxArgs a) {
  __shared__ float sv[16];
  __shared__ int si[16];
  __shared__ int s_last;
  const int64_t row = blockIdx.x;
  const int nch = gridDim.y, ch = blockIdx.y;
  if (row == 0 && ch == 0 && a.inc_dst != nullptr) {
    for (int i = threadIdx.x; i < a.inc_n; i += 1024) a.inc_dst[i] += a.inc_alpha * a.inc_src[i];
  }
  float best = -INFINITY;
  int bi = 0x7fffffff;
  const int nvec = a.V / 8;
  const int per = (nvec + nch - 1) / nch;
  const int v0 = ch * per, v1 = min(nvec, v0 + per);
  for (int i = v0 + threadIdx.x; i < v1; i += 1024) {
    float v[8];
    load8<DT>(a.x, row * a.ld + (int64_t)i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (v[j] > best) { best = v[j]; bi = i * 8 + j; }
  }
  if (ch == nch - 1) {
    for (int c = nvec * 8 + threadIdx.x; c < a.V; c += 1024) {
      const float v = Elem<DT>::ld(a.x, row * a.ld + c);
      if (v > best) { best = v; bi = c; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sv[wave] = best; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    if (nch == 1) {
      a.out[row] = bi == 0x7fffffff ? 0 : bi;
    } else {
      // candidates and ticket as device-scope atomics: written through, read by the last arriver past its own L1
      __hip_atomic_store(&a.part_v[row * nch + ch], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.part_i[row * nch + ch], bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int ticket = __hip_atomic_fetch_add(&a.tickets[row], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ticket == nch - 1) {
        float bv = -INFINITY;
        int bj = 0x7fffffff;
        for (int c = 0; c < nch; ++c) {
          const float v = __hip_atomic_load(&a.part_v[row * nch + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int j = __hip_atomic_load(&a.part_i[row * nch + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v > bv || (v == bv && j < bj)) { bv = v; bj = j; }
        }
        a.out[row] = bj == 0x7fffffff ? 0 : bj;
        __hip_atomic_store(&a.tickets[row], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      }
    }
  }
}

}  // namespace bigdl

using namespace bigdl;

#define DISPATCH_DT(dt, ...)                                  \
  switch (dt) {                                               \
    case DT_F16: { constexpr int DT = DT_F16; __VA_ARGS__; break; }   \
    case DT_BF16: { constexpr int DT = DT_BF16; __VA_ARGS__; break; } \
    case DT_F32: { constexpr int DT = DT_F32; __VA_ARGS__; break; }   \
    default: return -2;                                       \
  }

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// ---------------------------------------------------------------------------------------------
// Row-wise dynamic quantization of activations to OCP e4m3 for the opt-in fp8 prompt pass: per row
// scale = max|x| / 448, codes = round(x / scale).  One workgroup per row, the row stays in registers between the
// maximum and the conversion (K <= 256 threads x QR_V vectors x 8 halves = 16384).  The codes feed hipBLASLt's fp8 GEMM
// (torch._scaled_mm with row-wise scales), whose MFMA rate on gfx950 is twice the fp16 one.
// ---------------------------------------------------------------------------------------------
constexpr int QR_V = 8;
__global__ __launch_bounds__(256) void quant_rows_e4m3_kernel(const _Float16* __restrict__ x, uint8_t* __restrict__ q,
                                                               float* __restrict__ scale, int K, int64_t ldx, int64_t ldq) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const _Float16* xr = x + row * ldx;
  const int nvec = K >> 3;
  half8 v[QR_V];
  float amax = 0.0f;
#pragma unroll
  for (int i = 0; i < QR_V; ++i) {
    const int vi = threadIdx.x + 256 * i;
    v[i] = half8{0, 0, 0, 0, 0, 0, 0, 0};
    if (vi < nvec) v[i] = *(const half8*)(xr + 8 * vi);
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf((float)v[i][e]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = amax > 0.0f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / sc;
  if (threadIdx.x == 0) scale[row] = sc;
#pragma unroll
  for (int i = 0; i < QR_V; ++i) {
    const int vi = threadIdx.x + 256 * i;
    if (vi >= nvec) continue;
    uint32_t w0 = 0, w1 = 0;
    // v_cvt_pk_fp8_f32: two floats -> two e4m3 bytes in the low / high half of the destination word
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[i][0] * inv, (float)v[i][1] * inv, w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[i][2] * inv, (float)v[i][3] * inv, w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[i][4] * inv, (float)v[i][5] * inv, w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[i][6] * inv, (float)v[i][7] * inv, w1, true);
    *(uint2v*)(q + row * ldq + 8 * vi) = uint2v{w0, w1};
  }
}

// x fp16 [rows, K] (K % 8 == 0, K <= 16384, 16-byte aligned rows) -> q e4m3 bytes [rows, K] (8-byte aligned rows), scale fp32 [rows]
extern "C" int bigdl_quant_rows_e4m3(const void* x, void* q, float* scale, int64_t rows, int K, int64_t ldx, int64_t ldq,
                                     hipStream_t st) {
  if (rows <= 0 || K <= 0 || (K & 7) || K > 256 * QR_V * 8 || (ldx & 7) || (ldq & 7) || ((uintptr_t)x & 15) || ((uintptr_t)q & 7)) return -1;
  hipLaunchKernelGGL(quant_rows_e4m3_kernel, dim3((unsigned)rows), dim3(256), 0, st, (const _Float16*)x, (uint8_t*)q, scale, K, ldx, ldq);
  return HIP_CHECK_LAUNCH();
}

// bf16 -> fp16, 8 values per thread (the row-wise fp8 GEMM of hipBLASLt only writes bf16; PyTorch's own conversion
// kernel for this pair of types is not vectorized and took longer than the GEMM it followed)
__global__ __launch_bounds__(256) void bf16_to_f16_kernel(const uint4v* __restrict__ src, uint4v* __restrict__ dst, int64_t nvec) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const uint4v v = src[i];
  uint4v o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
    const half2v h = {(_Float16)fminf(fmaxf(lo, -65504.0f), 65504.0f), (_Float16)fminf(fmaxf(hi, -65504.0f), 65504.0f)};
    o[e] = __builtin_bit_cast(uint32_t, h);
  }
  dst[i] = o;
}

// n values (n % 8 == 0), both buffers 16-byte aligned and contiguous
extern "C" int bigdl_bf16_to_f16(const void* src, void* dst, int64_t n, hipStream_t st) {
  if (n <= 0 || (n & 7) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return -1;
  const int64_t nvec = n >> 3;
  hipLaunchKernelGGL(bf16_to_f16_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st, (const uint4v*)src, (uint4v*)dst, nvec);
  return HIP_CHECK_LAUNCH();
}

extern "C" int bigdl_rms_norm(const void* x, const void* residual, const void* w, void* out, void* res_out, int64_t rows,
                              int H, int64_t ldx, float eps, float offset, int dtype, hipStream_t st) {
  if (rows <= 0 || H <= 0 || (H & 7) || (ldx & 7) || H > 8 * 256 * 8) return -1;
  if (!aligned16(x) || !aligned16(w) || !aligned16(out) || (residual && !aligned16(residual)) ||
      (res_out && !aligned16(res_out))) return -1;
  const int vpt = (H / 8 + 255) / 256;
  DISPATCH_DT(dtype,
    if (vpt <= 1) hipLaunchKernelGGL((rms_norm_kernel<DT, 1>), dim3(rows), dim3(256), 0, st, x, residual, w, out, res_out, H, ldx, eps, offset);
    else if (vpt <= 2) hipLaunchKernelGGL((rms_norm_kernel<DT, 2>), dim3(rows), dim3(256), 0, st, x, residual, w, out, res_out, H, ldx, eps, offset);
    else if (vpt <= 4) hipLaunchKernelGGL((rms_norm_kernel<DT, 4>), dim3(rows), dim3(256), 0, st, x, residual, w, out, res_out, H, ldx, eps, offset);
    else hipLaunchKernelGGL((rms_norm_kernel<DT, 8>), dim3(rows), dim3(256), 0, st, x, residual, w, out, res_out, H, ldx, eps, offset))
  return HIP_CHECK_LAUNCH();
}

extern "C" int bigdl_layer_norm(const void* x, const void* w, const void* b, void* out, int64_t rows, int H, int64_t ldx,
                                float eps, int dtype, hipStream_t st) {
  if (rows <= 0 || H <= 0 || (H & 7) || (ldx & 7) || H > 8 * 256 * 8) return -1;
  if (!aligned16(x) || !aligned16(w) || !aligned16(out) || (b && !aligned16(b))) return -1;
  const int vpt = (H / 8 + 255) / 256;
  DISPATCH_DT(dtype,
    if (vpt <= 1) hipLaunchKernelGGL((layer_norm_kernel<DT, 1>), dim3(rows), dim3(256), 0, st, x, w, b, out, H, ldx, eps);
    else if (vpt <= 2) hipLaunchKernelGGL((layer_norm_kernel<DT, 2>), dim3(rows), dim3(256), 0, st, x, w, b, out, H, ldx, eps);
    else if (vpt <= 4) hipLaunchKernelGGL((layer_norm_kernel<DT, 4>), dim3(rows), dim3(256), 0, st, x, w, b, out, H, ldx, eps);
    else hipLaunchKernelGGL((layer_norm_kernel<DT, 8>), dim3(rows), dim3(256), 0, st, x, w, b, out, H, ldx, eps))
  return HIP_CHECK_LAUNCH();
}

extern "C" int bigdl_act_mul(void* gate, const void* up, void* out, int64_t rows, int H, int64_t ldg, int64_t ldu,
                             int64_t ldo, int act, int dtype, hipStream_t st) {
  if (rows <= 0 || H <= 0 || (H & 7) || (ldg & 7) || (ldu & 7) || (ldo & 7)) return -1;
  if (!aligned16(gate) || !aligned16(up) || !aligned16(out)) return -1;
  const int64_t nvec = rows * (H / 8);
  const int grid = (int)((nvec + 255) / 256 < 2048 ? (nvec + 255) / 256 : 2048);
  DISPATCH_DT(dtype, hipLaunchKernelGGL((act_mul_kernel<DT>), dim3(grid), dim3(256), 0, st, gate, up, out, rows, H, ldg, ldu, ldo, act))
  return HIP_CHECK_LAUNCH();
}

extern "C" int bigdl_silu_mul_bwd(const void* gate, const void* up, const void* grad, void* dgate, void* dup, int64_t rows,
                                  int H, int64_t ldg, int64_t ldu, int64_t ldgr, int dtype, hipStream_t st) {
  if (rows <= 0 || H <= 0 || (H & 7) || (ldg & 7) || (ldu & 7) || (ldgr & 7)) return -1;
  if (!aligned16(gate) || !aligned16(up) || !aligned16(grad) || !aligned16(dgate) || !aligned16(dup)) return -1;
  const int64_t nvec = rows * (H / 8);
  const int grid = (int)((nvec + 255) / 256 < 4096 ? (nvec + 255) / 256 : 4096);
  DISPATCH_DT(dtype, hipLaunchKernelGGL((silu_mul_bwd_kernel<DT>), dim3(grid), dim3(256), 0, st, gate, up, grad, dgate, dup, rows, H, ldg, ldu, ldgr))
  return HIP_CHECK_LAUNCH();
}

extern "C" int bigdl_rms_norm_bwd(const void* x, const void* w, const void* grad, void* dx, int64_t rows, int H, int64_t ldx,
                                  int64_t ldg, float eps, float offset, int dtype, hipStream_t st) {
  if (rows <= 0 || H <= 0 || (H & 7) || (ldx & 7) || (ldg & 7) || H > 8 * 256 * 8) return -1;
  if (!aligned16(x) || !aligned16(w) || !aligned16(grad) || !aligned16(dx)) return -1;
  const int vpt = (H / 8 + 255) / 256;
  DISPATCH_DT(dtype,
    if (vpt <= 1) hipLaunchKernelGGL((rms_norm_bwd_kernel<DT, 1>), dim3(rows), dim3(256), 0, st, x, w, grad, dx, H, ldx, ldg, eps, offset);
    else if (vpt <= 2) hipLaunchKernelGGL((rms_norm_bwd_kernel<DT, 2>), dim3(rows), dim3(256), 0, st, x, w, grad, dx, H, ldx, ldg, eps, offset);
    else if (vpt <= 4) hipLaunchKernelGGL((rms_norm_bwd_kernel<DT, 4>), dim3(rows), dim3(256), 0, st, x, w, grad, dx, H, ldx, ldg, eps, offset);
    else hipLaunchKernelGGL((rms_norm_bwd_kernel<DT, 8>), dim3(rows), dim3(256), 0, st, x, w, grad, dx, H, ldx, ldg, eps, offset))
  return HIP_CHECK_LAUNCH();
}

extern "C" int bigdl_softmax_inplace(void* x, int64_t rows, int L, float scale, int dtype, hipStream_t st) {
  if (rows <= 0 || L <= 0) return -1;
  DISPATCH_DT(dtype, hipLaunchKernelGGL((softmax_kernel<DT>), dim3(rows), dim3(256), 0, st, x, L, scale))
  return HIP_CHECK_LAUNCH();
}

extern "C" int bigdl_rope_inplace(void* x, int64_t sb, int64_t sh, int64_t ss, const int64_t* pos, const float* inv_freq,
                                  const void* cos_t, const void* sin_t, int64_t tab_sb, int64_t tab_ld, int tab_full,
                                  int B, int H, int S, int D, int rot, int interleaved, float scaling, int dtype,
                                  hipStream_t st) {
  if (B <= 0 || H <= 0 || S <= 0 || rot <= 0 || rot > D || (rot & 1)) return -1;
  if (!inv_freq && !(cos_t && sin_t)) return -1;
  if (inv_freq && !pos) return -1;
  RopeArgs a;
  a.x = x; a.sb = sb; a.sh = sh; a.ss = ss; a.pos = pos; a.inv_freq = inv_freq; a.cos_t = cos_t; a.sin_t = sin_t;
  a.tab_sb = tab_sb; a.tab_ld = tab_ld; a.tab_full = tab_full; a.B = B; a.H = H; a.S = S; a.D = D; a.rot = rot;
  a.interleaved = interleaved; a.scaling = scaling;
  DISPATCH_DT(dtype, hipLaunchKernelGGL((rope_kernel<DT>), dim3(S, H, B), dim3(64), 0, st, a))
  return HIP_CHECK_LAUNCH();
}

extern "C" int bigdl_to_e5m2(const void* src, void* dst, int64_t rows, int D, int64_t src_ld, int64_t dst_ld, int dtype,
                             hipStream_t st) {
  if (rows <= 0 || D <= 0) return -1;
  const int64_t n = rows * D;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  DISPATCH_DT(dtype, hipLaunchKernelGGL((to_e5m2_kernel<DT>), dim3(grid), dim3(256), 0, st, src, (uint8_t*)dst, rows, D, src_ld, dst_ld))
  return HIP_CHECK_LAUNCH();
}

extern "C" int bigdl_from_e5m2(const void* src, void* dst, int64_t rows, int D, int64_t src_ld, int64_t dst_ld, int dtype,
                               hipStream_t st) {
  if (rows <= 0 || D <= 0) return -1;
  const int64_t n = rows * D;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  DISPATCH_DT(dtype, hipLaunchKernelGGL((from_e5m2_kernel<DT>), dim3(grid), dim3(256), 0, st, (const uint8_t*)src, dst, rows, D, src_ld, dst_ld))
  return HIP_CHECK_LAUNCH();
}

extern "C" int bigdl_repetition_penalty(void* scores, const int64_t* ids, int B, int S, int V, float penalty, int dtype,
                                        hipStream_t st) {
  if (B <= 0 || S <= 0 || V <= 0) return -1;
  DISPATCH_DT(dtype, hipLaunchKernelGGL((rep_penalty_kernel<DT>), dim3(B), dim3(256), 0, st, scores, ids, B, S, V, penalty))
  return HIP_CHECK_LAUNCH();
}

// ws (nch > 1, rows <= 64): 64 tickets (zero before the first launch; every launch leaves them zero), then 64 * nch
// candidate values and 64 * nch candidate indices: the tickets never move when the row count changes between calls
extern "C" int bigdl_argmax(const void* x, int64_t* out, int64_t rows, int V, int64_t ld, int dtype, int nch, void* ws,
                            int* inc_dst, const int* inc_src, int inc_n, int inc_alpha, hipStream_t st) {
  if (rows <= 0 || V <= 0 || (ld & 7) || !aligned16(x) || nch < 1 || (nch > 1 && (ws == nullptr || rows > 64))) return -1;
  ArgmaxArgs a;
  a.x = x; a.out = out; a.V = V; a.ld = ld;
  a.tickets = (int*)ws; a.part_v = (float*)ws + 64; a.part_i = (int*)ws + 64 + 64 * nch;
  a.inc_dst = inc_dst; a.inc_src = inc_src; a.inc_n = inc_dst ? inc_n : 0; a.inc_alpha = inc_alpha;
  DISPATCH_DT(dtype, hipLaunchKernelGGL((argmax_kernel<DT>), dim3(rows, nch), dim3(1024), 0, st, a))
  return HIP_CHECK_LAUNCH();
}
