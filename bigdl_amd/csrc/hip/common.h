// Shared device helpers for the gfx950 kernels: vector types, MFMA wrappers, wave64
// reductions, quantized-code -> fp16 fragment decoders for the T16 weight layout
// (layout contract: bigdl_amd/csrc/host/relayout.cpp header comment).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace bigdl {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));
typedef uint32_t uint2v __attribute__((ext_vector_type(2)));

enum QType : int {
  Q_SYM_INT4 = 2, Q_ASYM_INT4 = 3, Q_SYM_INT5 = 6, Q_ASYM_INT5 = 7, Q_SYM_INT8 = 8,
  Q_NF4 = 10, Q_NF3 = 11, Q_FP16 = 12, Q_FP8_E4M3 = 15, Q_FP4 = 16, Q_FP8_E5M2 = 19, Q_BF16 = 20,
  Q_IQ2_XXS = 21, Q_IQ2_XS = 22, Q_Q2_K = 23, Q_IQ1_S = 24, Q_IQ1_M = 25, Q_Q6_K = 26, Q_Q4_K = 27,
  Q_Q5_K = 28, Q_FP6 = 29, Q_FP6_K = 30, Q_WOQ_INT4 = 34,
};

enum OutType : int { OUT_F16 = 0, OUT_BF16 = 1, OUT_F32 = 2 };
enum ActType : int { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2, ACT_GELU_TANH = 3, ACT_RELU = 4 };

__device__ __forceinline__ float4v mfma16(half8 a, half8 b, float4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t orv) { return (a & mask) | orv; }

union H8 {
  half8 v;
  uint32_t u[4];
  half2v h2[4];
  uint4v q;
};

__device__ __forceinline__ half2v u2h(uint32_t u) {
  union { uint32_t u; half2v h; } c;
  c.u = u;
  return c.h;
}
__device__ __forceinline__ uint32_t h2u(half2v h) {
  union { uint32_t u; half2v h; } c;
  c.h = h;
  return c.u;
}

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  uint32_t x = __float_as_uint(f);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40);
  x += 0x7fffu + ((x >> 16) & 1);
  return (uint16_t)(x >> 16);
}

template <int OUT>
__device__ __forceinline__ void store_out(void* p, int64_t idx, float v) {
  if constexpr (OUT == OUT_F16) ((_Float16*)p)[idx] = (_Float16)v;
  else if constexpr (OUT == OUT_BF16) ((uint16_t*)p)[idx] = f32_to_bf16_bits(v);
  else ((float*)p)[idx] = v;
}
template <int OUT>
__device__ __forceinline__ float load_out(const void* p, int64_t idx) {
  if constexpr (OUT == OUT_F16) return (float)((const _Float16*)p)[idx];
  else if constexpr (OUT == OUT_BF16) return bf16_bits_to_f32(((const uint16_t*)p)[idx]);
  else return ((const float*)p)[idx];
}

__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case ACT_SILU: return x / (1.0f + __expf(-x));
    case ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    case ACT_GELU_TANH: {
      const float t = 0.7978845608028654f * (x + 0.044715f * x * x * x);
      return 0.5f * x * (1.0f + tanhf(t));
    }
    case ACT_RELU: return x > 0.0f ? x : 0.0f;
    default: return x;
  }
}

__device__ __forceinline__ uint8_t f16_to_e5m2(_Float16 h) {
  // round-to-nearest-even on the dropped 8 mantissa bits, saturate to the largest finite value
  union { _Float16 h; uint16_t u; } c;
  c.h = h;
  uint32_t u = c.u;
  const uint32_t sign = u & 0x8000u;
  uint32_t mag = u & 0x7fffu;
  if (mag > 0x7c00u) return (uint8_t)((sign >> 8) | 0x7f);   // NaN
  mag += 0x7fu + ((mag >> 8) & 1u);
  if (mag >= 0x7c00u) mag = 0x7b00u;
  return (uint8_t)((sign | mag) >> 8);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// wave_sum with the four in-row steps as DPP moves (a few cycles each) and only the two cross-row steps as LDS-routed
// permutes: the chain of six dependent permutes of wave_sum is most of a microsecond where a kernel waits on it.  (Another
// summation order than wave_sum: the two are not interchangeable where results are compared bit for bit.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_mov<0xB1>(v);     // quad_perm [1, 0, 3, 2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2, 3, 0, 1]
  v += dpp_mov<0x141>(v);    // row_half_mirror
  v += dpp_mov<0x140>(v);    // row_mirror
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---------------------------------------------------------------------------------
// Decoders: one 32-bit word (4-bit plane) or two words (8-bit plane) of T16 weight codes
// -> the 8 fp16 values of a B fragment, *unscaled*.  `post` is the factor by which the
// caller must multiply the block scale (power of two folded out of the bit tricks).
// ---------------------------------------------------------------------------------

// sym_int4 / sym_int5 low nibbles: value = code - 8 (exact small integers in fp16)
__device__ __forceinline__ half8 dec_int4(uint32_t w, float bias) {
  H8 r;
  const half2v off = {(_Float16)(1024.0f + bias), (_Float16)(1024.0f + bias)};
  r.h2[0] = u2h(and_or(w, 0x000F000Fu, 0x64006400u)) - off;
  r.h2[1] = u2h(and_or(w >> 4, 0x000F000Fu, 0x64006400u)) - off;
  r.h2[2] = u2h(and_or(w >> 8, 0x000F000Fu, 0x64006400u)) - off;
  r.h2[3] = u2h(and_or(w >> 12, 0x000F000Fu, 0x64006400u)) - off;
  return r.v;
}

// 5-bit: low nibbles in w, high bit of code j in bit j of hb
__device__ __forceinline__ half8 dec_int5(uint32_t w, uint32_t hb, float bias) {
  H8 r;
  const half2v off = {(_Float16)(1024.0f + bias), (_Float16)(1024.0f + bias)};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // codes (2i, 2i+1): high bits are bits 2i and 2i+1 of hb -> bit 4 of each half
    const uint32_t hi = (((hb >> (2 * i)) & 1u) << 4) | (((hb >> (2 * i + 1)) & 1u) << 20);
    r.h2[i] = u2h(and_or(w >> (4 * i), 0x000F000Fu, 0x64006400u) | hi) - off;
  }
  return r.v;
}

// fp4 e2m1: code = s|e1 e0|m -> fp16 bits (s<<15)|(eem<<9); true value = fp16 * 2^14
__device__ __forceinline__ half8 dec_fp4(uint32_t w) {
  H8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t t = w >> (4 * i);
    r.u[i] = ((t & 0x00070007u) << 9) | ((t & 0x00080008u) << 12);
  }
  return r.v;
}
constexpr float FP4_POST = 16384.0f;

// fp6 e3m2: code = s|e2 e1 e0|m1 m0, low nibble in w, high 2 bits (s,e2) of code j in bits 2j..2j+1 of hb
// fp16 bits = (s<<15) | ((code & 31) << 8); true value = fp16 * 2^12
__device__ __forceinline__ half8 dec_fp6(uint32_t w, uint32_t hb) {
  H8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t lo = ((w >> (4 * i)) & 0x000F000Fu) << 8;
    const uint32_t a = (hb >> (4 * i)) & 3u;        // code 2i   : bit0 = e2, bit1 = s
    const uint32_t b = (hb >> (4 * i + 2)) & 3u;    // code 2i+1
    const uint32_t hi = ((a & 1u) << 12) | ((a >> 1) << 15) | ((b & 1u) << 28) | ((b >> 1) << 31);
    r.u[i] = lo | hi;
  }
  return r.v;
}
constexpr float FP6_POST = 4096.0f;

__constant__ const float NF4_VALUES[16] = {
    -1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
    -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
    0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
    0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};

// nf4: 16-entry table lookup; the table lives in LDS as 16 fp16 values replicated per lane
// group to keep bank conflicts low (see gemv kernel)
__device__ __forceinline__ half8 dec_nf4(uint32_t w, const _Float16* lut) {
  H8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t t = w >> (4 * i);
    half2v h = {lut[t & 15u], lut[(t >> 16) & 15u]};
    r.h2[i] = h;
  }
  return r.v;
}

// nf4 through a bank-private table: 16 entries x 32 dwords in LDS, entry i of lane l at dword i * 32 + (l & 31), the
// fp16 value replicated in both halves of the dword.  Every lane only ever touches bank l & 31 (lanes l and l + 32
// read the same dword: a broadcast), so the 8 lookups of a fragment are conflict free whatever the codes are; the
// replicated halves let one v_bfi merge two lookups into a packed pair.  `lut` already points at the lane's column.
__device__ __forceinline__ half8 dec_nf4_banked(uint32_t w, const uint32_t* lut) {
  H8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t t = w >> (4 * i);
    const uint32_t a = lut[(t & 15u) << 5], b = lut[((t >> 16) & 15u) << 5];
    r.u[i] = (a & 0x0000ffffu) | (b & 0xffff0000u);
  }
  return r.v;
}
__device__ __forceinline__ void fill_nf4_banked(uint32_t* table, int tid, int nthreads) {
  for (int i = tid; i < 16 * 32; i += nthreads) {
    union { _Float16 h; uint16_t u; } c;
    c.h = (_Float16)NF4_VALUES[i >> 5];
    table[i] = (uint32_t)c.u | ((uint32_t)c.u << 16);
  }
}

// int8 codes, 8 bytes (two words): value = (int8)code.  fp16 0x6400|b' = 1024 + b', b' = b ^ 0x80
__device__ __forceinline__ half8 dec_int8(uint32_t w0, uint32_t w1) {
  H8 r;
  const half2v off = {(_Float16)1152.0f, (_Float16)1152.0f};
  w0 ^= 0x80808080u;
  w1 ^= 0x80808080u;
  r.h2[0] = u2h(__builtin_amdgcn_perm(0x64646464u, w0, 0x04010400u)) - off;
  r.h2[1] = u2h(__builtin_amdgcn_perm(0x64646464u, w0, 0x04030402u)) - off;
  r.h2[2] = u2h(__builtin_amdgcn_perm(0x64646464u, w1, 0x04010400u)) - off;
  r.h2[3] = u2h(__builtin_amdgcn_perm(0x64646464u, w1, 0x04030402u)) - off;
  return r.v;
}

// fp8 e5m2 is the top byte of an fp16: exact
__device__ __forceinline__ half8 dec_e5m2(uint32_t w0, uint32_t w1) {
  H8 r;
  r.u[0] = __builtin_amdgcn_perm(0u, w0, 0x010c000cu);
  r.u[1] = __builtin_amdgcn_perm(0u, w0, 0x030c020cu);
  r.u[2] = __builtin_amdgcn_perm(0u, w1, 0x010c000cu);
  r.u[3] = __builtin_amdgcn_perm(0u, w1, 0x030c020cu);
  return r.v;
}

// fp8 e4m3fn: fp16 bits = (s<<15) | ((b & 0x7f) << 7); true value = fp16 * 2^8
__device__ __forceinline__ half8 dec_e4m3(uint32_t w0, uint32_t w1) {
  H8 r = {dec_e5m2(w0, w1)};
#pragma unroll
  for (int i = 0; i < 4; ++i) r.u[i] = (r.u[i] & 0x80008000u) | ((r.u[i] >> 1) & 0x3F803F80u);
  return r.v;
}
constexpr float E4M3_POST = 256.0f;


// k-quants with 32-wide sub-blocks are tiled as their asymmetric per-32 equivalents on the device
// (bigdl_amd/csrc/host/relayout.cpp): the kernels only ever see the equivalent format
__host__ __device__ inline int t16_kernel_qtype(int qtype) {
  return qtype == Q_Q4_K ? Q_ASYM_INT4 : (qtype == Q_Q5_K ? Q_ASYM_INT5 : qtype);
}

struct T16Planes {
  const uint8_t* lo;
  const uint8_t* hi;
  const _Float16* sc;
  const _Float16* mn;
};

__host__ __device__ constexpr inline int lo_bits_of(int qtype) {
  return (qtype == Q_SYM_INT8 || qtype == Q_FP8_E4M3 || qtype == Q_FP8_E5M2) ? 8 : 4;
}
__host__ __device__ constexpr inline int hi_bits_of(int qtype) {
  return (qtype == Q_SYM_INT5 || qtype == Q_ASYM_INT5) ? 1 : (qtype == Q_FP6 ? 2 : 0);
}
__host__ __device__ constexpr inline bool has_min_of(int qtype) {
  return qtype == Q_ASYM_INT4 || qtype == Q_ASYM_INT5 || qtype == Q_WOQ_INT4;
}
__host__ __device__ inline bool is_t16(int qtype) {
  switch (qtype) {
    case Q_SYM_INT4: case Q_ASYM_INT4: case Q_FP4: case Q_NF4: case Q_WOQ_INT4: case Q_SYM_INT5:
    case Q_ASYM_INT5: case Q_FP6: case Q_SYM_INT8: case Q_FP8_E4M3: case Q_FP8_E5M2: return true;
    default: return false;
  }
}

__host__ __device__ inline T16Planes t16_planes(const void* base, int qtype, int64_t N, int64_t K) {
  const int64_t Np = (N + 15) / 16 * 16, Kp = (K + 127) / 128 * 128;
  T16Planes p;
  const uint8_t* cur = (const uint8_t*)base;
  p.lo = cur;
  cur += Np * Kp * lo_bits_of(qtype) / 8;
  p.hi = hi_bits_of(qtype) ? cur : nullptr;
  cur += Np * Kp * hi_bits_of(qtype) / 8;
  p.sc = (const _Float16*)cur;
  cur += Np * (Kp / 32) * 2;
  p.mn = has_min_of(qtype) ? (const _Float16*)cur : nullptr;
  return p;
}

}  // namespace bigdl

#define HIP_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -100)
