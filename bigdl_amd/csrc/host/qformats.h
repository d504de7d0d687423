// Block-quantized weight formats: ids, block geometry, on-disk ("host") layouts.
//
// The qtype ids are the persisted contract: they match the reference table in
// python/llm/src/ipex_llm/ggml/quantize.py:28-61 so configs and checkpoints keep
// the same names/ids.  The host layouts for sym_int4/asym_int4/sym_int5/asym_int5/
// sym_int8/q2_k/q4_k/q5_k/q6_k are the public ggml block layouts (reference reader:
// python/llm/src/ipex_llm/transformers/gguf/gguf.py:273-367).  The layouts of
// nf4/nf3/fp4/fp6/fp8_*/fp6_k/woq_int4 are defined by this library (the reference
// keeps them inside a closed binary) and are documented next to each struct.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <cmath>

namespace bigdl {

enum QType : int {
  Q_SYM_INT4 = 2,
  Q_ASYM_INT4 = 3,
  Q_SYM_INT5 = 6,
  Q_ASYM_INT5 = 7,
  Q_SYM_INT8 = 8,
  Q_NF4 = 10,
  Q_NF3 = 11,
  Q_FP16 = 12,
  Q_FP8_E4M3 = 15,
  Q_FP4 = 16,
  Q_MIXED_FP4 = 17,
  Q_MIXED_FP8 = 18,
  Q_FP8_E5M2 = 19,
  Q_BF16 = 20,
  Q_IQ2_XXS = 21,
  Q_IQ2_XS = 22,
  Q_Q2_K = 23,
  Q_IQ1_S = 24,
  Q_IQ1_M = 25,
  Q_Q6_K = 26,
  Q_Q4_K = 27,
  Q_Q5_K = 28,
  Q_FP6 = 29,
  Q_FP6_K = 30,
  Q_SYM_INT4_RTN = 31,
  Q_SYM_INT8_RTN = 32,
  Q_ASYM_INT4_RTN = 33,
  Q_WOQ_INT4 = 34,
};

constexpr int QK = 32;     // elements per block for the "simple" formats
constexpr int QK_NF4 = 64; // nf4 uses 64-element blocks (one absmax per 64)
constexpr int QK_K = 256;  // k-quant super-block
constexpr int QK_WOQ = 128;

using fp16_t = uint16_t;

// ---- fp16 <-> fp32 (bit exact, round-to-nearest-even, handles subnormals) ----
static inline float fp16_to_fp32(fp16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1f;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal: renormalise
      int e = -1;
      do { man <<= 1; ++e; } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 112) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

static inline fp16_t fp32_to_fp16(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (fp16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
  if (x >= 0x477ff000u) return (fp16_t)(sign | 0x7c00u);  // overflow -> inf
  if (x < 0x33000001u) return (fp16_t)sign;              // underflow -> 0
  int exp = (int)(x >> 23) - 127;
  uint32_t man = (x & 0x7fffffu) | 0x800000u;
  int shift;
  uint32_t hexp;
  if (exp < -14) { shift = 13 + (-14 - exp); hexp = 0; }
  else { shift = 13; hexp = (uint32_t)(exp + 15); man &= 0x7fffffu; }
  uint32_t half = man >> shift;
  const uint32_t rem = man & ((1u << shift) - 1);
  const uint32_t halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (half & 1))) ++half;
  return (fp16_t)(sign | ((hexp << 10) + half));  // mantissa carry bumps the exponent
}

static inline float bf16_to_fp32(uint16_t h) {
  uint32_t bits = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

static inline uint16_t fp32_to_bf16(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40);
  x += 0x7fffu + ((x >> 16) & 1);
  return (uint16_t)(x >> 16);
}

#pragma pack(push, 1)
// ggml q4_0: value = (q - 8) * d ; byte j: low nibble = element j, high = element j+16
struct block_q4_0 { fp16_t d; uint8_t qs[QK / 2]; };
// ggml q4_1: value = q * d + m
struct block_q4_1 { fp16_t d; fp16_t m; uint8_t qs[QK / 2]; };
// ggml q5_0: 5th bit of element j in bit j of qh; value = (q - 16) * d
struct block_q5_0 { fp16_t d; uint8_t qh[4]; uint8_t qs[QK / 2]; };
struct block_q5_1 { fp16_t d; fp16_t m; uint8_t qh[4]; uint8_t qs[QK / 2]; };
// ggml q8_0: value = q * d
struct block_q8_0 { fp16_t d; int8_t qs[QK]; };
// nf4 (own layout): 64 elements, d = absmax; byte j: low nibble = code of element j,
// high nibble = code of element j+32; value = NF4_TABLE[code] * d
struct block_nf4 { fp16_t d; uint8_t qs[QK_NF4 / 2]; };
// nf3 (own layout): 32 elements; ql packs the 2 low code bits (4 per byte, element
// 4j+i in bits 2i..2i+1 of ql[j]); qh bit j%8 of qh[j/8] is the high code bit.
struct block_nf3 { fp16_t d; uint8_t ql[QK / 4]; uint8_t qh[QK / 8]; };
// fp4 e2m1 (own layout): same nibble packing as q4_0; code = s|e1 e0|m ;
// value = E2M1[code] * d with d = absmax / 6
struct block_fp4 { fp16_t d; uint8_t qs[QK / 2]; };
// fp6 e3m2 (own layout): code = s|e2 e1 e0|m1 m0 ; ql = low nibbles packed like q4_0,
// qh = two high bits (element j in bits 2(j%4)..  of qh[j/4]); value = E3M2[code]*d,
// d = absmax / 28
struct block_fp6 { fp16_t d; uint8_t ql[QK / 2]; uint8_t qh[QK / 4]; };
// fp8 (own layout): one OCP fp8 byte per element (e4m3fn: max 448, e5m2: max 57344);
// value = fp8(code) * d with d = absmax / fp8_max
struct block_fp8 { fp16_t d; uint8_t qs[QK]; };
// woq_int4 (own layout): 128 elements, asymmetric; value = q * d + m ;
// byte j: low nibble = element j, high nibble = element j+64
struct block_woq4 { fp16_t d; fp16_t m; uint8_t qs[QK_WOQ / 2]; };

// ggml k-quants
struct block_q2_K { uint8_t scales[QK_K / 16]; uint8_t qs[QK_K / 4]; fp16_t d; fp16_t dmin; };
struct block_q4_K { fp16_t d; fp16_t dmin; uint8_t scales[12]; uint8_t qs[QK_K / 2]; };
struct block_q5_K { fp16_t d; fp16_t dmin; uint8_t scales[12]; uint8_t qh[QK_K / 8]; uint8_t qs[QK_K / 2]; };
struct block_q6_K { uint8_t ql[QK_K / 2]; uint8_t qh[QK_K / 4]; int8_t scales[QK_K / 16]; fp16_t d; };
// fp6_k (own layout): 256 elements = 8 sub-blocks of 32 e3m2 codes with int8 sub-scales
// value = E3M2[code] * d * scales[sub]
struct block_fp6_K { fp16_t d; int8_t scales[QK_K / 32]; uint8_t ql[QK_K / 2]; uint8_t qh[QK_K / 4]; };
// 2-bit/1-bit lattice formats (own codebook, see lattice.cpp): 256 elements.
// iq2_xxs: per 32 elements four 8-bit grid indices, 28 sign bits, 4-bit scale
struct block_iq2_xxs { fp16_t d; uint16_t qs[QK_K / 8]; };
// iq2_xs: per 8 elements a 9-bit grid index + 7 sign bits; 4-bit scale per 16
struct block_iq2_xs { fp16_t d; uint16_t qs[QK_K / 8]; uint8_t scales[QK_K / 32]; };
// iq1_s: per 8 elements an 11-bit index into a ternary grid; 3-bit scale + sign of delta per 32
struct block_iq1_s { fp16_t d; uint8_t qs[QK_K / 8]; uint16_t qh[QK_K / 32]; };
#pragma pack(pop)

static_assert(sizeof(block_q4_0) == 18, "q4_0");
static_assert(sizeof(block_q4_1) == 20, "q4_1");
static_assert(sizeof(block_q5_0) == 22, "q5_0");
static_assert(sizeof(block_q5_1) == 24, "q5_1");
static_assert(sizeof(block_q8_0) == 34, "q8_0");
static_assert(sizeof(block_q2_K) == 84, "q2_K");
static_assert(sizeof(block_q4_K) == 144, "q4_K");
static_assert(sizeof(block_q5_K) == 176, "q5_K");
static_assert(sizeof(block_q6_K) == 210, "q6_K");

// elements per block / bytes per block ; 0 = not a block format
int qk_size(int qtype);
int type_size(int qtype);

extern const float NF4_TABLE[16];
extern const float NF3_TABLE[8];
extern const float E2M1_TABLE[16];
extern float E3M2_TABLE[64];
extern float E4M3_TABLE[256];
extern float E5M2_TABLE[256];
void init_tables();

// nearest-code encoders
uint8_t encode_fp8_e4m3(float x);
uint8_t encode_fp8_e5m2(float x);
uint8_t encode_e3m2(float x);
uint8_t encode_e2m1(float x);
uint8_t encode_nf4(float x);
uint8_t encode_nf3(float x);

// row-level kernels: n must be a multiple of the block size
void quantize_row(int qtype, const float* x, void* y, int64_t n, const float* imatrix, bool scale_search);
void dequantize_row(int qtype, const void* x, float* y, int64_t n);

// lattice formats (lattice.cpp)
void quantize_row_iq2_xxs(const float* x, block_iq2_xxs* y, int64_t n, const float* w);
void quantize_row_iq2_xs(const float* x, block_iq2_xs* y, int64_t n, const float* w);
void quantize_row_iq1_s(const float* x, block_iq1_s* y, int64_t n, const float* w);
void dequantize_row_iq2_xxs(const block_iq2_xxs* x, float* y, int64_t n);
void dequantize_row_iq2_xs(const block_iq2_xs* x, float* y, int64_t n);
void dequantize_row_iq1_s(const block_iq1_s* x, float* y, int64_t n);

}  // namespace bigdl
