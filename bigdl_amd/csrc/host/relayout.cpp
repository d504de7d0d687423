// Host block layout  <->  gfx950 kernel layout ("T16").
//
// Role of the reference's ggml_q_format_convet_cpu2xpu / xpu2cpu
// (python/llm/src/ipex_llm/ggml/model/llama/llama_cpp.py:1102-1135, used from
// python/llm/src/ipex_llm/transformers/low_bit_linear.py:178-233): checkpoints keep the
// ggml block order, the device keeps an order chosen for the MI355X kernels.
//
// T16 layout of a [N, K] weight (N padded to 16, K padded to 128; padding blocks carry
// scale 0):
//   code planes are cut into tiles of 16 rows x 128 k.  One wavefront owns a tile; lane
//   l = 16*g + r holds row r and, for MFMA step s (0..3), the 8 codes of
//   k = 128*kt + 32*s + 8*g + j  (j = 0..7).  So MFMA step s of a tile covers exactly the
//   32-element scale block s of each of the 16 rows, and a lane fetches its share of the
//   tile with one 16-byte load (4-bit codes) or two (8-bit codes).
//   * 4-bit plane: lane data = 4 dwords (one per s); inside a dword code j sits in nibble
//     P4[j] = {0,4,1,5,2,6,3,7}[j] so that (w >> 4i) & 0x000F000F yields codes (2i, 2i+1)
//     as the two halves of a packed-half register.
//   * 8-bit plane: lane data = 8 dwords; dwords 2s, 2s+1 hold codes j = 0..7 in byte order.
//     Tile = [2 halves][64 lanes][16 B]; half h holds steps s = 2h, 2h+1.
//   * 1-bit / 2-bit high planes (5- and 6-bit formats): lane data = 4 bytes / 8 bytes,
//     byte(s) s holding the high bits of codes j = 0..7 (bit j, or bits 2j..2j+1).
//   scale (and min) planes: fp16 [N/16][K/128][16 rows][4 blocks]; formats whose host block
//   is wider than 32 (nf4: 64, woq_int4: 128) replicate their scale per 32.
// Plane order in the device buffer: codes, high bits (if any), scales, mins (if any).
#include "qformats.h"
#include <algorithm>
#include <vector>

namespace bigdl {

struct DevFormat {
  int lo_bits;   // 4 or 8
  int hi_bits;   // 0, 1 or 2
  bool has_min;
  int host_qk;
};

static bool dev_format(int qtype, DevFormat* f) {
  switch (qtype) {
    case Q_SYM_INT4: case Q_FP4: *f = {4, 0, false, 32}; return true;
    case Q_ASYM_INT4: *f = {4, 0, true, 32}; return true;
    case Q_NF4: *f = {4, 0, false, 64}; return true;
    case Q_WOQ_INT4: *f = {4, 0, true, 128}; return true;
    case Q_SYM_INT5: *f = {4, 1, false, 32}; return true;
    case Q_ASYM_INT5: *f = {4, 1, true, 32}; return true;
    case Q_FP6: *f = {4, 2, false, 32}; return true;
    case Q_SYM_INT8: case Q_FP8_E4M3: case Q_FP8_E5M2: *f = {8, 0, false, 32}; return true;
    // k-quants whose sub-blocks are 32 wide map onto the asymmetric tiled layouts: per 32 elements
    // scale = d * sc6 and min = -dmin * m6, rounded to fp16 (relative error 2^-11, far below the 4-/5-bit step)
    case Q_Q4_K: *f = {4, 0, true, 256}; return true;
    case Q_Q5_K: *f = {4, 1, true, 256}; return true;
    default: return false;
  }
}

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

int64_t device_nbytes(int qtype, int64_t N, int64_t K) {
  DevFormat f;
  if (!dev_format(qtype, &f)) return -1;
  const int64_t Np = round_up(N, 16), Kp = round_up(K, 128);
  int64_t b = Np * Kp * f.lo_bits / 8;
  b += Np * Kp * f.hi_bits / 8;
  b += Np * (Kp / 32) * 2;
  if (f.has_min) b += Np * (Kp / 32) * 2;
  return b;
}

static inline void k4_scale_min(int j, const uint8_t* q, uint8_t* d, uint8_t* m) {
  if (j < 4) {
    *d = q[j] & 63;
    *m = q[j + 4] & 63;
  } else {
    *d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4);
    *m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4);
  }
}

// ---- host row <-> (codes, per-32 scale, per-32 min) ----
static void unpack_row(int qtype, const uint8_t* row, int64_t K, uint8_t* codes, fp16_t* sc, fp16_t* mn) {
  switch (qtype) {
    case Q_SYM_INT4: case Q_FP4: {
      const block_q4_0* b = (const block_q4_0*)row;
      for (int64_t i = 0; i < K / 32; ++i) {
        sc[i] = b[i].d;
        for (int j = 0; j < 16; ++j) { codes[i * 32 + j] = b[i].qs[j] & 15; codes[i * 32 + j + 16] = b[i].qs[j] >> 4; }
      }
      break;
    }
    case Q_ASYM_INT4: {
      const block_q4_1* b = (const block_q4_1*)row;
      for (int64_t i = 0; i < K / 32; ++i) {
        sc[i] = b[i].d; mn[i] = b[i].m;
        for (int j = 0; j < 16; ++j) { codes[i * 32 + j] = b[i].qs[j] & 15; codes[i * 32 + j + 16] = b[i].qs[j] >> 4; }
      }
      break;
    }
    case Q_NF4: {
      const block_nf4* b = (const block_nf4*)row;
      for (int64_t i = 0; i < K / 64; ++i) {
        sc[2 * i] = sc[2 * i + 1] = b[i].d;
        for (int j = 0; j < 32; ++j) { codes[i * 64 + j] = b[i].qs[j] & 15; codes[i * 64 + j + 32] = b[i].qs[j] >> 4; }
      }
      break;
    }
    case Q_WOQ_INT4: {
      const block_woq4* b = (const block_woq4*)row;
      for (int64_t i = 0; i < K / 128; ++i) {
        for (int t = 0; t < 4; ++t) { sc[4 * i + t] = b[i].d; mn[4 * i + t] = b[i].m; }
        for (int j = 0; j < 64; ++j) { codes[i * 128 + j] = b[i].qs[j] & 15; codes[i * 128 + j + 64] = b[i].qs[j] >> 4; }
      }
      break;
    }
    case Q_SYM_INT5: {
      const block_q5_0* b = (const block_q5_0*)row;
      for (int64_t i = 0; i < K / 32; ++i) {
        sc[i] = b[i].d;
        uint32_t qh; std::memcpy(&qh, b[i].qh, 4);
        for (int j = 0; j < 16; ++j) {
          codes[i * 32 + j] = (b[i].qs[j] & 15) | (((qh >> j) & 1) << 4);
          codes[i * 32 + j + 16] = (b[i].qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4);
        }
      }
      break;
    }
    case Q_ASYM_INT5: {
      const block_q5_1* b = (const block_q5_1*)row;
      for (int64_t i = 0; i < K / 32; ++i) {
        sc[i] = b[i].d; mn[i] = b[i].m;
        uint32_t qh; std::memcpy(&qh, b[i].qh, 4);
        for (int j = 0; j < 16; ++j) {
          codes[i * 32 + j] = (b[i].qs[j] & 15) | (((qh >> j) & 1) << 4);
          codes[i * 32 + j + 16] = (b[i].qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4);
        }
      }
      break;
    }
    case Q_FP6: {
      const block_fp6* b = (const block_fp6*)row;
      for (int64_t i = 0; i < K / 32; ++i) {
        sc[i] = b[i].d;
        for (int j = 0; j < 16; ++j) { codes[i * 32 + j] = b[i].ql[j] & 15; codes[i * 32 + j + 16] = b[i].ql[j] >> 4; }
        for (int j = 0; j < 32; ++j) codes[i * 32 + j] |= ((b[i].qh[j / 4] >> (2 * (j % 4))) & 3) << 4;
      }
      break;
    }
    case Q_SYM_INT8: case Q_FP8_E4M3: case Q_FP8_E5M2: {
      const block_fp8* b = (const block_fp8*)row;
      for (int64_t i = 0; i < K / 32; ++i) {
        sc[i] = b[i].d;
        std::memcpy(codes + i * 32, b[i].qs, 32);
      }
      break;
    }
    case Q_Q4_K: case Q_Q5_K: {
      // super-block of 256 = 8 sub-blocks of 32; byte l of 32-byte group p holds sub-block 2p (low nibble) and
      // 2p + 1 (high nibble); q5_k adds bit (2p) / (2p + 1) of qh[l] as the fifth code bit
      for (int64_t i = 0; i < K / 256; ++i) {
        const uint8_t* scales; const uint8_t* qs; const uint8_t* qh = nullptr;
        float d, dmin;
        if (qtype == Q_Q4_K) {
          const block_q4_K* b = (const block_q4_K*)row + i;
          d = fp16_to_fp32(b->d); dmin = fp16_to_fp32(b->dmin); scales = b->scales; qs = b->qs;
        } else {
          const block_q5_K* b = (const block_q5_K*)row + i;
          d = fp16_to_fp32(b->d); dmin = fp16_to_fp32(b->dmin); scales = b->scales; qs = b->qs; qh = b->qh;
        }
        for (int sb = 0; sb < 8; ++sb) {
          uint8_t s6, m6;
          k4_scale_min(sb, scales, &s6, &m6);
          sc[i * 8 + sb] = fp32_to_fp16(d * s6);
          mn[i * 8 + sb] = fp32_to_fp16(-dmin * m6);
          const uint8_t* q = qs + (sb / 2) * 32;
          for (int l = 0; l < 32; ++l) {
            uint8_t c = (sb & 1) ? (q[l] >> 4) : (q[l] & 15);
            if (qh) c |= ((qh[l] >> sb) & 1) << 4;
            codes[i * 256 + sb * 32 + l] = c;
          }
        }
      }
      break;
    }
    default: break;
  }
}

static void pack_row(int qtype, uint8_t* row, int64_t K, const uint8_t* codes, const fp16_t* sc, const fp16_t* mn) {
  switch (qtype) {
    case Q_SYM_INT4: case Q_FP4: {
      block_q4_0* b = (block_q4_0*)row;
      for (int64_t i = 0; i < K / 32; ++i) {
        b[i].d = sc[i];
        for (int j = 0; j < 16; ++j) b[i].qs[j] = codes[i * 32 + j] | (codes[i * 32 + j + 16] << 4);
      }
      break;
    }
    case Q_ASYM_INT4: {
      block_q4_1* b = (block_q4_1*)row;
      for (int64_t i = 0; i < K / 32; ++i) {
        b[i].d = sc[i]; b[i].m = mn[i];
        for (int j = 0; j < 16; ++j) b[i].qs[j] = codes[i * 32 + j] | (codes[i * 32 + j + 16] << 4);
      }
      break;
    }
    case Q_NF4: {
      block_nf4* b = (block_nf4*)row;
      for (int64_t i = 0; i < K / 64; ++i) {
        b[i].d = sc[2 * i];
        for (int j = 0; j < 32; ++j) b[i].qs[j] = codes[i * 64 + j] | (codes[i * 64 + j + 32] << 4);
      }
      break;
    }
    case Q_WOQ_INT4: {
      block_woq4* b = (block_woq4*)row;
      for (int64_t i = 0; i < K / 128; ++i) {
        b[i].d = sc[4 * i]; b[i].m = mn[4 * i];
        for (int j = 0; j < 64; ++j) b[i].qs[j] = codes[i * 128 + j] | (codes[i * 128 + j + 64] << 4);
      }
      break;
    }
    case Q_SYM_INT5: case Q_ASYM_INT5: {
      for (int64_t i = 0; i < K / 32; ++i) {
        uint32_t qh = 0;
        uint8_t qs[16];
        for (int j = 0; j < 16; ++j) {
          const uint8_t c0 = codes[i * 32 + j], c1 = codes[i * 32 + j + 16];
          qs[j] = (c0 & 15) | ((c1 & 15) << 4);
          qh |= (uint32_t)(c0 >> 4) << j;
          qh |= (uint32_t)(c1 >> 4) << (j + 16);
        }
        if (qtype == Q_SYM_INT5) {
          block_q5_0* b = (block_q5_0*)row;
          b[i].d = sc[i]; std::memcpy(b[i].qh, &qh, 4); std::memcpy(b[i].qs, qs, 16);
        } else {
          block_q5_1* b = (block_q5_1*)row;
          b[i].d = sc[i]; b[i].m = mn[i]; std::memcpy(b[i].qh, &qh, 4); std::memcpy(b[i].qs, qs, 16);
        }
      }
      break;
    }
    case Q_FP6: {
      block_fp6* b = (block_fp6*)row;
      for (int64_t i = 0; i < K / 32; ++i) {
        b[i].d = sc[i];
        std::memset(b[i].qh, 0, 8);
        for (int j = 0; j < 16; ++j) b[i].ql[j] = (codes[i * 32 + j] & 15) | ((codes[i * 32 + j + 16] & 15) << 4);
        for (int j = 0; j < 32; ++j) b[i].qh[j / 4] |= (codes[i * 32 + j] >> 4) << (2 * (j % 4));
      }
      break;
    }
    case Q_SYM_INT8: case Q_FP8_E4M3: case Q_FP8_E5M2: {
      block_fp8* b = (block_fp8*)row;
      for (int64_t i = 0; i < K / 32; ++i) {
        b[i].d = sc[i];
        std::memcpy(b[i].qs, codes + i * 32, 32);
      }
      break;
    }
    default: break;
  }
}

static const int P4[8] = {0, 4, 1, 5, 2, 6, 3, 7};

struct Planes {
  uint8_t* lo; uint8_t* hi; fp16_t* sc; fp16_t* mn;
};

static Planes carve(const DevFormat& f, uint8_t* base, int64_t Np, int64_t Kp) {
  Planes p;
  p.lo = base;
  uint8_t* cur = base + Np * Kp * f.lo_bits / 8;
  p.hi = f.hi_bits ? cur : nullptr;
  cur += Np * Kp * f.hi_bits / 8;
  p.sc = (fp16_t*)cur;
  cur += Np * (Kp / 32) * 2;
  p.mn = f.has_min ? (fp16_t*)cur : nullptr;
  return p;
}

// returns 0 on success
int relayout_to_device(int qtype, const uint8_t* src, uint8_t* dst, int64_t N, int64_t K) {
  DevFormat f;
  if (!dev_format(qtype, &f)) return -1;
  if (K % f.host_qk != 0) return -2;
  const int64_t Np = round_up(N, 16), Kp = round_up(K, 128);
  const int64_t row_bytes = K / qk_size(qtype) * type_size(qtype);
  const int64_t total = device_nbytes(qtype, N, K);
  std::memset(dst, 0, (size_t)total);
  Planes p = carve(f, dst, Np, Kp);
  const int64_t KT = Kp / 128;
#pragma omp parallel
  {
    std::vector<uint8_t> codes(Kp);
    std::vector<fp16_t> sc(Kp / 32), mn(Kp / 32);
#pragma omp for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
      std::fill(codes.begin(), codes.end(), 0);
      std::fill(sc.begin(), sc.end(), 0);
      std::fill(mn.begin(), mn.end(), 0);
      unpack_row(qtype, src + n * row_bytes, K, codes.data(), sc.data(), mn.data());
      const int64_t nt = n / 16, r = n % 16;
      for (int64_t kt = 0; kt < KT; ++kt) {
        const int64_t tile = nt * KT + kt;
        for (int g = 0; g < 4; ++g) {
          const int lane = 16 * g + (int)r;
          for (int s = 0; s < 4; ++s) {
            const uint8_t* c = codes.data() + kt * 128 + 32 * s + 8 * g;
            if (f.lo_bits == 4) {
              uint32_t w = 0;
              for (int j = 0; j < 8; ++j) w |= (uint32_t)(c[j] & 15) << (4 * P4[j]);
              std::memcpy(p.lo + tile * 1024 + lane * 16 + s * 4, &w, 4);
            } else {
              uint8_t* d8 = p.lo + tile * 2048 + (s / 2) * 1024 + lane * 16 + (s % 2) * 8;
              std::memcpy(d8, c, 8);
            }
            if (f.hi_bits == 1) {
              uint8_t h = 0;
              for (int j = 0; j < 8; ++j) h |= (uint8_t)(((c[j] >> 4) & 1) << j);
              p.hi[tile * 256 + lane * 4 + s] = h;
            } else if (f.hi_bits == 2) {
              uint16_t h = 0;
              for (int j = 0; j < 8; ++j) h |= (uint16_t)(((c[j] >> 4) & 3) << (2 * j));
              std::memcpy(p.hi + tile * 512 + lane * 8 + s * 2, &h, 2);
            }
          }
        }
        for (int s = 0; s < 4; ++s) {
          p.sc[tile * 64 + r * 4 + s] = sc[kt * 4 + s];
          if (p.mn) p.mn[tile * 64 + r * 4 + s] = mn[kt * 4 + s];
        }
      }
    }
  }
  return 0;
}

int relayout_to_host(int qtype, const uint8_t* src, uint8_t* dst, int64_t N, int64_t K) {
  DevFormat f;
  if (!dev_format(qtype, &f)) return -1;
  if (qtype == Q_Q4_K || qtype == Q_Q5_K) return -3;   // the 6-bit sub-scales are folded on the way in: not invertible
  if (K % f.host_qk != 0) return -2;
  const int64_t Np = round_up(N, 16), Kp = round_up(K, 128);
  const int64_t row_bytes = K / qk_size(qtype) * type_size(qtype);
  Planes p = carve(f, const_cast<uint8_t*>(src), Np, Kp);
  const int64_t KT = Kp / 128;
#pragma omp parallel
  {
    std::vector<uint8_t> codes(Kp);
    std::vector<fp16_t> sc(Kp / 32), mn(Kp / 32);
#pragma omp for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
      const int64_t nt = n / 16, r = n % 16;
      for (int64_t kt = 0; kt < KT; ++kt) {
        const int64_t tile = nt * KT + kt;
        for (int g = 0; g < 4; ++g) {
          const int lane = 16 * g + (int)r;
          for (int s = 0; s < 4; ++s) {
            uint8_t* c = codes.data() + kt * 128 + 32 * s + 8 * g;
            if (f.lo_bits == 4) {
              uint32_t w;
              std::memcpy(&w, p.lo + tile * 1024 + lane * 16 + s * 4, 4);
              for (int j = 0; j < 8; ++j) c[j] = (w >> (4 * P4[j])) & 15;
            } else {
              std::memcpy(c, p.lo + tile * 2048 + (s / 2) * 1024 + lane * 16 + (s % 2) * 8, 8);
            }
            if (f.hi_bits == 1) {
              const uint8_t h = p.hi[tile * 256 + lane * 4 + s];
              for (int j = 0; j < 8; ++j) c[j] |= ((h >> j) & 1) << 4;
            } else if (f.hi_bits == 2) {
              uint16_t h;
              std::memcpy(&h, p.hi + tile * 512 + lane * 8 + s * 2, 2);
              for (int j = 0; j < 8; ++j) c[j] |= ((h >> (2 * j)) & 3) << 4;
            }
          }
        }
        for (int s = 0; s < 4; ++s) {
          sc[kt * 4 + s] = p.sc[tile * 64 + r * 4 + s];
          if (p.mn) mn[kt * 4 + s] = p.mn[tile * 64 + r * 4 + s];
        }
      }
      pack_row(qtype, dst + n * row_bytes, K, codes.data(), sc.data(), mn.data());
    }
  }
  return 0;
}

}  // namespace bigdl
