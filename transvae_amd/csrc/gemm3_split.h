// The exact 3-way bf16 split of the f32x3 arithmetic (gemm3.hip, gemm3_dma.hip) and the layout of pre-split weights.
#pragma once
#include "common.h"

namespace {

struct Stage8 { float v[8]; };

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pack_bf16(float a, float b) {
  // one v_cvt_pk_bf16_f32 (round to nearest even) for the pair; two scalar casts end up as two
  // single conversions plus a re-pack
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ inline float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ inline float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000U); }

// 8 fp32 values -> the hi / mid / lo bf16 planes (pairs packed in 32-bit words)
struct Split8 { uint32_t h[4], m[4], l[4]; };

__device__ inline void split8(const Stage8& s, Split8& o) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = s.v[2 * q], x1 = s.v[2 * q + 1];
    o.h[q] = pack_bf16(x0, x1);
    const float r0 = x0 - bf16_lo(o.h[q]), r1 = x1 - bf16_hi(o.h[q]);        // exact
    o.m[q] = pack_bf16(r0, r1);
    const float t0 = r0 - bf16_lo(o.m[q]), t1 = r1 - bf16_hi(o.m[q]);        // exact
    o.l[q] = pack_bf16(t0, t1);
  }
}

// ---- pre-split weights (tvae_split3, TVAE_GEMM_B_SPLIT3) -----------------------------------------------------------
// Tile-major: the planes of a weight (rows n, contraction index k, K a multiple of 16) are stored as blocks of
// 64 rows x 16 k, block (n / 64, k / 16) at ((n / 64) * (K / 16) + k / 16) * 6 KB, and inside a block six 1 KB pieces
// [k-half g = (k / 8) % 2][plane p = hi, mid, lo][row n % 64][8 bf16]: one piece is exactly what one LDS-DMA instruction moves
// (64 lanes x 16 bytes, consecutive lanes = consecutive rows) and lands in LDS as [row][16 bytes], where the MFMA fragment
// reads of 16 lanes hit 16 different 16-byte slots.  Rows are padded with zeros to a multiple of 64.
constexpr int SPLIT_PIECE = 512;                     // bf16 elements per piece (1 KB)
constexpr int SPLIT_BLOCK = 6 * SPLIT_PIECE;         // per (64 rows, 16 k)
__host__ __device__ inline size_t split_index(int n, int k, int p, int kt_total) {     // element index of (row n, k, plane p)
  return ((size_t)(n >> 6) * kt_total + (k >> 4)) * SPLIT_BLOCK + (((k >> 3) & 1) * 3 + p) * SPLIT_PIECE + (n & 63) * 8 + (k & 7);
}

}  // namespace
