// Shared device/host helpers for libtvae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/tvae_kernels.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

void tvae_set_error(const char* fmt, ...);

#define TVAE_CHECK(cond, code, ...)              \
  do {                                           \
    if (!(cond)) {                               \
      tvae_set_error(__VA_ARGS__);               \
      return (code);                             \
    }                                            \
  } while (0)

#define TVAE_LAUNCH_CHECK(name)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      tvae_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return TVAE_E_LAUNCH;                                                  \
    }                                                                        \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- counter-based dropout -------------------------------------------------------------
// Decisions come in GROUPS of four consecutive element indices: group g = idx >> 2 hashes to two
// 32-bit words (5 integer multiplies for 4 elements -- v_mul_lo_u32 is quarter rate, and the hash
// sits inside GEMM staging and the attention softmax), element idx & 3 takes a 16-bit field and is
// kept iff field >= p * 65536.  The same (rng, site, idx) always gives the same decision, which is
// what lets the backward kernels regenerate the mask instead of storing it.  mix = "lowbias32".
__host__ __device__ inline uint32_t tvae_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
struct DropCfg {
  uint32_t key;
  uint32_t thresh;  // keep iff 16-bit field >= thresh
  float scale;      // 1 / P(keep)
  bool on;
};
__device__ inline DropCfg drop_cfg(const uint32_t* rng, uint32_t site, float p) {
  DropCfg c;
  c.on = (rng != nullptr) && (p > 0.f);
  c.key = 0; c.thresh = 0; c.scale = 1.f;
  if (c.on) {
    uint32_t seed = rng[0], step = rng[1];
    c.key = tvae_mix32(seed ^ tvae_mix32(step * 0x9E3779B1U + site * 0x85EBCA6BU + 0x1234567U));
    float t = p * 65536.f + 0.5f;
    c.thresh = t >= 65535.f ? 65535U : (uint32_t)t;
    c.scale = 65536.f / (float)(65536U - c.thresh);     // exactly 1 / P(keep): the estimator stays unbiased
  }
  return c;
}
__device__ inline void drop_words(const DropCfg& c, uint32_t group, uint32_t& h1, uint32_t& h2) {
  h1 = tvae_mix32(group * 0x9E3779B1U ^ c.key);
  h2 = tvae_mix32(h1 ^ 0x3C6EF372U);
}
// keep flags of the four elements of `group` as bits 0..3
__device__ inline uint32_t drop_keep4(const DropCfg& c, uint32_t group) {
  uint32_t h1, h2;
  drop_words(c, group, h1, h2);
  return ((h1 & 0xFFFFU) >= c.thresh ? 1U : 0U) | ((h1 >> 16) >= c.thresh ? 2U : 0U) |
         ((h2 & 0xFFFFU) >= c.thresh ? 4U : 0U) | ((h2 >> 16) >= c.thresh ? 8U : 0U);
}
__device__ inline float drop_scale(const DropCfg& c, uint32_t idx) {
  if (!c.on) return 1.f;
  uint32_t h1, h2;
  drop_words(c, idx >> 2, h1, h2);
  const uint32_t w = (idx & 2U) ? h2 : h1;
  const uint32_t f = (idx & 1U) ? (w >> 16) : (w & 0xFFFFU);
  return f >= c.thresh ? c.scale : 0.f;
}
// keep-scales of elements base .. base+3 (one hash when base is a multiple of 4)
__device__ inline void drop_scale4(const DropCfg& c, uint32_t base, float (&s)[4]) {
  if (!c.on) { s[0] = s[1] = s[2] = s[3] = 1.f; return; }
  if ((base & 3U) == 0U) {
    uint32_t h1, h2;
    drop_words(c, base >> 2, h1, h2);
    s[0] = (h1 & 0xFFFFU) >= c.thresh ? c.scale : 0.f;
    s[1] = (h1 >> 16) >= c.thresh ? c.scale : 0.f;
    s[2] = (h2 & 0xFFFFU) >= c.thresh ? c.scale : 0.f;
    s[3] = (h2 >> 16) >= c.thresh ? c.scale : 0.f;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = drop_scale(c, base + e);
  }
}

// ---- wave helpers (wave = 64 lanes) ------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// row of a 32x32 MFMA accumulator register: reg r (0..15) on lane half hh (lane>>5)
__device__ inline int crow(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__device__ inline floatx16 zero16() {
  floatx16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
