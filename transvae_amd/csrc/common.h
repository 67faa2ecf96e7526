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
// keep(idx) = mix(idx ^ key) >= p * 2^32 with key = mix(seed, step, site).  The same
// (rng, site, idx) always gives the same decision, which is what lets the backward kernels
// regenerate the mask.  mix = "lowbias32" integer finaliser.
__host__ __device__ inline uint32_t tvae_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
struct DropCfg {
  uint32_t key;
  uint32_t thresh;  // keep iff hash >= thresh
  float scale;      // 1/(1-p)
  bool on;
};
__device__ inline DropCfg drop_cfg(const uint32_t* rng, uint32_t site, float p) {
  DropCfg c;
  c.on = (rng != nullptr) && (p > 0.f);
  c.key = 0; c.thresh = 0; c.scale = 1.f;
  if (c.on) {
    uint32_t seed = rng[0], step = rng[1];
    c.key = tvae_mix32(seed ^ tvae_mix32(step * 0x9E3779B1U + site * 0x85EBCA6BU + 0x1234567U));
    double t = (double)p * 4294967296.0;
    c.thresh = t >= 4294967295.0 ? 0xFFFFFFFFU : (uint32_t)t;
    c.scale = 1.f / (1.f - p);
  }
  return c;
}
__device__ inline float drop_scale(const DropCfg& c, uint32_t idx) {
  if (!c.on) return 1.f;
  return tvae_mix32(idx * 0x9E3779B1U ^ c.key) >= c.thresh ? c.scale : 0.f;
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
