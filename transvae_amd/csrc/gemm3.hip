// tvae_gemm, dtype TVAE_F32X3: fp32-accurate GEMM on the bf16 matrix cores.
//
// gfx950 runs fp32-input MFMA at 1/16 of the bf16 rate.  Every fp32 operand is therefore split
// EXACTLY into three bf16 values, x = hi + mid + lo (hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid); the two subtractions are exact in fp32 and the remainder after lo is
// below 2^-26 |x|), and the product is formed from the six partial products that reach 2^-24:
//     a b  ~=  a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_mid b_mid + a_hi b_lo + a_lo b_hi)
// with v_mfma_f32_32x32x16_bf16 (bf16 x bf16 products are exact in fp32; accumulation is fp32).
// The dropped terms (mid*lo, lo*lo) are below 2^-26 |a b|, i.e. under fp32's own rounding step.
// Six bf16 MFMAs do the work of eight fp32 MFMAs in 3/8 of the matrix-core time.
//
// Tile 128x128x16 per 256-thread workgroup (4 waves as 2x2 of 64x64), double-buffered LDS:
// per operand three bf16 planes [row][k] (row pitch 24 elements = 48 B: conflict-free b128
// fragment reads).  Every thread stages 8 consecutive k of one row per operand and k-step -- two
// float4 loads when k is contiguous in memory, eight coalesced dword loads when the operand is
// k-major -- splits them in registers and writes one ds_write_b128 per plane, so both layouts land
// in the same [row][k] tiles and the MFMA loop is layout-blind.
//
// Requires 16-byte aligned operands (the host falls back to the exact-fp32 kernel otherwise, and
// for a k-major A with the A-side dropout mask, whose hash groups run along m).
#include <stdlib.h>
#include "common.h"
#include "gemm_common.h"

namespace {

constexpr int BK3 = 16;
constexpr int PITCH = BK3 + 8;                         // bf16 elements per LDS row
constexpr int PLANE = BM * PITCH;                      // elements per plane
constexpr int OPER = 3 * PLANE;                        // elements per operand tile (hi, mid, lo)
constexpr int STAGE = 2 * OPER;                        // A + B
static_assert(2 * STAGE * 2 >= EPILOGUE_LDS_FLOATS * 4, "epilogue slabs must fit the staging buffers");

struct Stage8 { float v[8]; };

// global -> registers: 8 consecutive k of row (row0 + tid/2 | tid%128) -- loads only; out-of-range
// pieces read a clamped (valid) address and are zeroed in fix8, after the MFMAs of the current tile
template <bool KM>
__device__ inline void load8(Stage8& s, const float* __restrict__ P, int ld, int row0, int nrows, int k0, int kend,
                             int tid) {
  if (!KM) {
    const int row = tid >> 1, kh = (tid & 1) * 8;
    const int gr = row0 + row, gk = k0 + kh;
    const float* src = P + (size_t)(gr < nrows ? gr : 0) * ld;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = gk + 4 * q;
      float4 t = *reinterpret_cast<const float4*>(src + (k < kend ? k : 0));
      s.v[4 * q] = t.x; s.v[4 * q + 1] = t.y; s.v[4 * q + 2] = t.z; s.v[4 * q + 3] = t.w;
    }
  } else {
    const int row = tid & 127, kh = (tid >> 7) * 8;
    const int gr = row0 + row;
    const float* src = P + (gr < nrows ? gr : 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + kh + j;
      s.v[j] = src[(size_t)(k < kend ? k : 0) * ld];
    }
  }
}

template <bool KM, bool DROP>
__device__ inline void fix8(Stage8& s, int row0, int nrows, int k0, int kend, int tid, const DropCfg& dc, int dropK) {
  const int row = KM ? (tid & 127) : (tid >> 1), kh = KM ? (tid >> 7) * 8 : (tid & 1) * 8;
  const int gr = row0 + row;
  const bool rok = gr < nrows;
  float ks[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  if (DROP) {       // k-contiguous A only: the 8 elements are two hash groups of the dropped-out tensor
    float a[4], b[4];
    const uint32_t base = (uint32_t)gr * (uint32_t)dropK + (uint32_t)(k0 + kh);
    drop_scale4(dc, base, a);
    drop_scale4(dc, base + 4, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) { ks[e] = a[e]; ks[4 + e] = b[e]; }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) s.v[j] = (rok && k0 + kh + j < kend) ? s.v[j] * ks[j] : 0.f;
}

__device__ inline uint32_t pack_bf16(float a, float b) {
  // the plain casts compile to v_cvt_pk_bf16_f32 (round to nearest even)
  const unsigned short lo = __builtin_bit_cast(unsigned short, (__bf16)a);
  const unsigned short hi = __builtin_bit_cast(unsigned short, (__bf16)b);
  return (uint32_t)lo | ((uint32_t)hi << 16);
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

// one 16-byte LDS store per plane
template <bool KM>
__device__ inline void store_planes(const Split8& o, unsigned short* __restrict__ T, int tid) {
  const int row = KM ? (tid & 127) : (tid >> 1), kh = KM ? (tid >> 7) * 8 : (tid & 1) * 8;
  unsigned short* dst = T + row * PITCH + kh;
  *reinterpret_cast<uint4*>(dst) = make_uint4(o.h[0], o.h[1], o.h[2], o.h[3]);
  *reinterpret_cast<uint4*>(dst + PLANE) = make_uint4(o.m[0], o.m[1], o.m[2], o.m[3]);
  *reinterpret_cast<uint4*>(dst + 2 * PLANE) = make_uint4(o.l[0], o.l[1], o.l[2], o.l[3]);
}

template <bool A_KM, bool B_KM, bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void gemm3_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, hh = lane >> 5;
  const int ntn = (g.N + BN - 1) / BN;
  const int wgid = xcd_tile_id();
  const int tile_n = wgid % ntn, tile_m = wgid / ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = blockIdx.z * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nkt = (kend - kbeg + BK3 - 1) / BK3;

  DropCfg dc = drop_cfg(DROP ? g.rng : nullptr, g.site, g.p);
  DropCfg nodrop = drop_cfg(nullptr, 0, 0.f);

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();

  // fused bias gradient (k-major A): a thread's 8 values belong to one row m
  const bool do_rowsum = A_KM && (g.a_rowsum != nullptr) && (tile_n == 0);
  float rs = 0.f;

  // Register staging is two k-steps deep: a k-step is only ~800 matrix-core cycles, less than a
  // loaded-chip HBM round trip, so the loads of step kt+2 are issued before the MFMAs of step kt
  // and first touched (split + LDS store) after the MFMAs of step kt+1.
  Stage8 sa[2], sb[2];
  auto tileA = [&](int buf) { return lds + buf * STAGE; };
  auto tileB = [&](int buf) { return lds + buf * STAGE + OPER; };
  auto issue = [&](Stage8& ra, Stage8& rb, int k0) {
    load8<A_KM>(ra, g.A, g.lda, m0, g.M, k0, kend, tid);
    load8<B_KM>(rb, g.B, g.ldb, n0, g.N, k0, kend, tid);
  };
  // the VALU half of a commit (range masks, dropout scale, 3-way split) ...
  auto prepare = [&](Stage8& ra, Stage8& rb, Split8& pa, Split8& pb, int k0) {
    fix8<A_KM, DROP>(ra, m0, g.M, k0, kend, tid, dc, g.K);
    fix8<B_KM, false>(rb, n0, g.N, k0, kend, tid, nodrop, 0);
    if (do_rowsum) {
#pragma unroll
      for (int j = 0; j < 8; ++j) rs += ra.v[j];
    }
    split8(ra, pa);
    split8(rb, pb);
  };
  // ... and its LDS half
  auto publish = [&](const Split8& pa, const Split8& pb, int buf) {
    store_planes<A_KM>(pa, tileA(buf), tid);
    store_planes<B_KM>(pb, tileB(buf), tid);
  };
  auto commit = [&](Stage8& ra, Stage8& rb, int buf, int k0) {
    Split8 pa, pb;
    prepare(ra, rb, pa, pb, k0);
    publish(pa, pb, buf);
  };
  // MFMAs of one k-step from LDS buffer `buf`; with FILL the split arithmetic of the NEXT k-step
  // (whose loads landed a step ago) is issued in the gaps: an MFMA holds the SIMD's vector issue
  // for 8 of its 32 cycles, so ~5 VALU instructions ride along with each for free.
  auto mfma_step = [&](int buf, auto fill) {
    const unsigned short* TA = tileA(buf) + (wm * 64 + r) * PITCH + 8 * hh;
    const unsigned short* TB = tileB(buf) + (wn * 64 + r) * PITCH + 8 * hh;
    bf16x8 a[2][3], b[2][3];       // [32-row block][plane]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[i][p] = *reinterpret_cast<const bf16x8*>(TA + 32 * i * PITCH + p * PLANE);
        b[i][p] = *reinterpret_cast<const bf16x8*>(TB + 32 * i * PITCH + p * PLANE);
      }
    fill();
    // smallest terms first; the four accumulators interleave so no MFMA waits on its predecessor
#define TVAE_X3(PA, PB)                                                                                       \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] =     \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA], b[j][PB], acc[i][j], 0, 0, 0);
    TVAE_X3(2, 0) TVAE_X3(0, 2) TVAE_X3(1, 1) TVAE_X3(1, 0) TVAE_X3(0, 1) TVAE_X3(0, 0)
#undef TVAE_X3
  };
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // five VALU
    }
  };
  auto nofill = [] {};

  if (nkt > 0) {
    issue(sa[0], sb[0], kbeg);
    if (nkt > 1) issue(sa[1], sb[1], kbeg + BK3);
    commit(sa[0], sb[0], 0, kbeg);
  }
  __syncthreads();

  // Step kt computes from LDS buffer kt&1; register set kt&1 is free (committed) and takes step
  // kt+2.  Steady state (both a load and a commit every step), two steps per trip so that the
  // register sets are named statically:
  int kt = 0;
  for (; kt + 3 < nkt; kt += 2) {
    {
      Split8 pa, pb;
      issue(sa[0], sb[0], kbeg + (kt + 2) * BK3);
      mfma_step(0, [&] { prepare(sa[1], sb[1], pa, pb, kbeg + (kt + 1) * BK3); });
      interleave();
      publish(pa, pb, 1);
      __syncthreads();
    }
    {
      Split8 pa, pb;
      issue(sa[1], sb[1], kbeg + (kt + 3) * BK3);
      mfma_step(1, [&] { prepare(sa[0], sb[0], pa, pb, kbeg + (kt + 2) * BK3); });
      interleave();
      publish(pa, pb, 0);
      __syncthreads();
    }
  }
  // tail: at most three steps, kt even
  for (; kt < nkt; kt += 2) {
    if (kt + 2 < nkt) issue(sa[0], sb[0], kbeg + (kt + 2) * BK3);
    mfma_step(0, nofill);
    if (kt + 1 < nkt) commit(sa[1], sb[1], 1, kbeg + (kt + 1) * BK3);
    __syncthreads();
    if (kt + 1 < nkt) {
      mfma_step(1, nofill);
      if (kt + 2 < nkt) commit(sa[0], sb[0], 0, kbeg + (kt + 2) * BK3);
      __syncthreads();
    }
  }

  float* fl = reinterpret_cast<float*>(lds);
  if (do_rowsum) {   // every thread is past the last barrier of the k-loop: LDS is free
    fl[tid] = rs;    // thread (row = tid & 127, k-half = tid >> 7)
    __syncthreads();
    if (tid < BM && m0 + tid < g.M) atomicAdd(&g.a_rowsum[m0 + tid], fl[tid] + fl[tid + 128]);
  }
  gemm_epilogue<4>(g, acc, fl, m0, n0, do_rowsum);
}

template <bool A_KM, bool B_KM>
void launch3(const GemmArgs& g, dim3 grid, hipStream_t s, bool drop) {
  if (drop) hipLaunchKernelGGL((gemm3_kernel<A_KM, B_KM, true>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm3_kernel<A_KM, B_KM, false>), grid, dim3(256), 0, s, g);
}

}  // namespace

// called by tvae_gemm (gemm.hip) once shapes and flags are validated and `g` is filled in
void tvae_gemm3_launch(const void* gp, int tiles, int splits, bool a_km, bool b_km, bool drop, hipStream_t s) {
  const GemmArgs& g = *reinterpret_cast<const GemmArgs*>(gp);
  dim3 grid(tiles, 1, splits);
  if (!a_km && !b_km) launch3<false, false>(g, grid, s, drop);
  else if (!a_km && b_km) launch3<false, true>(g, grid, s, drop);
  else if (a_km && b_km) launch3<true, true>(g, grid, s, false);
  else launch3<true, false>(g, grid, s, false);
}
