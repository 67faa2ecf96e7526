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
#include <type_traits>
#include "common.h"
#include "gemm_common.h"
#include "gemm3_split.h"

namespace {

constexpr int BK3 = 16;
constexpr int PITCH = BK3 + 8;                         // bf16 elements per LDS row
constexpr int PLANE = BM * PITCH;                      // elements per plane
constexpr int OPER = 3 * PLANE;                        // elements per operand tile (hi, mid, lo)
constexpr int STAGE = 2 * OPER;                        // A + B
static_assert(2 * STAGE * 2 >= EPILOGUE_LDS_FLOATS * 4, "epilogue slabs must fit the staging buffers");

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

// `inside` (workgroup-uniform): every row of the tile and every k of the step exist -- no masking
template <bool KM, bool DROP>
__device__ inline void fix8(Stage8& s, int row0, int nrows, int k0, int kend, int tid, const DropCfg& dc, int dropK,
                            bool inside = false) {
  if (!DROP && inside) return;
  const int row = KM ? (tid & 127) : (tid >> 1), kh = KM ? (tid >> 7) * 8 : (tid & 1) * 8;
  const int gr = row0 + row;
  const bool rok = inside || gr < nrows;
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
  for (int j = 0; j < 8; ++j) s.v[j] = (rok && (inside || k0 + kh + j < kend)) ? s.v[j] * ks[j] : 0.f;
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

// pre-split B (TVAE_GEMM_B_SPLIT3): the tile-major planes of gemm3_split.h, ldb = 3 * Kb elements (Kb = the K the weight was
// split with, a multiple of 16) -- three 16-byte loads per thread and k-step, 1 KB apart, no arithmetic; consecutive row
// threads read consecutive 16-byte pieces
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native vector: HIP's uint4 struct went through scratch here
struct Planes8 { u32x4 h, m, l; };

__device__ inline void load_planes(Planes8& o, const unsigned short* __restrict__ B, int ldb, int n0, int N, int k0,
                                   int kend, int tid) {
  const int row = tid >> 1, kh = (tid & 1) * 8;
  const int gn = n0 + row, gk = k0 + kh;
  const bool ok = gn < N && gk < kend;
  const unsigned short* src = B + split_index(ok ? gn : 0, ok ? gk : 0, 0, ldb / 48);
  o.h = *reinterpret_cast<const u32x4*>(src);
  o.m = *reinterpret_cast<const u32x4*>(src + SPLIT_PIECE);
  o.l = *reinterpret_cast<const u32x4*>(src + 2 * SPLIT_PIECE);
}
__device__ inline void store_loaded_planes(const Planes8& o, unsigned short* __restrict__ T, int n0, int N, int k0, int kend,
                                           int tid) {
  const int row = tid >> 1, kh = (tid & 1) * 8;
  const bool ok = n0 + row < N && k0 + kh < kend;
  const u32x4 z = {0u, 0u, 0u, 0u};
  unsigned short* dst = T + row * PITCH + kh;
  *reinterpret_cast<u32x4*>(dst) = ok ? o.h : z;
  *reinterpret_cast<u32x4*>(dst + PLANE) = ok ? o.m : z;
  *reinterpret_cast<u32x4*>(dst + 2 * PLANE) = ok ? o.l : z;
}

// ---- interior tiles ------------------------------------------------------------------------------
// A tile whose 128 rows of both operands exist and whose k range is whole k-steps needs no range
// test at all, and its addresses are (workgroup-uniform base, advanced on the scalar unit) + (a
// per-thread byte offset fixed for the whole k-loop): the loads take the scalar-base form and the
// k-step carries no address or mask arithmetic on the vector unit, which the split shares with the
// MFMA issue.  (The general path above spends ~60 of its ~115 vector instructions per k-step there.)
template <bool KM>
__device__ inline uint32_t fast_off(int ld, int tid) {          // bytes
  if (!KM) return (uint32_t)(((tid >> 1) * ld + (tid & 1) * 8) * 4);
  return (uint32_t)((((tid >> 7) * 8) * ld + (tid & 127)) * 4);
}
template <bool KM>
__device__ inline void load8_fast(Stage8& s, const float* __restrict__ base, int ld, uint32_t off) {
  const char* b = reinterpret_cast<const char*>(base);
  if (!KM) {
    const float4 t0 = *reinterpret_cast<const float4*>(b + off);
    const float4 t1 = *reinterpret_cast<const float4*>(b + off + 16);
    s.v[0] = t0.x; s.v[1] = t0.y; s.v[2] = t0.z; s.v[3] = t0.w;
    s.v[4] = t1.x; s.v[5] = t1.y; s.v[6] = t1.z; s.v[7] = t1.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) s.v[j] = *reinterpret_cast<const float*>(b + (size_t)j * ld * 4 + off);
  }
}
__device__ inline void load_planes_fast(Planes8& o, const unsigned short* __restrict__ base, uint32_t off) {
  const char* b = reinterpret_cast<const char*>(base);
  o.h = *reinterpret_cast<const u32x4*>(b + off);
  o.m = *reinterpret_cast<const u32x4*>(b + off + 2 * SPLIT_PIECE);
  o.l = *reinterpret_cast<const u32x4*>(b + off + 4 * SPLIT_PIECE);
}
__device__ inline void store_loaded_planes_fast(const Planes8& o, unsigned short* __restrict__ T, int tid) {
  unsigned short* dst = T + (tid >> 1) * PITCH + (tid & 1) * 8;
  *reinterpret_cast<u32x4*>(dst) = o.h;
  *reinterpret_cast<u32x4*>(dst + PLANE) = o.m;
  *reinterpret_cast<u32x4*>(dst + 2 * PLANE) = o.l;
}

template <bool A_KM, bool B_KM, bool DROP, bool B_SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void gemm3_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, hh = lane >> 5;
  const int ntn = (g.N + BN - 1) / BN;
  int wgid;
  tile_and_split(g, wgid);
  const int tile_n = wgid % ntn, tile_m = wgid / ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = g.split * g.k_per_split;
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
  Planes8 sp0, sp1;
  const unsigned short* B16 = reinterpret_cast<const unsigned short*>(g.B);
  auto tileA = [&](int buf) { return lds + buf * STAGE; };
  auto tileB = [&](int buf) { return lds + buf * STAGE + OPER; };
  const bool a_rows_in = m0 + BM <= g.M, b_rows_in = n0 + BN <= g.N;
  // interior tile (workgroup-uniform): see load8_fast
  const bool fast_tile = a_rows_in && b_rows_in && nkt > 0 && ((kend - kbeg) % BK3 == 0);
  const uint32_t offA = fast_off<A_KM>(g.lda, tid);
  const uint32_t offB = B_SPLIT ? (uint32_t)(split_index(tid >> 1, (tid & 1) * 8, 0, g.ldb / 48) * 2) : fast_off<B_KM>(g.ldb, tid);
  const float* baseA = A_KM ? g.A + (size_t)kbeg * g.lda + m0 : g.A + (size_t)m0 * g.lda + kbeg;
  const float* baseB = B_KM ? g.B + (size_t)kbeg * g.ldb + n0 : g.B + (size_t)n0 * g.ldb + kbeg;
  const unsigned short* baseP = B16 + (B_SPLIT ? split_index(n0, kbeg, 0, g.ldb / 48) : 0);     // n0 % 128 == 0, kbeg % 16 == 0
  const size_t stepA = A_KM ? (size_t)BK3 * g.lda : (size_t)BK3, stepB = B_KM ? (size_t)BK3 * g.ldb : (size_t)BK3;

  // MFMAs of one k-step from LDS buffer `buf`; with a filler the split arithmetic of the NEXT k-step
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
  auto nofill = [] {};

  // the whole pipelined k-loop, once per addressing mode (FAST = interior tile)
  auto k_loop = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    constexpr int VALU_PER_GAP = FAST ? 3 : 5;
    auto issue = [&](Stage8& ra, Stage8& rb, Planes8& rp, int step) {
      const int k0 = kbeg + step * BK3;
      if (FAST) {
        load8_fast<A_KM>(ra, baseA + step * stepA, g.lda, offA);
        if (B_SPLIT) load_planes_fast(rp, baseP + (size_t)step * SPLIT_BLOCK, offB);
        else load8_fast<B_KM>(rb, baseB + step * stepB, g.ldb, offB);
      } else {
        load8<A_KM>(ra, g.A, g.lda, m0, g.M, k0, kend, tid);
        if (B_SPLIT) load_planes(rp, B16, g.ldb, n0, g.N, k0, kend, tid);
        else load8<B_KM>(rb, g.B, g.ldb, n0, g.N, k0, kend, tid);
      }
    };
    // the VALU half of a commit (range masks, dropout scale, 3-way split) ...
    auto prepare = [&](Stage8& ra, Stage8& rb, Split8& pa, Split8& pb, int step, bool k_in = false) {
      const int k0 = kbeg + step * BK3;
      fix8<A_KM, DROP>(ra, m0, g.M, k0, kend, tid, dc, g.K, FAST || (k_in && a_rows_in));
      if (do_rowsum) {
#pragma unroll
        for (int j = 0; j < 8; ++j) rs += ra.v[j];
      }
      split8(ra, pa);
      if (!B_SPLIT) {
        fix8<B_KM, false>(rb, n0, g.N, k0, kend, tid, nodrop, 0, FAST || (k_in && b_rows_in));
        split8(rb, pb);
      }
    };
    // ... and its LDS half
    auto publish = [&](const Split8& pa, const Split8& pb, const Planes8& rp, int buf, int step) {
      const int k0 = kbeg + step * BK3;
      store_planes<A_KM>(pa, tileA(buf), tid);
      if (B_SPLIT) {
        if (FAST) store_loaded_planes_fast(rp, tileB(buf), tid);
        else store_loaded_planes(rp, tileB(buf), n0, g.N, k0, kend, tid);
      } else {
        store_planes<B_KM>(pb, tileB(buf), tid);
      }
    };
    auto commit = [&](Stage8& ra, Stage8& rb, const Planes8& rp, int buf, int step) {
      Split8 pa, pb;
      prepare(ra, rb, pa, pb, step);
      publish(pa, pb, rp, buf, step);
    };
    auto interleave = [&]() {
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_GAP, 0);   // a few VALU
      }
    };

    if (nkt > 0) {
      issue(sa[0], sb[0], sp0, 0);
      if (nkt > 1) issue(sa[1], sb[1], sp1, 1);
      commit(sa[0], sb[0], sp0, 0, 0);
    }
    __syncthreads();

    // Step kt computes from LDS buffer kt&1; register set kt&1 is free (committed) and takes step
    // kt+2.  Steady state (both a load and a commit every step), two steps per trip so that the
    // register sets are named statically:
    int kt = 0;
    for (; kt + 3 < nkt; kt += 2) {
      {
        Split8 pa, pb;
        issue(sa[0], sb[0], sp0, kt + 2);
        __builtin_amdgcn_sched_barrier(0);     // the loads stay AHEAD of this step's MFMAs (two steps of flight time)
        mfma_step(0, [&] { prepare(sa[1], sb[1], pa, pb, kt + 1, true); });
        interleave();
        publish(pa, pb, sp1, 1, kt + 1);
        __syncthreads();
      }
      {
        Split8 pa, pb;
        issue(sa[1], sb[1], sp1, kt + 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(1, [&] { prepare(sa[0], sb[0], pa, pb, kt + 2, true); });
        interleave();
        publish(pa, pb, sp0, 0, kt + 2);
        __syncthreads();
      }
    }
    // tail: at most three steps, kt even
    for (; kt < nkt; kt += 2) {
      if (kt + 2 < nkt) issue(sa[0], sb[0], sp0, kt + 2);
      mfma_step(0, nofill);
      if (kt + 1 < nkt) commit(sa[1], sb[1], sp1, 1, kt + 1);
      __syncthreads();
      if (kt + 1 < nkt) {
        mfma_step(1, nofill);
        if (kt + 2 < nkt) commit(sa[0], sb[0], sp0, 0, kt + 2);
        __syncthreads();
      }
    }
  };
  if (fast_tile) k_loop(std::true_type{});
  else k_loop(std::false_type{});

  float* fl = reinterpret_cast<float*>(lds);
  if (do_rowsum) {   // every thread is past the last barrier of the k-loop: LDS is free
    fl[tid] = rs;    // thread (row = tid & 127, k-half = tid >> 7)
    __syncthreads();
    if (tid < BM && m0 + tid < g.M) atomicAdd(&g.a_rowsum[m0 + tid], fl[tid] + fl[tid + 128]);
  }
  gemm_epilogue<4>(g, acc, fl, m0, n0, do_rowsum);
}

template <bool A_KM, bool B_KM, bool B_SPLIT>
void launch3(const GemmArgs& g, dim3 grid, hipStream_t s, bool drop) {
  if (drop) hipLaunchKernelGGL((gemm3_kernel<A_KM, B_KM, true, B_SPLIT>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm3_kernel<A_KM, B_KM, false, B_SPLIT>), grid, dim3(256), 0, s, g);
}

// ---- tvae_split3: fp32 matrix -> hi/mid/lo bf16 planes, in the matrix's own orientation and transposed --
// out_n = planes of w (rows r, contraction index c) and out_t = planes of w^T (rows c, contraction index r), both in the
// tile-major layout of gemm3_split.h with their rows padded by zeros to a multiple of 64 (either may be null; C resp. R a
// multiple of 16): what TVAE_GEMM_B_SPLIT3 consumes as the weight operand of the forward (W) and of the data gradient (W^T).
// 32x32 tiles through LDS so that both outputs are written in runs of 8 elements.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ w, int ldw, unsigned short* __restrict__ out_n,
                                                     unsigned short* __restrict__ out_t, int R, int C) {
  __shared__ unsigned short tile[3][32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int Rpad = (R + 63) & ~63, Cpad = (C + 63) & ~63;       // the grid covers the padded extents
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    float x = (r < R && c < C) ? w[(size_t)r * ldw + c] : 0.f;
    const unsigned short h = __builtin_bit_cast(unsigned short, (__bf16)x);
    const float r1 = x - __builtin_bit_cast(float, (uint32_t)h << 16);
    const unsigned short m = __builtin_bit_cast(unsigned short, (__bf16)r1);
    const float r2 = r1 - __builtin_bit_cast(float, (uint32_t)m << 16);
    const unsigned short l = __builtin_bit_cast(unsigned short, (__bf16)r2);
    tile[0][ty + 8 * i][tx] = h; tile[1][ty + 8 * i][tx] = m; tile[2][ty + 8 * i][tx] = l;
    if (out_n && r < Rpad && c < C) {                            // pad rows get the zeros computed above
      unsigned short* o = out_n + split_index(r, c, 0, C >> 4);
      o[0] = h; o[SPLIT_PIECE] = m; o[2 * SPLIT_PIECE] = l;
    }
  }
  __syncthreads();
  if (out_t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i, r = r0 + tx;               // transposed: consecutive lanes -> consecutive r
      if (r < R && c < Cpad) {
        unsigned short* o = out_t + split_index(c, r, 0, R >> 4);
        o[0] = tile[0][tx][ty + 8 * i]; o[SPLIT_PIECE] = tile[1][tx][ty + 8 * i]; o[2 * SPLIT_PIECE] = tile[2][tx][ty + 8 * i];
      }
    }
  }
}

}  // namespace

extern "C" int tvae_split3(const float* w, int ldw, uint16_t* out_n, uint16_t* out_t, int rows, int cols, void* stream) {
  TVAE_CHECK(w && (out_n || out_t), TVAE_E_NULL, "tvae_split3: null pointer");
  TVAE_CHECK(rows > 0 && cols > 0 && ldw >= cols, TVAE_E_SHAPE, "tvae_split3: bad shape (%d x %d, ld %d)", rows, cols, ldw);
  TVAE_CHECK((!out_n || cols % 16 == 0) && (!out_t || rows % 16 == 0), TVAE_E_SHAPE,
             "tvae_split3: the contraction dimension of each output must be a multiple of 16 (%d x %d)", rows, cols);
  hipLaunchKernelGGL(split3_kernel, dim3(cdiv(cols, 64) * 2, cdiv(rows, 64) * 2), dim3(256), 0, (hipStream_t)stream, w, ldw, out_n,
                     out_t, rows, cols);
  TVAE_LAUNCH_CHECK("tvae_split3");
  return TVAE_OK;
}

// called by tvae_gemm (gemm.hip) once shapes and flags are validated and `g` is filled in
void tvae_gemm3_launch(const void* gp, int tiles, int splits, bool a_km, bool b_km, bool drop, bool b_split, hipStream_t s) {
  const GemmArgs& g = *reinterpret_cast<const GemmArgs*>(gp);
  dim3 grid(tiles, 1, splits);
  if (b_split) {                       // B is pre-split, k-contiguous by construction
    if (!a_km) launch3<false, false, true>(g, grid, s, drop);
    else launch3<true, false, true>(g, grid, s, false);
    return;
  }
  if (!a_km && !b_km) launch3<false, false, false>(g, grid, s, drop);
  else if (!a_km && b_km) launch3<false, true, false>(g, grid, s, drop);
  else if (a_km && b_km) launch3<true, true, false>(g, grid, s, false);
  else launch3<true, false, false>(g, grid, s, false);
}
