// tvae_gemm: LDS-tiled MFMA GEMM for gfx950 with fused epilogues.
//
// Tile 128x128x32 per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 2x2 MFMA
// 32x32 tiles).  fp32 mode uses v_mfma_f32_32x32x2_f32 (an exact fp32 fma chain, 64 cycles per
// SIMD => 157 TFLOP/s chip peak), bf16 mode rounds the operands on their way into LDS and
// uses v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
// The contraction index inside a 32-wide k-tile is PERMUTED between lane halves: lane half
// hh owns k = 16*hh + s (s = 0..15), so each lane reads 16 contiguous floats of its row with
// four ds_read_b128 instead of 16 strided ds_read_b32.  A and B use the same permutation, so
// the sum is unchanged (only its order, which fp32 parity tolerates at the 1e-7 level).
//
// (bf16 mode: k-major operands are staged [k][row] too and their fragments come from transposed
// LDS reads, ds_read_b64_tr_b16.)
//
// Operands whose global layout is k-contiguous are staged as [row][k] (+4 floats of padding:
// stride 36 floats makes the b128 fragment reads bank-conflict-free); operands whose layout
// is m-contiguous ("k-major": weight for dgrad, both operands for wgrad) are staged as
// [k][row] (+4) and read with conflict-free ds_read_b32.
#include <stdlib.h>
#include "common.h"
#include "gemm_common.h"

namespace {

constexpr int LDS_KM = BM + 4;    // [k][row] stride (floats)
// bf16 [k][row] stride (elements) of a k-major operand: 320 bytes, so the four k-rows a transposed
// read gathers for one 32-lane half land on four different 64-byte bank spans (conflict-free)
constexpr int BF_KM = BM + 32;

// MFMA fragment (8 consecutive k of this lane's row) out of a bf16 [k][row] image: two
// ds_read_b64_tr_b16, each a 4 (k) x 16 (rows) block per 16-lane group delivered column-major --
// lane 4q+p of a group addresses k-row q, rows 4p..4p+3 and receives row (lane & 15), k = q' = 0..3.
// Needs EXEC all ones (the gather crosses lanes): only called from workgroup-uniform control flow.
typedef short short4v __attribute__((ext_vector_type(4)));
__device__ inline bf16x8 tr_fragment(const unsigned short* T, int k0, int row0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned short* a = T + (k0 + 8 * (g >> 1) + q) * BF_KM + row0 + 16 * (g & 1) + 4 * p;
  typedef __attribute__((address_space(3))) short4v* lds_ptr;
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)a);
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * BF_KM));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

// The k-depth of a tile is a template parameter: BK = 32 halves the barrier count, BK = 16 halves
// LDS and the fragment registers so that three workgroups share a CU instead of two (the prologue
// and epilogue of one tile then overlap the MFMA loops of two others -- with K = 256 = d_model
// on most launches a tile has only 8..16 k-steps, so that overlap is what sets the rate).
template <int BK>
struct Tile {
  static constexpr int KC = BK + 4;                 // [row][k] stride (floats): conflict-free b128 fragment reads
  static constexpr int BF = BK + 8;                 // bf16 [row][k] stride (elements)
  static constexpr int FLOATS = (BM * KC > BK * LDS_KM) ? BM * KC : BK * LDS_KM;
  static constexpr int NJ = BK / 8;                 // float4 per thread per operand tile
  static constexpr int HK = BK / 2;                 // k values owned by one lane half
  static_assert(BK * (BM + 32) * 2 <= FLOATS * 4, "the bf16 [k][row] image must fit the fp32-sized tile");
};

__device__ inline unsigned short f2bf(float f) {
  // round-to-nearest-even fp32 -> bf16 (the plain cast compiles to v_cvt_pk_bf16_f32 and keeps NaNs)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

// ---- global -> register staging of one 128x32 operand tile (4 float4 per thread) ----------
// stage_load only ISSUES the loads (out-of-range pieces read the always-valid first float4);
// stage_fix, called after the MFMAs of the current tile, zeroes those pieces and applies the
// dropout keep-scale.  Touching the loaded registers any earlier would put the s_waitcnt vmcnt
// in front of the MFMA block and expose the full memory latency once per k-tile.
template <int BK, bool KM>
__device__ inline void stage_pos(int f, int& row, int& k) {
  if (!KM) { row = f / (BK / 4); k = (f % (BK / 4)) * 4; } else { k = f >> 5; row = (f & 31) * 4; }
}

template <int BK, bool KM, bool ALIGNED>
__device__ inline void stage_load(float4 (&st)[BK / 8], const float* __restrict__ P, int ld,
                                  int row0, int nrows, int k0, int kend, int tid) {
#pragma unroll
  for (int j = 0; j < BK / 8; ++j) {
    int f = tid + 256 * j;
    int row, k;
    stage_pos<BK, KM>(f, row, k);
    int gr = row0 + row, gk = k0 + k;
    if (ALIGNED) {
      const bool ok = gr < nrows && gk < kend;
      const size_t off = KM ? ((size_t)gk * ld + gr) : ((size_t)gr * ld + gk);
      st[j] = *reinterpret_cast<const float4*>(P + (ok ? off : 0));
    } else {
      float t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int rr = KM ? gr + e : gr, kk = KM ? gk : gk + e;
        const bool ok = rr < nrows && kk < kend;
        t[e] = P[ok ? (KM ? (size_t)kk * ld + rr : (size_t)rr * ld + kk) : 0];
      }
      st[j] = make_float4(t[0], t[1], t[2], t[3]);
    }
  }
}

// ---- the same for k-steps that lie wholly inside the operand: no range tests, no per-load address arithmetic -------------
// Every vector instruction in the k-loop is time the matrix cores stand still (measured, tools/debug/mfma_loop.hip: a bare
// LDS-fed loop of this shape runs 148 TFLOP/s, 131 with ~60 VALU instructions added per k-step).  So the byte offset of each
// float4 of a thread is computed ONCE per tile (rows beyond the operand are clamped to its last row: they only ever reach
// C rows / columns that are not stored, and a_rowsum entries that are not written), the k position advances in a UNIFORM
// base pointer (scalar registers), and the loads are `base + fixed 32-bit offset`.
template <int BK, bool KM>
__device__ inline void stage_offsets(unsigned (&off)[BK / 8], int ld, int row0, int nrows, int tid) {
#pragma unroll
  for (int j = 0; j < BK / 8; ++j) {
    int row, k;
    stage_pos<BK, KM>(tid + 256 * j, row, k);
    // relative to the tile's first row (the caller's base pointer carries row0): operands of any size, offsets < 2^32 as long
    // as 128 rows (or 32 k-rows of a k-major operand) span less than 4 GiB
    const int gr = min(row0 + row, KM ? nrows - 4 : nrows - 1) - row0;      // KM: a float4 spans rows gr .. gr + 3 (nrows % 4 == 0)
    off[j] = 4u * (KM ? (unsigned)k * (unsigned)ld + (unsigned)gr : (unsigned)gr * (unsigned)ld + (unsigned)k);
  }
}
template <int BK>
__device__ inline void stage_load_fast(float4 (&st)[BK / 8], const float* __restrict__ base, const unsigned (&off)[BK / 8]) {
#pragma unroll
  for (int j = 0; j < BK / 8; ++j)
    st[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off[j]);
}

// A-side dropout mask on a fast-staged tile: the element index of each float4 of a thread is (per-thread constant) +
// (uniform per k-step), so the per-step work is one add and the keep-scale hash per float4
template <int BK, bool KM>
__device__ inline void drop_indices(unsigned (&idx)[BK / 8], int row0, int nrows, int dropM, int dropK, int tid) {
#pragma unroll
  for (int j = 0; j < BK / 8; ++j) {
    int row, k;
    stage_pos<BK, KM>(tid + 256 * j, row, k);
    const int gr = min(row0 + row, KM ? nrows - 4 : nrows - 1);      // the clamp of stage_offsets: such rows are never stored
    idx[j] = KM ? (unsigned)k * (unsigned)dropM + (unsigned)gr : (unsigned)gr * (unsigned)dropK + (unsigned)k;
  }
}
template <int BK>
__device__ inline void stage_drop_fast(float4 (&st)[BK / 8], const unsigned (&idx)[BK / 8], unsigned kadd, const DropCfg& dc) {
#pragma unroll
  for (int j = 0; j < BK / 8; ++j) {
    float ks[4];
    drop_scale4(dc, idx[j] + kadd, ks);
    st[j].x *= ks[0]; st[j].y *= ks[1]; st[j].z *= ks[2]; st[j].w *= ks[3];
  }
}

template <int BK, bool KM, bool DROP>
__device__ inline void stage_fix(float4 (&st)[BK / 8], int row0, int nrows, int k0, int kend, int tid,
                                 const DropCfg& dc, int dropM, int dropK) {
#pragma unroll
  for (int j = 0; j < BK / 8; ++j) {
    int f = tid + 256 * j;
    int row, k;
    stage_pos<BK, KM>(f, row, k);
    int gr = row0 + row, gk = k0 + k;
    float* pv = reinterpret_cast<float*>(&st[j]);
    float ks[4] = {1.f, 1.f, 1.f, 1.f};
    if (DROP) {
      // element (m,k) of A carries the keep-scale of the dropped-out tensor it is the gradient of;
      // the four elements of a float4 are consecutive indices of that tensor in either layout
      drop_scale4(dc, KM ? (uint32_t)gk * (uint32_t)dropM + (uint32_t)gr
                         : (uint32_t)gr * (uint32_t)dropK + (uint32_t)gk, ks);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int rr = KM ? gr + e : gr, kk = KM ? gk : gk + e;
      const bool ok = rr < nrows && kk < kend;
      pv[e] = ok ? pv[e] * ks[e] : 0.f;
    }
  }
}

template <int BK, bool KM>
__device__ inline void stage_store_f32(const float4 (&st)[BK / 8], float* __restrict__ T, int tid) {
#pragma unroll
  for (int j = 0; j < BK / 8; ++j) {
    int row, k;
    stage_pos<BK, KM>(tid + 256 * j, row, k);
    if (!KM) *reinterpret_cast<float4*>(&T[row * Tile<BK>::KC + k]) = st[j];
    else *reinterpret_cast<float4*>(&T[k * LDS_KM + row]) = st[j];
  }
}

template <int BK, bool KM>
__device__ inline void stage_store_bf16(const float4 (&st)[BK / 8], unsigned short* __restrict__ T, int tid) {
  constexpr int LDS_BF = Tile<BK>::BF;
#pragma unroll
  for (int j = 0; j < BK / 8; ++j) {
    int row, k;
    stage_pos<BK, KM>(tid + 256 * j, row, k);
    if (!KM) {
      ushort4 o = make_ushort4(f2bf(st[j].x), f2bf(st[j].y), f2bf(st[j].z), f2bf(st[j].w));
      *reinterpret_cast<ushort4*>(&T[row * LDS_BF + k]) = o;
    } else {                                // k-major operands keep their orientation: [k][row], one 8-byte store
      ushort4 o = make_ushort4(f2bf(st[j].x), f2bf(st[j].y), f2bf(st[j].z), f2bf(st[j].w));
      *reinterpret_cast<ushort4*>(&T[k * BF_KM + row]) = o;
    }
  }
}

template <int BK, bool A_KM, bool B_KM, bool ALIGNED, bool BF16, bool DROP, bool FAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BK == 16 ? 4 : 2)))
void gemm_kernel(GemmArgs g) {
  static_assert(!FAST || ALIGNED, "fast staging: aligned operands");
  if (gridDim.y > 1) {   // tvae_gemm_batched: one matrix product per blockIdx.y
    const long by = blockIdx.y;
    g.A += by * g.sA; g.B += by * g.sB; g.C += by * g.sC;
    if (g.res) g.res += by * g.sR;
  }
  constexpr int TILE_FLOATS = Tile<BK>::FLOATS, LDS_KC = Tile<BK>::KC, LDS_BF = Tile<BK>::BF;
  constexpr int NJ = Tile<BK>::NJ, HK = Tile<BK>::HK;
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TILE_FLOATS];
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
  const int nkt = (kend - kbeg + BK - 1) / BK;
  // FAST: the k-steps that lie wholly inside [kbeg, kend) are staged without range tests or per-load address arithmetic;
  // a last partial step (the token count of a weight gradient is not a multiple of BK) goes through the checked path once,
  // after the loop
  const int nloop = FAST ? (kend - kbeg) / BK : nkt;

  DropCfg dc = drop_cfg(DROP ? g.rng : nullptr, g.site, g.p);

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();

  float4 sa[NJ], sb[NJ];
  DropCfg nodrop = drop_cfg(nullptr, 0, 0.f);
  // fused bias gradient: in k-major staging all four float4 of a thread cover the same 4 rows
  const bool do_rowsum = A_KM && (g.a_rowsum != nullptr) && (tile_n == 0);
  float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
  auto add_rowsum = [&]() {
#pragma unroll
    for (int j = 0; j < NJ; ++j) { rs.x += sa[j].x; rs.y += sa[j].y; rs.z += sa[j].z; rs.w += sa[j].w; }
  };

  auto tileA = [&](int buf) { return lds + (buf * 2 + 0) * TILE_FLOATS; };
  auto tileB = [&](int buf) { return lds + (buf * 2 + 1) * TILE_FLOATS; };

  unsigned offA[FAST ? NJ : 1], offB[FAST ? NJ : 1], didx[(FAST && DROP) ? NJ : 1];
  if constexpr (FAST) {
    stage_offsets<BK, A_KM>(offA, g.lda, m0, g.M, tid);
    stage_offsets<BK, B_KM>(offB, g.ldb, n0, g.N, tid);
    if constexpr (DROP) drop_indices<BK, A_KM>(didx, m0, g.M, g.M, g.K, tid);
  }
  // loads of the k-step that starts at k0 (only issued; see stage_load) / what must happen to the registers before the LDS store
  auto load_step = [&](int k0, bool checked) {
    if constexpr (FAST) {
      if (!checked) {
        stage_load_fast<BK>(sa, g.A + (A_KM ? (size_t)k0 * g.lda + m0 : (size_t)m0 * g.lda + k0), offA);
        stage_load_fast<BK>(sb, g.B + (B_KM ? (size_t)k0 * g.ldb + n0 : (size_t)n0 * g.ldb + k0), offB);
        return;
      }
    }
    stage_load<BK, A_KM, ALIGNED>(sa, g.A, g.lda, m0, g.M, k0, kend, tid);
    stage_load<BK, B_KM, ALIGNED>(sb, g.B, g.ldb, n0, g.N, k0, kend, tid);
  };
  auto fix_step = [&](int k0, bool checked) {
    if (FAST && !checked) {
      if constexpr (FAST && DROP) stage_drop_fast<BK>(sa, didx, A_KM ? (unsigned)k0 * (unsigned)g.M : (unsigned)k0, dc);
      return;
    }
    stage_fix<BK, A_KM, DROP>(sa, m0, g.M, k0, kend, tid, dc, g.M, g.K);
    stage_fix<BK, B_KM, false>(sb, n0, g.N, k0, kend, tid, nodrop, 0, 0);
  };
  auto store_step = [&](int buf) {
    if (do_rowsum) add_rowsum();
    if (BF16) {
      stage_store_bf16<BK, A_KM>(sa, reinterpret_cast<unsigned short*>(tileA(buf)), tid);
      stage_store_bf16<BK, B_KM>(sb, reinterpret_cast<unsigned short*>(tileB(buf)), tid);
    } else {
      stage_store_f32<BK, A_KM>(sa, tileA(buf), tid);
      stage_store_f32<BK, B_KM>(sb, tileB(buf), tid);
    }
  };
  auto compute = [&](int buf) {
    if (!BF16) {
      const float* TA = tileA(buf);
      const float* TB = tileB(buf);
      // fragments are fetched in sub-steps of 4 k values (one ds_read_b128 per 32-row block and
      // operand) so that only 16 fragment registers are live beside the 64 accumulators
#pragma unroll
      for (int q = 0; q < HK / 4; ++q) {
        // k values of this lane half in sub-step q: a 32-deep tile is walked as two 16-deep ones (k = 16 (q >> 1) +
        // 8 hh + 4 (q & 1) ..), so the BK = 32 and BK = 16 instantiations add the products of an output element in the
        // SAME order: which of them the host picks (it depends on M) cannot change a single bit of the result --
        // greedy decode of a slab, of its pieces and of its rows in another order give identical tokens
        const int kq = 16 * (q >> 1) + 8 * hh + 4 * (q & 1);
        float a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = wm * 64 + 32 * i + r;
          if (!A_KM) {
            float4 v = *reinterpret_cast<const float4*>(&TA[row * LDS_KC + kq]);
            a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) a[i][s] = TA[(kq + s) * LDS_KM + row];
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = wn * 64 + 32 * j + r;
          if (!B_KM) {
            float4 v = *reinterpret_cast<const float4*>(&TB[col * LDS_KC + kq]);
            b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) b[j][s] = TB[(kq + s) * LDS_KM + col];
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
      }
    } else {
      const unsigned short* TA = reinterpret_cast<const unsigned short*>(tileA(buf));
      const unsigned short* TB = reinterpret_cast<const unsigned short*>(tileB(buf));
      constexpr int KS = BK / 16;
      bf16x8 a[2][KS], b[2][KS];   // [tile][k-step of 16]
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (A_KM) a[i][ks] = tr_fragment(TA, 16 * ks, wm * 64 + 32 * i, lane);
          else a[i][ks] = *reinterpret_cast<const bf16x8*>(&TA[(wm * 64 + 32 * i + r) * LDS_BF + 16 * ks + 8 * hh]);
        }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (B_KM) b[j][ks] = tr_fragment(TB, 16 * ks, wn * 64 + 32 * j, lane);
          else b[j][ks] = *reinterpret_cast<const bf16x8*>(&TB[(wn * 64 + 32 * j + r) * LDS_BF + 16 * ks + 8 * hh]);
        }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], b[j][ks], acc[i][j], 0, 0, 0);
    }
  };

  if (nkt > 0) {
    const bool checked = nloop == 0;           // the only step is a partial one
    load_step(kbeg, checked);
    fix_step(kbeg, checked);
    store_step(0);
  }
  __syncthreads();

  for (int kt = 0; kt < nloop; ++kt) {
    const int buf = kt & 1;
    const bool more = (kt + 1 < nloop);
    if (more) load_step(kbeg + (kt + 1) * BK, false);
    compute(buf);
    if (more) {
      fix_step(kbeg + (kt + 1) * BK, false);   // after the MFMAs: the loads have landed by now
      store_step(buf ^ 1);
    }
    __syncthreads();
  }
  if (FAST && nkt > nloop) {                   // the partial last k-step, range-checked, not overlapped with anything
    const int buf = nloop & 1;
    if (nloop > 0) {
      load_step(kbeg + nloop * BK, true);
      fix_step(kbeg + nloop * BK, true);
      store_step(buf);
      __syncthreads();
    }
    compute(buf);
    __syncthreads();
  }

  if (do_rowsum) {   // every thread is past the last barrier of the k-loop: LDS is free
    float* red = lds;
    *reinterpret_cast<float4*>(&red[(tid >> 5) * BM + (tid & 31) * 4]) = rs;
    __syncthreads();
    if (tid < BM && m0 + tid < g.M) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[w * BM + tid];
      atomicAdd(&g.a_rowsum[m0 + tid], t);
    }
  }

  gemm_epilogue<(BK == 16) ? 2 : 4>(g, acc, lds, m0, n0, do_rowsum);
}

template <int BK, bool A_KM, bool B_KM, bool ALIGNED, bool BF16>
void launch2(const GemmArgs& g, dim3 grid, hipStream_t s, bool drop) {
  if (drop) {
    if constexpr (ALIGNED) {
      if (g.flags & FLAG_FAST_STAGE) hipLaunchKernelGGL((gemm_kernel<BK, A_KM, B_KM, true, BF16, true, true>), grid, dim3(256), 0, s, g);
      else hipLaunchKernelGGL((gemm_kernel<BK, A_KM, B_KM, true, BF16, true, false>), grid, dim3(256), 0, s, g);
    } else {
      hipLaunchKernelGGL((gemm_kernel<BK, A_KM, B_KM, ALIGNED, BF16, true, false>), grid, dim3(256), 0, s, g);
    }
  } else if constexpr (ALIGNED) {
    if (g.flags & FLAG_FAST_STAGE) hipLaunchKernelGGL((gemm_kernel<BK, A_KM, B_KM, true, BF16, false, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel<BK, A_KM, B_KM, true, BF16, false, false>), grid, dim3(256), 0, s, g);
  } else {
    hipLaunchKernelGGL((gemm_kernel<BK, A_KM, B_KM, ALIGNED, BF16, false, false>), grid, dim3(256), 0, s, g);
  }
}
template <int BK, bool A_KM, bool B_KM>
void launch1(const GemmArgs& g, dim3 grid, hipStream_t s, bool aligned, bool bf16, bool drop) {
  if (aligned) {
    if (bf16) launch2<BK, A_KM, B_KM, true, true>(g, grid, s, drop);
    else launch2<BK, A_KM, B_KM, true, false>(g, grid, s, drop);
  } else {
    if (bf16) launch2<BK, A_KM, B_KM, false, true>(g, grid, s, drop);
    else launch2<BK, A_KM, B_KM, false, false>(g, grid, s, drop);
  }
}
template <int BK>
void launch0(const GemmArgs& g, dim3 grid, hipStream_t s, bool a_km, bool b_km, bool aligned, bool bf16, bool drop) {
  if (!a_km && !b_km) launch1<BK, false, false>(g, grid, s, aligned, bf16, drop);
  else if (!a_km && b_km) launch1<BK, false, true>(g, grid, s, aligned, bf16, drop);
  else if (a_km && b_km) launch1<BK, true, true>(g, grid, s, aligned, bf16, drop);
  else launch1<BK, true, false>(g, grid, s, aligned, bf16, drop);
}

// Workgroups resident on the chip at once: 256 CUs x (LDS- and register-limited) 2 for BK = 32, 4 for BK = 16
constexpr int SLOTS32 = 512, SLOTS16 = 1024;

int forced_depth() {   // TVAE_GEMM_BK (16 | 32) overrides the choice below; read once
  static int forced = [] { const char* e = getenv("TVAE_GEMM_BK"); return e ? atoi(e) : 0; }();
  return (forced == 16 || forced == 32) ? forced : 0;
}

// BK = 16 runs four workgroups per CU and is a few percent faster when the grid fills whole
// rounds of them; a grid that leaves the last round mostly empty is better off with BK = 32.
int tile_depth(bool bf16, int tiles, bool splitk, bool drop) {
  if (int f = forced_depth()) return f;
  if (bf16) return 32;
  // split-K: half as many slabs (atomic epilogues) for the same K -- 276 vs 283 us on dW[1024,256], 90 vs 105 on dW[256,256];
  // the form that regenerates a dropout mask while staging is the other way round (304 vs 308, 167 vs 172)
  if (splitk) return drop ? 16 : 32;
  auto eff = [](int t, int slots) { return (double)t / (double)(cdiv(t, slots) * slots); };
  return eff(tiles, SLOTS16) * 1.03 >= eff(tiles, SLOTS32) ? 16 : 32;
}

}  // namespace

void tvae_gemm3_launch(const void* gp, int tiles, int splits, bool a_km, bool b_km, bool drop, bool b_split,
                       hipStream_t s);   // gemm3.hip
void tvae_gemm_dma_launch(const void* gp, bool bf16, hipStream_t s);   // gemm_dma.hip
void tvae_gemm_wide_launch(const void* gp, int splits, bool km, hipStream_t s);   // gemm_wide.hip
void tvae_gemm3_dma_launch(const void* gp, hipStream_t s);   // gemm3_dma.hip

namespace {
// TVAE_GEMM_WIDE: 0 never use the 128 x 160 kernel, 1 (default) for N in (128, 160], 2 for every N that wastes fewer columns on
// 160-wide tiles than on 128-wide ones (experiment switch; read once)
int wide_mode() {
  static int m = [] { const char* e = getenv("TVAE_GEMM_WIDE"); return e ? atoi(e) : 1; }();
  return m;
}
bool auto_split_enabled() {   // TVAE_GEMM_AUTOSPLIT=0 switches the tail-balance split of pure products off (read once)
  static bool on = [] { const char* e = getenv("TVAE_GEMM_AUTOSPLIT"); return !e || atoi(e) != 0; }();
  return on;
}
bool x3_dma_enabled() {   // TVAE_GEMM3_DMA=0: f32x3 products against pre-split weights stay on the register-staged kernel (read once)
  static bool on = [] { const char* e = getenv("TVAE_GEMM3_DMA"); return !e || atoi(e) != 0; }();
  return on;
}
bool fp32_dma_enabled() {   // TVAE_GEMM_DMA=1: exact-fp32 NT products take the LDS-DMA kernel too (experiment switch; read once)
  static bool on = [] { const char* e = getenv("TVAE_GEMM_DMA"); return e && atoi(e) == 1; }();
  return on;
}
}  // namespace

static int gemm_impl(const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                     int M, int N, int K, const float* bias, const float* residual, int ldr,
                     float alpha, int flags, const uint32_t* rng, uint32_t site, float p_drop,
                     float* a_rowsum, int dtype, void* stream, int batch, long sA, long sB, long sC, long sR) {
  TVAE_CHECK(A && B && C, TVAE_E_NULL, "tvae_gemm: null operand");
  TVAE_CHECK(M > 0 && N > 0 && K > 0, TVAE_E_SHAPE, "tvae_gemm: M,N,K must be positive (%d,%d,%d)", M, N, K);
  TVAE_CHECK(dtype == TVAE_F32 || dtype == TVAE_BF16 || dtype == TVAE_F32X3, TVAE_E_DTYPE, "tvae_gemm: dtype %d", dtype);
  const bool a_km = flags & TVAE_GEMM_A_KMAJOR, b_km = flags & TVAE_GEMM_B_KMAJOR;
  // lda / ldb may be SMALLER than the contiguous extent: overlapping rows are how the conv
  // bottlenecks present their im2col operands (lda = C_in < K = k*C_in).  Only C must not overlap.
  TVAE_CHECK(lda >= 1 && ldb >= 1 && ldc >= N, TVAE_E_SHAPE,
             "tvae_gemm: leading dimension too small (lda %d ldb %d ldc %d, N %d)", lda, ldb, ldc, N);
  TVAE_CHECK(!((flags & 64) && !residual), TVAE_E_NULL, "tvae_gemm: RELUMASK needs the saved activation");
  TVAE_CHECK(!residual || ldr >= N, TVAE_E_SHAPE, "tvae_gemm: ldr %d < N %d", ldr, N);
  TVAE_CHECK(!(rng && p_drop > 0.f) || ((long)M * N < (1L << 32) && (long)M * K < (1L << 32)), TVAE_E_SHAPE,
             "tvae_gemm: dropout index space exceeds 32 bits");
  if (flags & TVAE_GEMM_SPLITK)
    TVAE_CHECK(!(flags & (TVAE_GEMM_RELU | 64)) && !residual && !(rng && p_drop > 0.f && !(flags & TVAE_GEMM_A_DROPMASK)), TVAE_E_SHAPE,
               "tvae_gemm: split-K supports bias/alpha only (no ReLU, residual or output dropout: slabs are summed with atomics)");

  // float4 staging needs 16-byte aligned rows and a whole number of float4 along the
  // contiguous dimension (k for row-major operands, m/n for k-major ones)
  bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && aligned16(A) && aligned16(B) &&
                 (a_km ? (M % 4 == 0) : (K % 4 == 0)) && (b_km ? (N % 4 == 0) : (K % 4 == 0));
  if (batch > 1 && (sA % 4 || sB % 4)) aligned = false;
  // Tail balance for pure products (nothing fused, C contiguous): a launch of few tiles with a long K -- the data gradients of
  // the conv bottleneck: 51-536 tiles of K = 512..2816 -- leaves CUs idle or gives some CUs 3 tiles and others 2, and the
  // launch takes as long as the busiest CU.  Such a product is cut into 2-8 K slabs that ADD into a zeroed C (the split-K
  // path of the weight gradients; the memset is asynchronous on the same stream), so the work items are several per CU.
  const bool bf16_ = dtype == TVAE_BF16;
  if (!(flags & ~(TVAE_GEMM_A_KMAJOR | TVAE_GEMM_B_KMAJOR)) && !bias && !residual && !a_rowsum && batch == 1 && !bf16_ &&
      !(rng && p_drop > 0.f) && ldc == N && K >= 512 && auto_split_enabled()) {
    const bool wide = wide_mode() >= 1 && N > 128 && N <= 160 && (flags & TVAE_GEMM_A_KMAJOR ? 1 : 0) == (flags & TVAE_GEMM_B_KMAJOR ? 1 : 0);
    const int t = wide ? cdiv(M, 128) * cdiv(N, 160) : cdiv(M, BM) * cdiv(N, BN);
    // Measured (profiles/r3_gemm_shapes_step_fp32.txt): worth it only where less than half of the CUs would have a tile at all
    // (51 tiles: 103 -> 53 us) -- then the slab count with the best balance -- and on the wide tile when the busiest CU
    // has 3 tiles against an average of 2.1 (536 tiles: 744 -> 701 us); 172 tiles cut in 4 ran SLOWER (136 -> 175 us: the
    // atomic epilogue of a 14-step slab costs more than the idle CUs).  Every slab keeps at least 256 of K.
    auto eff = [&](int sp) { const double x = (double)t * sp / 256.0; return x / (double)cdiv((long)t * sp, 256); };
    int best = 1;
    if (t < 128) {
      for (int sp = 2; sp <= 8 && K / sp >= 256 && (long)t * sp <= 512; ++sp)
        if (eff(sp) > eff(best) + 0.02) best = sp;
    } else if (wide && eff(1) < 0.75) {
      for (int sp = 2; sp <= 4 && K / sp >= 256; ++sp)
        if (eff(sp) >= 0.88) { best = sp; break; }
    }
    if (best > 1) {
      if (hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), (hipStream_t)stream) != hipSuccess) {
        tvae_set_error("tvae_gemm: hipMemsetAsync of the split product's output failed");
        return TVAE_E_LAUNCH;
      }
      return gemm_impl(A, lda, B, ldb, C, ldc, M, N, K, nullptr, nullptr, 0, alpha, flags | TVAE_GEMM_SPLITK | (best << 24), rng, site,
                       p_drop, nullptr, dtype, stream, 1, 0, 0, 0, 0);
    }
  }
  const int forced_splits = (flags >> 24) & 15;      // internal: set by the tail-balance path above
  flags &= ~(15 << 24);
  GemmArgs g;
  g.sA = sA; g.sB = sB; g.sC = sC; g.sR = sR;
  g.split = 0;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.res = residual;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.M = M; g.N = N; g.K = K;
  g.alpha = alpha; g.flags = flags; g.rng = rng; g.site = site; g.p = p_drop;
  g.a_rowsum = a_rowsum;
  if (!(flags & TVAE_GEMM_SPLITK) && N % 4 == 0 && ldc % 4 == 0 && aligned16(C) && (batch == 1 || (sC % 4 == 0 && sR % 4 == 0)) &&
      (!bias || aligned16(bias)) && (!residual || (ldr % 4 == 0 && aligned16(residual))))
    g.flags |= FLAG_VEC_EPILOGUE;
  TVAE_CHECK(!a_rowsum || a_km, TVAE_E_SHAPE, "tvae_gemm: a_rowsum needs a k-major A (wgrad form)");
  {
    // fast staging needs the byte offsets inside ONE TILE's rows below 4 GiB (32-bit per-thread offsets relative to the
    // tile's first row: the operand itself may be of any size -- the 8.3 M-row memory expansion of a 65536-latent decode
    // slab is 8.5 GB) and, for the clamped rows, at least one whole row group to clamp to
    const double extA = 4.0 * (a_km ? (double)32 * lda + 128 : (double)128 * lda + 32), extB = 4.0 * (b_km ? (double)32 * ldb + 128 : (double)128 * ldb + 32);
    if (aligned && extA < 4.0e9 && extB < 4.0e9 && M >= 4 && N >= 4) g.flags |= FLAG_FAST_STAGE;
  }
  const bool bf16 = dtype == TVAE_BF16;
  const bool drop = (flags & TVAE_GEMM_A_DROPMASK) && rng && p_drop > 0.f;
  // split-bf16 kernel: aligned operands only; a k-major A with the A-side dropout mask stays on the fp32 kernel
  const bool b_split = flags & TVAE_GEMM_B_SPLIT3;
  if (b_split) {
    // B holds the bf16 planes written by tvae_split3 (tile-major, gemm3_split.h), ldb = 3 * Kb elements
    TVAE_CHECK(dtype == TVAE_F32X3 && !b_km && !(flags & TVAE_GEMM_SPLITK) && !(a_km && drop), TVAE_E_SHAPE,
               "tvae_gemm: B_SPLIT3 needs dtype F32X3, an n-major B and no split-K");
    TVAE_CHECK(ldb % 48 == 0 && ldb / 3 >= K && K % 16 == 0 && aligned16(B) && aligned16(A) && lda % 4 == 0 &&
                   (a_km ? M % 4 == 0 : true),
               TVAE_E_SHAPE, "tvae_gemm: B_SPLIT3 needs K %% 16 == 0, ldb = 3*Kb with Kb %% 16 == 0, aligned operands");
  }
  const bool x3 = batch == 1 && dtype == TVAE_F32X3 && (b_split || (aligned && !(a_km && drop)));
  int tiles = cdiv(M, BM) * cdiv(N, BN);
  // streamed NT kernel (gemm_dma.hip): bf16 arithmetic against a weight that is bf16 in memory already, or (switch) exact fp32
  const bool b_bf16 = flags & TVAE_GEMM_B_BF16;
  if (b_bf16) {
    TVAE_CHECK(dtype == TVAE_BF16 && !a_km && !b_km && !(flags & TVAE_GEMM_SPLITK) && !drop, TVAE_E_SHAPE,
               "tvae_gemm: B_BF16 needs dtype BF16, row-major A, an n-major B, no split-K and no A-side dropout mask");
    TVAE_CHECK(K % 32 == 0 && lda % 4 == 0 && aligned16(A) && ldb % 8 == 0 && ldb >= K && aligned16(B), TVAE_E_SHAPE,
               "tvae_gemm: B_BF16 needs K %% 32 == 0 and 16-byte aligned rows (K %d lda %d ldb %d)", K, lda, ldb);
  }
  const bool dma32 = batch == 1 && dtype == TVAE_F32 && fp32_dma_enabled() && !a_km && !b_km && !(flags & TVAE_GEMM_SPLITK) && !drop &&
                     K % 16 == 0 && aligned;
  if (b_bf16 || dma32) {
    g.k_per_split = K;
    tvae_gemm_dma_launch(&g, b_bf16, (hipStream_t)stream);
    TVAE_LAUNCH_CHECK("tvae_gemm");
    return TVAE_OK;
  }
  // wide-tile kernel (gemm_wide.hip): exact fp32, N just above a multiple of 128 (the conv bottleneck's 148 / 160 channels), both
  // operands in the same orientation, nothing fused beyond alpha / bias / ReLU / split-K.  In f32x3 mode too for N <= 160: the
  // 3xbf16 kernel on a 58-62 % full tile pair is no faster than exact fp32 on a full tile.
  {
    const int wm = wide_mode();
    const bool wide_n = wm >= 1 && N > 128 && (N <= 160 || (wm >= 2 && dtype == TVAE_F32 && cdiv(N, 160) * 160 < cdiv(N, 128) * 128));
    const int other = flags & ~(TVAE_GEMM_RELU | TVAE_GEMM_SPLITK | TVAE_GEMM_A_KMAJOR | TVAE_GEMM_B_KMAJOR | FLAG_VEC_EPILOGUE | FLAG_FAST_STAGE);
    const bool any_drop = rng && p_drop > 0.f;          // epilogue dropout or the A-side mask: the wide kernel has neither
    if (wide_n && batch == 1 && !bf16 && a_km == b_km && other == 0 && !residual && !a_rowsum && !any_drop && K >= 256 &&
        (g.flags & FLAG_FAST_STAGE)) {
      int splits = 1;
      g.k_per_split = K;
      if (flags & TVAE_GEMM_SPLITK) {
        const int wtiles = cdiv(M, 128) * cdiv(N, 160), ktiles = cdiv(K, 32);
        splits = forced_splits ? forced_splits : 768 / wtiles;     // one round of resident workgroups: 256 CUs x 3
        if (splits > ktiles / 8) splits = ktiles / 8;
        if (splits < 1) splits = 1;
        g.k_per_split = cdiv(ktiles, splits) * 32;
        splits = cdiv(K, g.k_per_split);
      }
      tvae_gemm_wide_launch(&g, splits, a_km, (hipStream_t)stream);
      TVAE_LAUNCH_CHECK("tvae_gemm");
      return TVAE_OK;
    }
  }
  const int BK = x3 ? 32 : tile_depth(bf16, tiles, flags & TVAE_GEMM_SPLITK, drop);
  int splits = 1;
  if (flags & TVAE_GEMM_SPLITK) {
    // one full round of resident workgroups, each with at least 256 of K: more splits only add
    // atomic epilogues, fewer leave CUs idle
    int ktiles = cdiv(K, 32);
    splits = forced_splits ? forced_splits : (BK == 16 ? SLOTS16 : SLOTS32) / tiles;
    if (splits > ktiles / 8) splits = ktiles / 8;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    g.k_per_split = cdiv(ktiles, splits) * 32;
    splits = cdiv(K, g.k_per_split);
  } else {
    g.k_per_split = cdiv(K, 32) * 32;
  }
  hipStream_t s = (hipStream_t)stream;
  if (x3 && b_split && !a_km && !drop && x3_dma_enabled()) {      // streamed kernel: no register staging, no LDS stores
    g.k_per_split = K;
    tvae_gemm3_dma_launch(&g, s);
    TVAE_LAUNCH_CHECK("tvae_gemm");
    return TVAE_OK;
  }
  if (x3) {
    tvae_gemm3_launch(&g, tiles, splits, a_km, b_km, drop, b_split, s);
    TVAE_LAUNCH_CHECK("tvae_gemm");
    return TVAE_OK;
  }
  dim3 grid(tiles, batch, splits);
  if (BK == 16) launch0<16>(g, grid, s, a_km, b_km, aligned, bf16, drop);
  else launch0<32>(g, grid, s, a_km, b_km, aligned, bf16, drop);
  TVAE_LAUNCH_CHECK("tvae_gemm");
  return TVAE_OK;
}

extern "C" int tvae_gemm(const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                         int M, int N, int K, const float* bias, const float* residual, int ldr,
                         float alpha, int flags, const uint32_t* rng, uint32_t site, float p_drop,
                         float* a_rowsum, int dtype, void* stream) {
  return gemm_impl(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, alpha, flags, rng, site, p_drop, a_rowsum, dtype,
                   stream, 1, 0, 0, 0, 0);
}

// `batch` independent products of one shape in one launch (blockIdx.y): C_i = epilogue(op(A_i) op(B_i)), matrices of
// consecutive entries strideA / strideB / strideC (/ strideR for the residual or saved activation) elements apart.
// The exact-fp32 kernel only: no split-K, no dropout, no pre-split or bf16 operands.
extern "C" int tvae_gemm_batched(const float* A, int lda, long strideA, const float* B, int ldb, long strideB,
                                 float* C, int ldc, long strideC, int M, int N, int K, int batch,
                                 const float* bias, const float* residual, int ldr, long strideR,
                                 float alpha, int flags, void* stream) {
  TVAE_CHECK(batch >= 1 && batch <= 65535, TVAE_E_SHAPE, "tvae_gemm_batched: batch %d outside 1..65535", batch);
  TVAE_CHECK(!(flags & (TVAE_GEMM_SPLITK | TVAE_GEMM_A_DROPMASK | TVAE_GEMM_B_SPLIT3 | TVAE_GEMM_B_BF16)), TVAE_E_SHAPE,
             "tvae_gemm_batched: split-K, dropout and pre-converted operands are not supported");
  TVAE_CHECK(strideA >= 0 && strideB >= 0 && strideC >= (long)(M - 1) * ldc + N, TVAE_E_SHAPE,
             "tvae_gemm_batched: output matrices overlap (strideC %ld)", strideC);
  return gemm_impl(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, alpha, flags, nullptr, 0, 0.f, nullptr, TVAE_F32,
                   stream, batch, strideA, strideB, strideC, strideR);
}
