// Pieces shared by the GEMM kernels of libtvae_hip.so: argument block, tile constants, fused epilogue.
#pragma once
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int FLAG_VEC_EPILOGUE = 1 << 20;   // internal: float4 epilogue through LDS (set by the host wrapper)
constexpr int FLAG_FAST_STAGE = 1 << 21;     // internal: per-tile 32-bit staging offsets + uniform k base (set by the host wrapper)
constexpr int EPILOGUE_LDS_FLOATS = 4 * 32 * 68;   // what gemm_epilogue needs of the kernel's LDS

struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias; const float* res;
  int lda, ldb, ldc, ldr, M, N, K;
  float alpha; int flags;
  const uint32_t* rng; uint32_t site; float p;
  int k_per_split;
  float* a_rowsum;   // A_KMAJOR only: a_rowsum[m] += sum_k A(m,k) (bias gradient fused into wgrad)
  long sA, sB, sC, sR;   // tvae_gemm_batched: element strides between the matrices of consecutive batch entries (blockIdx.y)
  int n_splits;          // split-K: slabs of K (the grid's z extent is this rounded up to a multiple of 8, see tile_and_split)
  int split;             // written by the kernel: the slab this workgroup sums
};


// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (id % 8 shares an L2),
// so give each XCD a CONTIGUOUS run of tiles -- the column tiles that re-read one A row panel then
// hit the same L2 instead of fetching the panel once per XCD.  (Bijective for any grid size;
// placement is a speed matter only.)
__device__ inline int xcd_tile_id() {
  const int nwg = gridDim.x, id = blockIdx.x, q = nwg >> 3, rr = nwg & 7, xcd = id & 7;
  return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
}

// Split-K placement (grid = tiles x 1 x splits rounded up to 8).  With blockIdx.z as the slab, an XCD saw ONE row panel of
// tiles for every slab, so each slab of the operand that all row panels share was fetched by all 8 L2s (PMC: 830 MB per
// launch of the 63500-token dW[1024,256] against 325 MB of operands).  Dealt this way, the tiles of one slab are consecutive
// workgroups of ONE XCD and both operand slabs enter one L2 once.  Returns false for the padding workgroups.
__device__ inline bool tile_and_split(GemmArgs& g, int& wgid) {
  if (gridDim.z == 1) { wgid = xcd_tile_id(); g.split = 0; return true; }
  const int f = blockIdx.z * gridDim.x + blockIdx.x, j = f >> 3;
  wgid = j % (int)gridDim.x;
  g.split = (j / (int)gridDim.x) * 8 + (f & 7);
  return g.split < g.n_splits;
}

// Epilogue of a 128x128 tile held as 2x2 MFMA 32x32 accumulators per wave (4 waves as 2x2):
// alpha, bias, ReLU / ReLU-mask, dropout, residual, accumulate -- or split-K atomics.  EB = row
// segments in flight per lane in the vector path (register budget).  `lds` is the kernel's staging
// memory (>= EPILOGUE_LDS_FLOATS floats); every wave must be past its last read of it.
// WN = waves along n (the workgroup has 2 * WN waves, tile 128 x 64 WN; needs 2 * WN * 32 * 68 floats of LDS).
template <int EB, int WN = 2>
__device__ inline void gemm_epilogue(const GemmArgs& g, floatx16 (&acc)[2][2], float* lds, int m0, int n0,
                                     bool sync_first) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, hh = lane >> 5;
  const bool do_rowsum = sync_first;
  // ---- epilogue ----------------------------------------------------------------------------
  const bool relu = g.flags & TVAE_GEMM_RELU, accum = g.flags & TVAE_GEMM_ACCUM;
  const bool splitk = g.flags & TVAE_GEMM_SPLITK, relumask = g.flags & 64;
  DropCfg de = drop_cfg((g.flags & TVAE_GEMM_A_DROPMASK) ? nullptr : g.rng, g.site, g.p);
  const bool lead = (g.split == 0);
  if (g.flags & FLAG_VEC_EPILOGUE) {
    // Row-per-lane accumulator stores are issue-bound (64 dword stores per lane).  Turn the wave's
    // 64x64 tile through its own LDS slab instead and leave as 16 float4 per lane: 16 lanes cover
    // one 256-byte row segment, residual / saved-activation reads are float4 too.
    if (do_rowsum) __syncthreads();
    constexpr int ES = 68;
    float* T = lds + wave * (32 * ES);
    const int c = (lane & 15) * 4;
    const int gcol = n0 + wn * 64 + c;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias && gcol < g.N) bv = *reinterpret_cast<const float4*>(g.bias + gcol);
    const float* __restrict__ aux = g.res;      // residual or saved activation; must not alias C
    const bool need_aux = (aux != nullptr);
    const bool col_ok = gcol < g.N;
#pragma unroll
    for (int i = 0; i < 2; ++i) {               // 32 rows per pass (LDS ops of one wave stay in order)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) T[crow(q, hh) * ES + 32 * j + r] = acc[i][j][q];
      for (int it0 = 0; it0 < 8; it0 += EB) {
        // batch of EB row segments: issue every global read first, then compute, then store
        float4 m[EB], old[EB], v[EB];
        bool ok[EB];
        size_t coff[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) {
          const int rl = (it0 + u) * 4 + (lane >> 4);
          const int grow = m0 + wm * 64 + 32 * i + rl;
          ok[u] = col_ok && grow < g.M;
          coff[u] = (size_t)(ok[u] ? grow : 0) * g.ldc + (col_ok ? gcol : 0);
          v[u] = *reinterpret_cast<const float4*>(&T[rl * ES + c]);
          m[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          old[u] = m[u];
          if (need_aux) m[u] = *reinterpret_cast<const float4*>(aux + (size_t)(ok[u] ? grow : 0) * g.ldr + (col_ok ? gcol : 0));
          if (accum) old[u] = *reinterpret_cast<const float4*>(g.C + coff[u]);
        }
#pragma unroll
        for (int u = 0; u < EB; ++u) {
          const int grow = m0 + wm * 64 + 32 * i + (it0 + u) * 4 + (lane >> 4);
          float o[4] = {v[u].x * g.alpha + bv.x, v[u].y * g.alpha + bv.y, v[u].z * g.alpha + bv.z, v[u].w * g.alpha + bv.w};
          const float mm[4] = {m[u].x, m[u].y, m[u].z, m[u].w};
          if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
          }
          if (relumask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = mm[e] > 0.f ? o[e] : 0.f;
          } else {
            if (de.on) {
              float ks[4];
              drop_scale4(de, (uint32_t)grow * (uint32_t)g.N + (uint32_t)gcol, ks);
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] *= ks[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += mm[e];      // zero when there is no residual
          }
          if (ok[u])
            *reinterpret_cast<float4*>(g.C + coff[u]) =
                make_float4(o[0] + old[u].x, o[1] + old[u].y, o[2] + old[u].z, o[3] + old[u].w);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int col = n0 + wn * 64 + 32 * j + r;
      if (col >= g.N) continue;
      float bv = (g.bias && lead) ? g.bias[col] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        int row = m0 + wm * 64 + 32 * i + crow(q, hh);
        if (row >= g.M) continue;
        float v = acc[i][j][q] * g.alpha + bv;
        if (splitk) {
          atomicAdd(&g.C[(size_t)row * g.ldc + col], v);
          continue;
        }
        if (relu) v = fmaxf(v, 0.f);
        if (relumask) {
          // backward of relu(+dropout): pass the gradient where the saved activation is > 0
          v = (g.res[(size_t)row * g.ldr + col] > 0.f) ? v : 0.f;
        } else {
          v *= drop_scale(de, (uint32_t)row * (uint32_t)g.N + (uint32_t)col);
          if (g.res) v += g.res[(size_t)row * g.ldr + col];
        }
        float* dst = &g.C[(size_t)row * g.ldc + col];
        *dst = accum ? (*dst + v) : v;
      }
    }
}

}  // namespace
