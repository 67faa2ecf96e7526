// Pieces shared by the GEMM kernels of libtvae_hip.so: argument block, tile constants, fused epilogue.
#pragma once
#include "common.h"
#include <type_traits>

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
  int split;             // written by the kernel: the slab this workgroup sums
};


// XCD-aware work order: workgroups are dealt round-robin over the 8 XCDs (flat id % 8 shares an L2), so give each XCD a
// CONTIGUOUS run of the work list -- the column tiles that re-read one A row panel then hit the same L2 instead of
// fetching the panel once per XCD.  (Bijective for any count; placement is a speed matter only.)
__device__ inline int xcd_order(int nwg, int id) {
  const int q = nwg >> 3, rr = nwg & 7, xcd = id & 7;
  return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (id >> 3);
}
__device__ inline int xcd_tile_id() { return xcd_order(gridDim.x, blockIdx.x); }

// Split-K (grid = tiles x 1 x slabs): the work list is slab-major, so an XCD's run is whole slabs (and at most two partial
// ones) with all their tiles, and both operand slabs enter one L2 once.  With blockIdx.z as the slab an XCD saw ONE row
// panel of tiles for every slab, and each slab of the operand that all row panels share was fetched by all 8 L2s (PMC:
// 830 MB per launch of the 63500-token dW[1024,256] against 325 MB of operands; now 329 MB).
__device__ inline void tile_and_split(GemmArgs& g, int& wgid) {
  if (gridDim.z == 1) { wgid = xcd_tile_id(); g.split = 0; return; }
  const int nt = gridDim.x, w = xcd_order(nt * (int)gridDim.z, (int)blockIdx.z * nt + (int)blockIdx.x);
  g.split = w / nt;
  wgid = w - g.split * nt;
}

// Epilogue of a 128x128 tile held as 2x2 MFMA 32x32 accumulators per wave (4 waves as 2x2):
// alpha, bias, ReLU / ReLU-mask, dropout, residual, accumulate -- or split-K atomics.  EB = row
// segments in flight per lane in the vector path (register budget).  `lds` is the kernel's staging
// memory (>= EPILOGUE_LDS_FLOATS floats); every wave must be past its last read of it.
// WN = waves along n (the workgroup has 2 * WN waves, tile 128 x 64 WN; needs 2 * WN * 32 * 68 floats of LDS).
template <int EB, int WN = 2>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, floatx16 (&acc)[2][2], float* lds, int m0, int n0,
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
    if (m0 + BM <= g.M && n0 + 64 * WN <= g.N) {
      // A tile wholly inside C takes a straight-line epilogue specialised on what is fused.  In the general loop below the
      // flag tests are branches, and at their joins the compiler's count of outstanding memory operations is merged
      // conservatively: every batch there starts with s_waitcnt vmcnt(0), i.e. waits until the previous batch's stores
      // have been acknowledged -- eight round trips to memory in sequence per 32 rows.
      const size_t row_base = (size_t)(m0 + wm * 64 + (lane >> 4));
      float* __restrict__ dst = g.C + row_base * g.ldc + gcol;
      const float* __restrict__ auxp = need_aux ? aux + row_base * g.ldr + gcol : nullptr;
      const uint32_t didx = (uint32_t)row_base * (uint32_t)g.N + (uint32_t)gcol;
      auto run = [&](auto AUX_, auto ACC_, auto DROP_) __attribute__((always_inline)) {
        constexpr int AUX = decltype(AUX_)::value;        // 0 none, 1 residual, 2 saved activation (ReLU mask)
        constexpr bool ACC = decltype(ACC_)::value, DROPE = decltype(DROP_)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) T[crow(q, hh) * ES + 32 * j + r] = acc[i][j][q];
#pragma unroll
          for (int it0 = 0; it0 < 8; it0 += EB) {
            float4 m[EB], old[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
              const int rr = 32 * i + 4 * (it0 + u);
              if (AUX) m[u] = *reinterpret_cast<const float4*>(auxp + (size_t)rr * g.ldr);
              if (ACC) old[u] = *reinterpret_cast<const float4*>(dst + (size_t)rr * g.ldc);
            }
#pragma unroll
            for (int u = 0; u < EB; ++u) {
              const int rr = 32 * i + 4 * (it0 + u);
              const float4 v = *reinterpret_cast<const float4*>(&T[((it0 + u) * 4 + (lane >> 4)) * ES + c]);
              float o[4] = {v.x * g.alpha + bv.x, v.y * g.alpha + bv.y, v.z * g.alpha + bv.z, v.w * g.alpha + bv.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {             // a select, not a branch (and not fmaxf against -inf: NaN must pass)
                const float y = fmaxf(o[e], 0.f);
                o[e] = relu ? y : o[e];
              }
              if (AUX == 2) {
                const float mm[4] = {m[u].x, m[u].y, m[u].z, m[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = mm[e] > 0.f ? o[e] : 0.f;
              } else {
                if (DROPE) {
                  uint32_t h1, h2;
                  drop_words(de, (didx + (uint32_t)rr * (uint32_t)g.N) >> 2, h1, h2);
                  o[0] *= (h1 & 0xFFFFU) >= de.thresh ? de.scale : 0.f;
                  o[1] *= (h1 >> 16) >= de.thresh ? de.scale : 0.f;
                  o[2] *= (h2 & 0xFFFFU) >= de.thresh ? de.scale : 0.f;
                  o[3] *= (h2 >> 16) >= de.thresh ? de.scale : 0.f;
                }
                if (AUX == 1) { o[0] += m[u].x; o[1] += m[u].y; o[2] += m[u].z; o[3] += m[u].w; }
              }
              if (ACC) { o[0] += old[u].x; o[1] += old[u].y; o[2] += old[u].z; o[3] += old[u].w; }
              *reinterpret_cast<float4*>(dst + (size_t)rr * g.ldc) = make_float4(o[0], o[1], o[2], o[3]);
            }
          }
        }
      };
      using std::integral_constant;
      using F = integral_constant<bool, false>; using Y = integral_constant<bool, true>;
      using A0 = integral_constant<int, 0>; using A1 = integral_constant<int, 1>; using A2 = integral_constant<int, 2>;
      const int kind = relumask ? 2 : (need_aux ? 1 : 0);
      if (kind == 0 && !accum && !de.on) { run(A0{}, F{}, F{}); return; }     // bias (+ReLU): projections, generator
      if (kind == 0 && !accum && de.on) { run(A0{}, F{}, Y{}); return; }      // + dropout: first FFN layer
      if (kind == 1 && !accum && de.on) { run(A1{}, F{}, Y{}); return; }      // + dropout + residual: sublayer outputs
      if (kind == 1 && !accum && !de.on) { run(A1{}, F{}, F{}); return; }     // the same in eval mode
      if (kind == 2 && !accum) { run(A2{}, F{}, F{}); return; }               // gradient through ReLU
      if (kind == 0 && accum && !de.on) { run(A0{}, Y{}, F{}); return; }      // gradient added to the residual stream's
      if (kind == 2 && accum) { run(A2{}, Y{}, F{}); return; }
    }
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
  if (splitk) {
    // Its own block, and branch-free for a tile that lies wholly inside C.  In the general loop below, and behind any
    // per-element range test, the compiler starts every element's block with s_waitcnt vmcnt(0) (its bookkeeping of the
    // pending bias / residual loads is merged conservatively at each join): the 64 atomics of a lane became 64 memory round
    // trips in sequence, 20-29 us at the end of every weight-gradient launch.  Straight-line, nothing waits between them.
    const bool inside = m0 + BM <= g.M && n0 + 64 * WN <= g.N;
    const int col0 = n0 + wn * 64 + r, row0 = m0 + wm * 64;
    float bv[2] = {0.f, 0.f};
    if (g.bias && lead) {
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = col0 + 32 * j < g.N ? g.bias[col0 + 32 * j] : 0.f;
    }
    float* __restrict__ dst = g.C + (size_t)row0 * g.ldc + col0;
    if (inside) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q)
            atomicAdd(dst + (size_t)(32 * i + crow(q, hh)) * g.ldc + 32 * j, acc[i][j][q] * g.alpha + bv[j]);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (col0 + 32 * j < g.N && row0 + 32 * i + crow(q, hh) < g.M)
              atomicAdd(dst + (size_t)(32 * i + crow(q, hh)) * g.ldc + 32 * j, acc[i][j][q] * g.alpha + bv[j]);
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
