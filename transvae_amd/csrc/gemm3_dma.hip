// tvae_gemm, dtype TVAE_F32X3, "streamed" NT kernel: fp32 activations x pre-split weights with NO register staging and NO
// LDS stores.
//
// gemm3.hip brings both operands through registers: per thread and 16-deep k-step two float4 + three 16-byte loads, the
// 3-way split of the activation values, six ds_write_b128 (13 cycles each on the LDS store path) and one barrier, at two
// workgroups per CU.  Its matrix cores are busy 44 % of the time (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES; a bare loop of
// the same fragment reads and MFMAs: 100 % at 292 fp32-equivalent TFLOP/s, tools/debug/mfma_x3.hip).  Here, for the
// products that have a weight on the B side (forward and data gradient):
//   * the activation operand A (fp32 in HBM) travels global -> LDS by LDS-DMA as it is (64-byte rows per k-tile, chunk
//     swizzle on the source side as in gemm_dma.hip), three k-tiles resident: one being read, two in flight;
//   * the weight planes (tvae_split3, tile-major: gemm3_split.h) are copied piece by piece -- one DMA instruction moves the
//     1 KB piece [64 rows][8 bf16] of one (k-half, plane), which lands as [row][16 bytes]: conflict-free for the MFMA
//     fragment reads without any swizzle;
//   * the lanes that feed the MFMAs read their 8 consecutive k of A as two ds_read_b128 and split them in registers
//     (hi / mid / lo, exact) -- for the NEXT k-tile, in the issue gaps of the current tile's 24 MFMAs; the weight fragments
//     of the next tile are fetched at the same time.  The MFMAs of a k-tile therefore start right behind the barrier from
//     registers, and that barrier is the only synchronisation of a k-step (counted s_waitcnt vmcnt in front of it).
// Wave layout: the four waves of a workgroup take 32 rows x 128 columns each (not 64 x 64): every activation value is then
// split by exactly ONE wave -- with 64 x 64 wave tiles the two waves that share a row block both split it, ~100 vector
// instructions per wave and k-step beside 24 MFMAs, and the vector issue port (which MFMAs share) was as busy as the matrix
// pipe (measured with parts of the kernel removed, tools/debug/x3_parts.py: 175 us whole, 102 us WITHOUT its MFMAs on
// 63500 x 256 x 1024).  Per wave and k-step now: 2 + 12 fragment reads, ~46 vector instructions, 24 MFMAs.
// Limits (gemm.hip falls back to gemm3.hip otherwise): NT, TVAE_GEMM_B_SPLIT3, K % 16 == 0, 16-byte aligned A rows, no
// split-K, no A-side dropout mask.  Rows of A beyond M are read from the last valid row, weight row blocks beyond N from
// the last block (padded with zeros by tvae_split3): both only reach C rows / columns that are not stored.
#include <stdlib.h>
#include "common.h"
#include "gemm_common.h"
#include "gemm3_split.h"

#ifndef TVAE_X3DBG
#define TVAE_X3DBG 0
#endif

namespace {

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

constexpr int XBK = 16;
constexpr int XA_ROWB = XBK * 4;                    // 64-byte fp32 rows
constexpr int XA_BYTES = BM * XA_ROWB;              // 8 KB
constexpr int XB_PIECE = BN * 16;                   // one (k-half, plane) of 128 weight rows: 2 KB
constexpr int XB_BYTES = 6 * XB_PIECE;              // 12 KB
constexpr int XSTAGE = XA_BYTES + XB_BYTES;         // 20 KB
#ifndef TVAE_X3_STAGES
#define TVAE_X3_STAGES 3
#endif
constexpr int XS = TVAE_X3_STAGES;                  // k-tiles resident: one being read, XS - 1 in flight
constexpr int X_LDS = XS * XSTAGE;                  // 60 KB (80 KB with 4): two workgroups per CU
static_assert(XS == 3 || XS == 4, "stages");
constexpr int XA_INSTR = XA_BYTES / 1024 / 4, XB_INSTR = XB_BYTES / 1024 / 4;      // DMA instructions per wave and k-tile: 2 + 3
constexpr int X_PER_TILE = XA_INSTR + XB_INSTR;
static_assert(X_PER_TILE == 5, "the vmcnt literals below");
static_assert(X_LDS >= 4 * 32 * 68 * 4, "the epilogue slabs must fit");

__device__ inline int xswz(int row) { return (row >> 2) & 3; }      // chunk swizzle of a 64-byte row (gemm_dma.hip)

// LDS reads of DMA-written tiles go through inline asm (hipcc would put s_waitcnt vmcnt(0) in front of an ordinary
// ds_read that follows an LDS-DMA); ordered by the counted vmcnt + barrier above them, retired by xlds_done()
template <int OFF = 0>
__device__ inline float4 xlds_read16(unsigned addr) {
  float4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
  return v;
}
__device__ inline void xlds_done() {
  __builtin_amdgcn_sched_barrier(0);        // what was placed in front of the wait stays there (register-only MFMAs would drift)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// the value must exist in its register HERE: keeps the split of the next tile inside the block of the current tile's MFMAs
// (left alone, the compiler sinks it to its first use, behind the next barrier, where nothing overlaps it)
__device__ inline void xkeep(bf16x8& v) { asm volatile("" : "+v"(v)); }

struct XFrag { bf16x8 a[3]; float4 b[4][3]; };      // a[plane] of the wave's 32 rows; b[32-column block][plane] as the asm reads deliver it (bf16 x 8)

// Epilogue of the row-wise wave layout: wave w holds rows 32 w .. 32 w + 31 of the tile and all 128 columns,
// acc[i][j] = columns 64 i + 32 j .. + 31.  The same fused operations as gemm_epilogue (alpha, bias, ReLU / ReLU mask, dropout,
// residual, accumulate; no split-K: the streamed kernel never sums slabs), turned through the wave's own LDS slab
// (32 x 68 floats) so that a lane leaves with float4 stores and 16 lanes cover one 256-byte row segment.
__device__ __forceinline__ void x3_epilogue(const GemmArgs& g, floatx16 (&acc)[2][2], float* lds, int m0, int n0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const bool relu = g.flags & TVAE_GEMM_RELU, accum = g.flags & TVAE_GEMM_ACCUM, relumask = g.flags & 64;
  DropCfg de = drop_cfg((g.flags & TVAE_GEMM_A_DROPMASK) ? nullptr : g.rng, g.site, g.p);
  const int row0 = m0 + wave * 32;
  if (g.flags & FLAG_VEC_EPILOGUE) {
    __syncthreads();                                   // every wave is past its last fragment read: the staging LDS is free
    constexpr int ES = 68;
    float* T = lds + wave * (32 * ES);
    const int c = (lane & 15) * 4;
    const float* __restrict__ aux = g.res;             // residual or saved activation; must not alias C
    const bool need_aux = aux != nullptr;
    if (m0 + BM <= g.M && n0 + BN <= g.N) {
      // a tile wholly inside C: straight-line code specialised on what is fused (see gemm_epilogue)
      auto run = [&](auto AUX_, auto ACC_, auto DROP_) __attribute__((always_inline)) {
        constexpr int AUX = decltype(AUX_)::value;     // 0 none, 1 residual, 2 saved activation (ReLU mask)
        constexpr bool ACC = decltype(ACC_)::value, DROPE = decltype(DROP_)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int gcol = n0 + 64 * i + c;
          const size_t row_base = (size_t)(row0 + (lane >> 4));
          float* __restrict__ dst = g.C + row_base * g.ldc + gcol;
          const float* __restrict__ auxp = AUX ? aux + row_base * g.ldr + gcol : nullptr;
          const uint32_t didx = (uint32_t)row_base * (uint32_t)g.N + (uint32_t)gcol;
          float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (g.bias) bv = *reinterpret_cast<const float4*>(g.bias + gcol);
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) T[crow(q, hh) * ES + 32 * j + r] = acc[i][j][q];
#pragma unroll
          for (int it0 = 0; it0 < 8; it0 += 4) {
            float4 m[4], old[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int rr = 4 * (it0 + u);
              if (AUX) m[u] = *reinterpret_cast<const float4*>(auxp + (size_t)rr * g.ldr);
              if (ACC) old[u] = *reinterpret_cast<const float4*>(dst + (size_t)rr * g.ldc);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int rr = 4 * (it0 + u);
              const float4 v = *reinterpret_cast<const float4*>(&T[(rr + (lane >> 4)) * ES + c]);
              float o[4] = {v.x * g.alpha + bv.x, v.y * g.alpha + bv.y, v.z * g.alpha + bv.z, v.w * g.alpha + bv.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {             // a select, not a branch (NaN must pass)
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
      if (kind == 0 && !accum && !de.on) { run(A0{}, F{}, F{}); return; }
      if (kind == 0 && !accum && de.on) { run(A0{}, F{}, Y{}); return; }
      if (kind == 1 && !accum && de.on) { run(A1{}, F{}, Y{}); return; }
      if (kind == 1 && !accum && !de.on) { run(A1{}, F{}, F{}); return; }
      if (kind == 2 && !accum) { run(A2{}, F{}, F{}); return; }
      if (kind == 0 && accum && !de.on) { run(A0{}, Y{}, F{}); return; }
      if (kind == 2 && accum) { run(A2{}, Y{}, F{}); return; }
    }
    // edge tiles and the remaining combinations: the same through range tests
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int gcol = n0 + 64 * i + c;
      const bool col_ok = gcol < g.N;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g.bias && col_ok) bv = *reinterpret_cast<const float4*>(g.bias + gcol);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) T[crow(q, hh) * ES + 32 * j + r] = acc[i][j][q];
      for (int it = 0; it < 8; ++it) {
        const int rl = 4 * it + (lane >> 4);
        const int grow = row0 + rl;
        const bool ok = col_ok && grow < g.M;
        const float4 v = *reinterpret_cast<const float4*>(&T[rl * ES + c]);
        if (!ok) continue;
        float o[4] = {v.x * g.alpha + bv.x, v.y * g.alpha + bv.y, v.z * g.alpha + bv.z, v.w * g.alpha + bv.w};
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        if (need_aux) m = *reinterpret_cast<const float4*>(aux + (size_t)grow * g.ldr + gcol);
        const float mm[4] = {m.x, m.y, m.z, m.w};
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
        float* dst = g.C + (size_t)grow * g.ldc + gcol;
        if (accum) {
          const float4 old = *reinterpret_cast<const float4*>(dst);
          o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
        }
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    return;
  }
  // C (or bias / residual) not float4-addressable: element by element
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 64 * i + 32 * j + r;
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = row0 + crow(q, hh);
        if (row >= g.M) continue;
        float v = acc[i][j][q] * g.alpha + bv;
        if (relu) v = fmaxf(v, 0.f);
        if (relumask) {
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

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void gemm3_dma_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[X_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int ntn = (g.N + BN - 1) / BN;
  const int wgid = xcd_tile_id();
  const int tile_n = wgid % ntn, tile_m = wgid / ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = g.K / XBK;
  const int kt_total = g.ldb / 48;                 // k-tiles of the split weight
  const int nrb = (g.N + 63) >> 6;                 // its 64-row blocks

  // ---- per-lane DMA sources (advanced by one k-tile per issue) ----------------------------------------------
  const unsigned char* asrc[XA_INSTR];
  const unsigned char* bsrc[XB_INSTR];
  int bdst[XB_INSTR];
#pragma unroll
  for (int i = 0; i < XA_INSTR; ++i) {
    const int row = (wave * XA_INSTR + i) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ xswz(row);
    const int gr = min(m0 + row, g.M - 1);
    asrc[i] = reinterpret_cast<const unsigned char*>(g.A) + ((size_t)gr * g.lda) * 4 + c * 16;
  }
#pragma unroll
  for (int i = 0; i < XB_INSTR; ++i) {
    const int q = wave * XB_INSTR + i;             // 0..11: (row block of the tile, piece)
    const int rbl = q / 6, piece = q % 6;
    const int rb = min((n0 >> 6) + rbl, nrb - 1);
    bsrc[i] = reinterpret_cast<const unsigned char*>(g.B) + (((size_t)rb * kt_total) * SPLIT_BLOCK + piece * SPLIT_PIECE) * 2 + lane * 16;
    bdst[i] = XA_BYTES + piece * XB_PIECE + rbl * 1024;
  }
  auto issue = [&](int stage) {
    unsigned char* st = lds + stage * XSTAGE;
#pragma unroll
    for (int i = 0; i < XA_INSTR; ++i) {
      __builtin_amdgcn_global_load_lds((glb_vp)asrc[i], (lds_vp)(st + (wave * XA_INSTR + i) * 1024), 16, 0, 0);
      asrc[i] += XA_ROWB;
    }
#pragma unroll
    for (int i = 0; i < XB_INSTR; ++i) {
      __builtin_amdgcn_global_load_lds((glb_vp)bsrc[i], (lds_vp)(st + bdst[i]), 16, 0, 0);
      bsrc[i] += SPLIT_BLOCK * 2;
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;

  // fragment addresses in stage 0 (LDS byte addresses): the wave's 32 rows of A, all 128 rows (output columns) of B
  unsigned a_addr[2], b_addr[4];
  {
    const int row = wave * 32 + r;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) a_addr[c2] = lds_base + row * XA_ROWB + (((2 * hh + c2) ^ xswz(row)) << 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) b_addr[j] = lds_base + XA_BYTES + (hh * 3) * XB_PIECE + (32 * j + r) * 16;       // + plane * XB_PIECE
  }

  float4 raw[2];                            // the next k-tile's A values of this lane, fp32
  XFrag f0, f1;
  auto fetch = [&](int stage, XFrag& fn) {  // only ISSUES the reads: nothing here may be touched before xlds_done()
    const unsigned st = stage * XSTAGE;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) raw[c2] = xlds_read16(st + a_addr[c2]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned ad = st + b_addr[j];
      fn.b[j][0] = xlds_read16<0>(ad);
      fn.b[j][1] = xlds_read16<XB_PIECE>(ad);
      fn.b[j][2] = xlds_read16<2 * XB_PIECE>(ad);
    }
  };
  auto split_raw = [&](XFrag& fn) {
    Stage8 s;
    s.v[0] = raw[0].x; s.v[1] = raw[0].y; s.v[2] = raw[0].z; s.v[3] = raw[0].w;
    s.v[4] = raw[1].x; s.v[5] = raw[1].y; s.v[6] = raw[1].z; s.v[7] = raw[1].w;
    Split8 o;
    split8(s, o);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 h = {o.h[0], o.h[1], o.h[2], o.h[3]}, m = {o.m[0], o.m[1], o.m[2], o.m[3]}, l = {o.l[0], o.l[1], o.l[2], o.l[3]};
    fn.a[0] = __builtin_bit_cast(bf16x8, h);
    fn.a[1] = __builtin_bit_cast(bf16x8, m);
    fn.a[2] = __builtin_bit_cast(bf16x8, l);
  };
  // acc[j >> 1][j & 1] = the wave's 32 rows x columns 32 j .. 32 j + 31
#if TVAE_X3DBG == 2
#define TVAE_X3(F, PA, PB)
#else
#define TVAE_X3(F, PA, PB)                                                                  \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[j >> 1][j & 1] =                        \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[PA], __builtin_bit_cast(bf16x8, F.b[j][PB]), acc[j >> 1][j & 1], 0, 0, 0);
#endif

  int st_cur = 0;                           // stage of the tile in `cur`
  // one k-step that has a successor: tile t is in `cur` (registers), tile t + 1 becomes visible at the barrier
  auto step = [&](int t, XFrag& cur, XFrag& nxt) {
    // tile t + 1 has landed once only the tiles issued after it (t + 2 .. t + XS - 1, as far as they exist) are outstanding
    if (XS == 4 && t + 3 < nkt) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (t + 2 < nkt) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // ... for every wave's share; and nobody reads stage t % 3 any more
#if TVAE_X3DBG != 1     // debug builds (tools/debug/x3_parts.py): 1 = no DMA in the loop, 2 = no MFMAs, 3 = no epilogue
    if (t + XS < nkt) issue(st_cur);
#endif
    st_cur = st_cur == XS - 1 ? 0 : st_cur + 1;      // = (t + 1) % XS
    fetch(st_cur, nxt);
    __builtin_amdgcn_sched_barrier(0);
    TVAE_X3(cur, 2, 0) TVAE_X3(cur, 0, 2)    // smallest terms first
    xlds_done();
    split_raw(nxt);
    TVAE_X3(cur, 1, 1) TVAE_X3(cur, 1, 0) TVAE_X3(cur, 0, 1) TVAE_X3(cur, 0, 0)
#pragma unroll
    for (int i = 0; i < 16; ++i) {          // the split rides in the MFMAs' issue gaps
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) xkeep(nxt.a[p]);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto last = [&](XFrag& cur) {
    TVAE_X3(cur, 2, 0) TVAE_X3(cur, 0, 2) TVAE_X3(cur, 1, 1) TVAE_X3(cur, 1, 0) TVAE_X3(cur, 0, 1) TVAE_X3(cur, 0, 0)
  };

  // prologue: three tiles on their way, tile 0 into registers
#pragma unroll
  for (int t = 0; t < XS; ++t)
    if (t < nkt) issue(t);
  if (XS == 4 && nkt >= 4) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if (nkt >= 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (nkt == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  fetch(0, f0);
  xlds_done();
  split_raw(f0);
  __builtin_amdgcn_sched_barrier(0);

  int t = 0;
  for (; t + 2 < nkt; t += 2) {
    step(t, f0, f1);
    step(t + 1, f1, f0);
  }
  if (t + 1 < nkt) {
    step(t, f0, f1);
    last(f1);
  } else {
    last(f0);
  }
#undef TVAE_X3
  // every DMA has landed (vmcnt(0) in the last step); gemm_epilogue starts with a barrier before it reuses the LDS
#if TVAE_X3DBG == 3
  if (g.M < 0)
#endif
  x3_epilogue(g, acc, reinterpret_cast<float*>(lds), m0, n0);
#if TVAE_X3DBG == 2      // keep what fed the removed MFMAs alive
  if (g.M < 0) g.C[tid] = __builtin_bit_cast(float4, f0.a[0]).x + f1.b[1][2].x + __builtin_bit_cast(float4, f1.a[2]).y + f0.b[3][0].w;
#endif
}

}  // namespace

// called by tvae_gemm (gemm.hip) after it has checked the limits above
void tvae_gemm3_dma_launch(const void* gp, hipStream_t s) {
  const GemmArgs& g = *static_cast<const GemmArgs*>(gp);
  const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
  hipLaunchKernelGGL(gemm3_dma_kernel, dim3(tiles), dim3(256), 0, s, g);
}
