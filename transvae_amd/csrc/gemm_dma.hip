// tvae_gemm, "streamed" NT kernel: operand tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4), never
// through registers.
//
// Why: with K = d_model = 256 a 128x128 output tile has only 8 k-tiles of 32, and in bf16 arithmetic their MFMAs take
// ~2,000 cycles per wave.  The register-staged kernel (gemm.hip) spends 14,000 instruction-issue cycles per wave and
// tile on everything around them -- 64-bit address arithmetic, range selects, fp32 -> bf16 conversion and ds_write of
// BOTH operands, once per k-tile -- and waits for one global round trip per k-tile: measured 153 us on
// x[63500,256] @ W[1024,256]^T, 2.5 TB/s of HBM traffic, waves parked 50 % / issuing 32 % of their cycles at two
// waves per SIMD (profiles/r2_gemm_sq_bf16.txt): bound by neither the matrix cores nor HBM.  Here
//   * the activation operand A (fp32 in HBM) is copied as it is into LDS, 16 bytes per lane and instruction, three
//     k-tiles deep, with a counted s_waitcnt vmcnt and ONE raw s_barrier per k-tile; the lanes that feed the MFMA read
//     their 8 consecutive k as two ds_read_b128 and round them to bf16 in registers (4 v_cvt_pk_bf16_f32);
//   * the weight operand B is bf16 in memory already (tvae_cast_bf16, once per optimizer step: W and W^T, so that the
//     data gradient dY W is the same NT product against W^T) and goes straight from LDS into the MFMA;
//   * no staging registers, no per-k-tile address arithmetic beyond one pointer increment per DMA.
// LDS images are lane-linear per DMA instruction (the hardware writes base + lane * 16), so bank conflicts are avoided
// on the SOURCE side: the 16-byte chunk that lands at position c' of row `row` is chunk c' ^ f(row) of that row in
// memory, and the fragment reads apply the same involution.  f(row) = (row >> 1) & 7 for 128-byte rows (fp32, BK = 32),
// (row >> 2) & 3 for 64-byte rows: the 16 lanes a ds_read_b128 serves together then hit 16 different 16-byte slots.
//
// The same skeleton in exact fp32 (both operands fp32, v_mfma_f32_32x32x2_f32, BK = 16) is the F32 instantiation.
//
// Limits (the host wrapper falls back to gemm.hip otherwise): NT only, K % BK == 0, 16-byte aligned rows, no split-K,
// no A-side dropout mask (the caller masks the gradient operand first).  Rows of A / B beyond M / N are read from the
// last valid row (always readable) and only ever reach C rows / columns that are not stored.
#include <stdlib.h>
#include "common.h"
#include "gemm_common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

// WN = waves along n: the workgroup is 2 x WN waves of 64 x 64, its tile 128 x (64 WN).  What a CU can take in from
// L2 (~11 bytes per cycle, whatever the hit rate) is what bounds these launches once the staging instructions are gone:
// a 128 x 128 tile re-reads the activation panel once per 128 output columns (763 MB through the CUs for the 326 MB
// FFN w_1 product), a 128 x 256 tile half as often, a 128 x 512 tile a quarter.
template <bool BF16, int WN>
struct Dma {
  static constexpr int BK = BF16 ? 32 : 16;
  static constexpr int NWAVES = 2 * WN, TN = 64 * WN;
  static constexpr int A_ROWB = BK * 4;                     // bytes per A row in LDS (fp32)
  static constexpr int B_ROWB = BF16 ? BK * 2 : BK * 4;     // bf16 weights | fp32
  static constexpr int A_BYTES = BM * A_ROWB, B_BYTES = TN * B_ROWB;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int A_TOTAL = A_BYTES / 1024, B_TOTAL = B_BYTES / 1024;            // DMA instructions per k-tile
  static constexpr int A_INSTR = (A_TOTAL + NWAVES - 1) / NWAVES, B_INSTR = B_TOTAL / NWAVES;   // ... per wave
  // every wave issues the SAME number of DMAs per k-tile: the counted wait below (vmcnt(PER_TILE)) is only right for a wave
  // whose own count per tile is PER_TILE -- one that issued fewer would let a DMA of the tile about to be read stay in flight
  static_assert(B_TOTAL % NWAVES == 0 && A_TOTAL % NWAVES == 0, "DMA split: equal shares per wave");
  static constexpr int S = 3;                               // k-tiles resident (one being read, two in flight / landed)
  static constexpr int EPI = NWAVES * 32 * 68 * 4;
  static constexpr int LDS_BYTES = (S * STAGE > EPI) ? S * STAGE : EPI;
};

// chunk swizzle of a row of ROWB bytes
template <int ROWB>
__device__ inline int swz(int row) {
  return ROWB == 128 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

__device__ inline short bf_bits(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(short, b);
}
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ inline unsigned pk2(float a, float b) {          // one v_cvt_pk_bf16_f32
  bf16x2v v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// LDS reads of DMA-written tiles go through inline asm: hipcc treats every LDS-DMA as a pending write to any LDS
// address and puts s_waitcnt vmcnt(0) in front of an ordinary ds_read that follows one -- which would drain the two
// k-tiles in flight once per k-tile.  The asm reads are ordered by the counted vmcnt + barrier above them and retired
// by the explicit lgkmcnt(0) below them (sched_barrier: register-only MFMAs must not be hoisted above that wait).
__device__ inline float4 lds_read16(unsigned addr) {
  float4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ inline void lds_reads_done() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <bool BF16, int WN>
__global__ __launch_bounds__(128 * WN) __attribute__((amdgpu_waves_per_eu(WN == 8 ? 4 : (BF16 ? 2 : 3))))
void gemm_dma_kernel(GemmArgs g) {
  using D = Dma<BF16, WN>;
  constexpr int BK = D::BK, S = D::S;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[D::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, hh = lane >> 5;
  const int ntn = (g.N + D::TN - 1) / D::TN;
  const int wgid = xcd_tile_id();
  const int tile_n = wgid % ntn, tile_m = wgid / ntn;
  const int m0 = tile_m * BM, n0 = tile_n * D::TN;
  const int nkt = g.K / BK;
  constexpr bool a_mine = true;      // equal shares per wave (static_assert in Dma)

  // ---- per-lane DMA sources (advanced by one k-tile per issue) ----------------------------------------------
  constexpr int A_CPR = D::A_ROWB / 16, B_CPR = D::B_ROWB / 16;       // 16-byte chunks per row
  constexpr int A_RPI = 64 / A_CPR, B_RPI = 64 / B_CPR;               // rows per DMA instruction
  const unsigned char* asrc[D::A_INSTR];
  const unsigned char* bsrc[D::B_INSTR];
#pragma unroll
  for (int i = 0; i < D::A_INSTR; ++i) {
    const int row = (wave * D::A_INSTR + i) * A_RPI + lane / A_CPR;
    const int c = (lane % A_CPR) ^ swz<D::A_ROWB>(row);
    const int gr = min(m0 + row, g.M - 1);
    asrc[i] = reinterpret_cast<const unsigned char*>(g.A) + ((size_t)gr * g.lda) * 4 + c * 16;
  }
#pragma unroll
  for (int i = 0; i < D::B_INSTR; ++i) {
    const int row = (wave * D::B_INSTR + i) * B_RPI + lane / B_CPR;
    const int c = (lane % B_CPR) ^ swz<D::B_ROWB>(row);
    const int gr = min(n0 + row, g.N - 1);
    bsrc[i] = reinterpret_cast<const unsigned char*>(g.B) + ((size_t)gr * g.ldb) * (BF16 ? 2 : 4) + c * 16;
  }
  auto issue = [&](int stage) {
    unsigned char* sa = lds + stage * D::STAGE;
    unsigned char* sb = sa + D::A_BYTES;
#pragma unroll
    for (int i = 0; i < D::A_INSTR; ++i) {
      if (a_mine) __builtin_amdgcn_global_load_lds((glb_vp)asrc[i], (lds_vp)(sa + (wave * D::A_INSTR + i) * 1024), 16, 0, 0);
      asrc[i] += D::A_ROWB;
    }
#pragma unroll
    for (int i = 0; i < D::B_INSTR; ++i) {
      __builtin_amdgcn_global_load_lds((glb_vp)bsrc[i], (lds_vp)(sb + (wave * D::B_INSTR + i) * 1024), 16, 0, 0);
      bsrc[i] += D::B_ROWB;
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;

  // fragment addresses inside a stage (bytes), without the chunk term
  int arow[2], afz[2], brow[2], bfz[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 64 + 32 * i + r;
    arow[i] = row * D::A_ROWB;
    afz[i] = swz<D::A_ROWB>(row);
    const int col = wn * 64 + 32 * i + r;
    brow[i] = D::A_BYTES + col * D::B_ROWB;
    bfz[i] = swz<D::B_ROWB>(col);
  }

#pragma unroll
  for (int t = 0; t < S - 1; ++t)
    if (t < nkt) issue(t);

  constexpr int PER_TILE = D::A_INSTR + D::B_INSTR;     // this wave's DMA instructions per k-tile
  for (int t = 0; t < nkt; ++t) {
    // tile t has landed once at most the DMAs of the tiles issued after it are outstanding; then the barrier makes the
    // other waves' parts visible and tells everybody that stage (t-1) % S is no longer being read
    if (t + 1 < nkt) {
      static_assert(PER_TILE == 6 || PER_TILE == 4 || PER_TILE == 3, "vmcnt literal");
      if (PER_TILE == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (PER_TILE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (t + S - 1 < nkt) issue((t + S - 1) % S);
    const unsigned st = lds_base + (t % S) * D::STAGE;
    if (BF16) {
      // 16 waves (4 per SIMD, 128 registers each) fetch and use one 16-deep k-step at a time; with 2 per SIMD all 12
      // fragment reads of the k-tile are in flight before the first conversion
      constexpr int KS_PER = (WN == 8) ? 1 : 2;
#pragma unroll
      for (int k0 = 0; k0 < 2; k0 += KS_PER) {
        float4 fa[KS_PER][2][2], fb[KS_PER][2];        // [ks][block][half] fp32 A chunks, [ks][block] bf16 B chunks
#pragma unroll
        for (int kk = 0; kk < KS_PER; ++kk)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int ks = k0 + kk;
            const int c0 = 4 * ks + 2 * hh;            // fp32 chunks c0, c0 + 1 = k 16 ks + 8 hh .. + 7
            fa[kk][i][0] = lds_read16(st + arow[i] + ((c0 ^ afz[i]) << 4));
            fa[kk][i][1] = lds_read16(st + arow[i] + (((c0 + 1) ^ afz[i]) << 4));
            const int cb = 2 * ks + hh;                // bf16 chunk = 8 k
            fb[kk][i] = lds_read16(st + brow[i] + ((cb ^ bfz[i]) << 4));
          }
        lds_reads_done();
#pragma unroll
        for (int kk = 0; kk < KS_PER; ++kk) {
          bf16x8 a[2], b[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float4 lo = fa[kk][i][0], hi = fa[kk][i][1];
            const u32x4 w = {pk2(lo.x, lo.y), pk2(lo.z, lo.w), pk2(hi.x, hi.y), pk2(hi.z, hi.w)};
            a[i] = __builtin_bit_cast(bf16x8, w);
            b[i] = __builtin_bit_cast(bf16x8, fb[kk][i]);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
    } else {
      // exact fp32: lane half hh takes chunk 2q + hh of the row (4 k values) for both operands -- any k permutation
      // shared by A and B leaves the product unchanged
      float4 va[BK / 8][2], vb[BK / 8][2];
#pragma unroll
      for (int q = 0; q < BK / 8; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c = 2 * q + hh;
          va[q][i] = lds_read16(st + arow[i] + ((c ^ afz[i]) << 4));
          vb[q][i] = lds_read16(st + brow[i] + ((c ^ bfz[i]) << 4));
        }
      lds_reads_done();
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i][0] = va[q][i].x; a[i][1] = va[q][i].y; a[i][2] = va[q][i].z; a[i][3] = va[q][i].w;
          b[i][0] = vb[q][i].x; b[i][1] = vb[q][i].y; b[i][2] = vb[q][i].z; b[i][3] = vb[q][i].w;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
      }
    }
  }
  // every DMA has landed (vmcnt(0) above); gemm_epilogue starts with a barrier (sync_first) before it reuses the LDS
  gemm_epilogue<(BF16 && WN < 8) ? 4 : 2, WN>(g, acc, reinterpret_cast<float*>(lds), m0, n0, true);
}

// fp32 (rows, cols) -> bf16 copies: out_n (rows, cols) and out_t (cols, rows); 32x32 tiles through LDS
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ w, int ldw, unsigned short* __restrict__ out_n,
                                                        unsigned short* __restrict__ out_t, int rows, int cols) {
  __shared__ unsigned short T[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rr = r0 + ty + 8 * q, cc = c0 + tx;
    unsigned short v = 0;
    if (rr < rows && cc < cols) {
      v = (unsigned short)bf_bits(w[(size_t)rr * ldw + cc]);
      if (out_n) out_n[(size_t)rr * cols + cc] = v;
    }
    T[ty + 8 * q][tx] = v;
  }
  __syncthreads();
  if (out_t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = c0 + ty + 8 * q, rr = r0 + tx;
      if (rr < rows && cc < cols) out_t[(size_t)cc * rows + rr] = T[tx][ty + 8 * q];
    }
  }
}

}  // namespace

// called by tvae_gemm (gemm.hip) after it has checked the limits above
static int dma_wn(int N) {     // TVAE_DMA_WN = 2 | 4 | 8 forces the tile width (experiments); default by N
  static int forced = [] { const char* e = getenv("TVAE_DMA_WN"); return e ? atoi(e) : 0; }();
  if (forced == 2 || forced == 4 || forced == 8) return forced;
  return N >= 512 ? 8 : (N > 128 ? 4 : 2);     // measured on M = 63500: 99 / 118 / 143 us at N = 1024, 29 / 31 / 41 at N = 256
}

void tvae_gemm_dma_launch(const void* gp, bool bf16, hipStream_t s) {
  const GemmArgs& g = *static_cast<const GemmArgs*>(gp);
  const int wn = bf16 ? dma_wn(g.N) : 2;
  const int tiles = cdiv(g.M, BM) * cdiv(g.N, 64 * wn);
  if (!bf16) hipLaunchKernelGGL((gemm_dma_kernel<false, 2>), dim3(tiles), dim3(256), 0, s, g);
  else if (wn == 2) hipLaunchKernelGGL((gemm_dma_kernel<true, 2>), dim3(tiles), dim3(256), 0, s, g);
  else if (wn == 4) hipLaunchKernelGGL((gemm_dma_kernel<true, 4>), dim3(tiles), dim3(512), 0, s, g);
  else hipLaunchKernelGGL((gemm_dma_kernel<true, 8>), dim3(tiles), dim3(1024), 0, s, g);
}

extern "C" int tvae_cast_bf16(const float* w, int ldw, uint16_t* out_n, uint16_t* out_t, int rows, int cols, void* stream) {
  TVAE_CHECK(w && (out_n || out_t), TVAE_E_NULL, "tvae_cast_bf16: null pointer");
  TVAE_CHECK(rows > 0 && cols > 0 && ldw >= cols, TVAE_E_SHAPE, "tvae_cast_bf16: bad shape %d x %d (ld %d)", rows, cols, ldw);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, w, ldw,
                     out_n, out_t, rows, cols);
  TVAE_LAUNCH_CHECK("tvae_cast_bf16");
  return TVAE_OK;
}
