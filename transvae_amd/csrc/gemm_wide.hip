// tvae_gemm, wide-tile exact-fp32 kernel: 128 x 160 output tile for products whose N is just above 128.
//
// The conv / deconv bottleneck (transvae/trans_models.py:957-1009) has channel counts 148 and 160 (trans4x-256), so its
// conv-as-GEMM products have N = 148 / 160 with K in the thousands:
//     conv0 forward         63492 x 160 x 2304      deconv2 data gradient   68500 x 148 x 2816
//     conv1 data gradient   29500 x 160 x  896      conv0 weight gradient   2304 x 160 x 63492 (operands swapped, see ops.py)
// On 128-wide column tiles the second tile of such a launch is 15-25 % inside C: 37-42 % of the MFMAs and of the A-panel
// staging produce nothing (60-77 TFLOP/s where full tiles reach 105-118).  Skipping the empty 32 x 32 blocks inside the
// 128 x 128 kernel does not help (a mostly-empty tile stages as many bytes as a full one).  Here the
// tile is as wide as N: four waves stacked along m, each 32 rows x 160 columns = 1 x 5 MFMA 32x32 blocks (80 accumulator
// registers), so the A panel is staged ONCE and every MFMA issued lands in C for N = 160 (92.5 % for N = 148).
//
// v_mfma_f32_32x32x2_f32 only (exact fp32): with K >= 896 a tile runs >= 56 k-steps of 40 MFMAs per wave, so prologue and
// epilogue are a few percent and stay simple -- alpha, bias, ReLU, or split-K atomics, stored straight from the
// accumulators.  Both operands row-major (NT) or both k-major (TN, the weight-gradient form); aligned operands only (the
// host wrapper in gemm.hip checks and otherwise keeps the 128 x 128 kernel).  k-depth 16, register-staged double
// buffering and the lane-half k permutation are those of gemm_kernel (gemm.hip); LDS 46 KB, 3 workgroups per CU.
#include "common.h"
#include "gemm_common.h"

namespace {

constexpr int WBM = 128, WBN = 160, WBK = 16;
constexpr int W_KC = WBK + 4;                    // [row][k] stride (floats)
constexpr int W_AKM = WBM + 4, W_BKM = WBN + 4;  // [k][row] strides
constexpr int W_A_FLOATS = (WBM * W_KC > WBK * W_AKM) ? WBM * W_KC : WBK * W_AKM;
constexpr int W_B_FLOATS = (WBN * W_KC > WBK * W_BKM) ? WBN * W_KC : WBK * W_BKM;

// position of float4 number f of a (rows x 16) operand tile: row-major tiles are walked row by row (4 float4 per row),
// k-major tiles k-row by k-row (rows / 4 float4 per k)
template <bool KM, int ROWS>
__device__ inline void wide_pos(int f, int& row, int& k) {
  if (!KM) { row = f >> 2; k = (f & 3) * 4; } else { k = f / (ROWS / 4); row = (f % (ROWS / 4)) * 4; }
}

// NF float4 per thread; float4 number tid + 256 j exists iff it is < ROWS * 4 (wave-uniform: ROWS * 4 is a multiple of 128)
template <bool KM, int ROWS, int NF>
struct WideStage {
  unsigned off[NF];
  float4 v[NF];
  __device__ inline void offsets(int ld, int row0, int nrows, int tid) {
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      int row, k;
      wide_pos<KM, ROWS>(tid + 256 * j, row, k);
      // rows past the operand are clamped to its last row (group): their products only reach C rows / columns that are
      // not stored
      const int gr = min(row0 + row, KM ? nrows - 4 : nrows - 1) - row0;
      off[j] = 4u * (KM ? (unsigned)k * (unsigned)ld + (unsigned)gr : (unsigned)gr * (unsigned)ld + (unsigned)k);
    }
  }
  __device__ inline void load(const float* __restrict__ base, int tid) {
#pragma unroll
    for (int j = 0; j < NF; ++j)
      if (tid + 256 * j < ROWS * 4) v[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off[j]);
  }
  // the partial last k-step: elements at k >= kleft are zero (their addresses are not read)
  __device__ inline void load_checked(const float* __restrict__ base, int ld, int kleft, int tid) {
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      if (tid + 256 * j >= ROWS * 4) continue;
      int row, k;
      wide_pos<KM, ROWS>(tid + 256 * j, row, k);
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      if (KM) {
        if (k < kleft) {
          const float4 q = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off[j]);
          t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w;
        }
      } else {
        const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off[j]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < kleft) t[e] = p[e];
      }
      v[j] = make_float4(t[0], t[1], t[2], t[3]);
    }
  }
  __device__ inline void store(float* __restrict__ T, int tid) const {
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      if (tid + 256 * j >= ROWS * 4) continue;
      int row, k;
      wide_pos<KM, ROWS>(tid + 256 * j, row, k);
      if (!KM) *reinterpret_cast<float4*>(&T[row * W_KC + k]) = v[j];
      else *reinterpret_cast<float4*>(&T[k * (ROWS + 4) + row]) = v[j];
    }
  }
};

template <bool KM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
void gemm_wide_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (W_A_FLOATS + W_B_FLOATS)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int ntn = (g.N + WBN - 1) / WBN;
  int wgid;
  tile_and_split(g, wgid);
  const int tile_n = wgid % ntn, tile_m = wgid / ntn;
  const int m0 = tile_m * WBM, n0 = tile_n * WBN;
  const int kbeg = g.split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nfull = (kend - kbeg) / WBK, kleft = (kend - kbeg) - nfull * WBK;

  floatx16 acc[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) acc[j] = zero16();

  WideStage<KM, WBM, 2> sa;
  WideStage<KM, WBN, 3> sb;
  sa.offsets(g.lda, m0, g.M, tid);
  sb.offsets(g.ldb, n0, g.N, tid);
  auto baseA = [&](int k0) { return g.A + (KM ? (size_t)k0 * g.lda + m0 : (size_t)m0 * g.lda + k0); };
  auto baseB = [&](int k0) { return g.B + (KM ? (size_t)k0 * g.ldb + n0 : (size_t)n0 * g.ldb + k0); };
  auto tileA = [&](int buf) { return lds + buf * (W_A_FLOATS + W_B_FLOATS); };
  auto tileB = [&](int buf) { return lds + buf * (W_A_FLOATS + W_B_FLOATS) + W_A_FLOATS; };

  auto compute = [&](int buf) {
    const float* TA = tileA(buf);
    const float* TB = tileB(buf);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int kq = 8 * hh + 4 * q;             // the k values of this lane half in sub-step q (same permutation for A and B)
      float a[4], b[5][4];
      const int row = wave * 32 + r;
      if (!KM) {
        const float4 v = *reinterpret_cast<const float4*>(&TA[row * W_KC + kq]);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = TA[(kq + s) * W_AKM + row];
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int col = 32 * j + r;
        if (!KM) {
          const float4 v = *reinterpret_cast<const float4*>(&TB[col * W_KC + kq]);
          b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) b[j][s] = TB[(kq + s) * W_BKM + col];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[j][s], acc[j], 0, 0, 0);
      // the second sub-step's 24 fragment registers must not be fetched above the first's MFMAs (168-register budget of
      // three workgroups per CU)
      if (q == 0) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // steps: nfull whole k-steps, then one range-checked partial step if kleft > 0
  const int nsteps = nfull + (kleft > 0 ? 1 : 0);
  auto load_step = [&](int st) {
    const int k0 = kbeg + st * WBK;
    if (st < nfull) { sa.load(baseA(k0), tid); sb.load(baseB(k0), tid); }
    else { sa.load_checked(baseA(k0), g.lda, kleft, tid); sb.load_checked(baseB(k0), g.ldb, kleft, tid); }
  };
  if (nsteps > 0) {
    load_step(0);
    sa.store(tileA(0), tid);
    sb.store(tileB(0), tid);
  }
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    const bool more = st + 1 < nsteps;
    if (more) load_step(st + 1);                 // only issued: the registers are first touched after the MFMAs
    compute(buf);
    if (more) {
      sa.store(tileA(buf ^ 1), tid);
      sb.store(tileB(buf ^ 1), tid);
    }
    __syncthreads();
  }

  // ---- epilogue: alpha, bias, ReLU -> store, or split-K atomics; lane (r, hh) holds rows crow(q, hh), column 32 j + r ----
  const bool relu = g.flags & TVAE_GEMM_RELU, splitk = g.flags & TVAE_GEMM_SPLITK;
  const bool lead = g.split == 0;
  const int row0 = m0 + wave * 32, col0 = n0 + r;
  float bv[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) bv[j] = (g.bias && lead && col0 + 32 * j < g.N) ? g.bias[col0 + 32 * j] : 0.f;
  float* __restrict__ dst = g.C + (size_t)row0 * g.ldc + col0;
  const bool inside = m0 + WBM <= g.M && n0 + WBN <= g.N;
  // four straight-line blocks (a flag test inside the unrolled loop is a branch whose join makes hipcc wait for the previous
  // store / atomic before the next one, gemm_common.h)
  auto val = [&](int j, int q) { return acc[j][q] * g.alpha + bv[j]; };
  if (inside && splitk) {
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) atomicAdd(dst + (size_t)crow(q, hh) * g.ldc + 32 * j, val(j, q));
  } else if (inside) {
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float v = val(j, q), y = fmaxf(v, 0.f);
        dst[(size_t)crow(q, hh) * g.ldc + 32 * j] = relu ? y : v;
      }
  } else {
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (col0 + 32 * j >= g.N || row0 + crow(q, hh) >= g.M) continue;
        const float v = val(j, q);
        float* p = dst + (size_t)crow(q, hh) * g.ldc + 32 * j;
        if (splitk) atomicAdd(p, v);
        else *p = relu ? fmaxf(v, 0.f) : v;
      }
  }
}

}  // namespace

// called by tvae_gemm (gemm.hip) once shapes, alignment and flags are validated and `g` (k_per_split included) is filled in
void tvae_gemm_wide_launch(const void* gp, int splits, bool km, hipStream_t s) {
  const GemmArgs& g = *reinterpret_cast<const GemmArgs*>(gp);
  dim3 grid(cdiv(g.M, WBM) * cdiv(g.N, WBN), 1, splits);
  if (km) hipLaunchKernelGGL((gemm_wide_kernel<true>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_wide_kernel<false>), grid, dim3(256), 0, s, g);
}
