// Fused multi-head attention for TransVAE shapes (L <= 128, dk in {32, 64}), gfx950.
//
// One 256-thread workgroup (4 waves) per (molecule, head).  The whole score tile lives in
// MFMA accumulators; K and V live in LDS; nothing of size L x L ever touches HBM (unless the
// caller asks for the attention weights).
//
// Orientation trick: the scores are computed TRANSPOSED, S^T = K Q^T, with the query on the
// MFMA lane and the keys in the 16 accumulator registers (x4 key tiles).  A softmax row is
// then 64 in-lane values + one cross-half shuffle (lane ^ 32) -- the "wavefront-shuffle
// softmax" -- and the normalised P^T accumulators ARE the B operand of O^T = V^T P^T
// (v_mfma_f32_32x32x2_f32 takes one fp32 per lane: B[k = lane>>5][j = lane&31]); no LDS
// round trip, no register transposition.
//
// Masking follows the reference exactly (transvae/tvae_util.py:43-45): masked scores are
// REPLACED by -1e9 (not -inf), so a fully masked row attends uniformly, and masked entries get
// no gradient.  Keys beyond Lk (tile padding) are excluded outright.
#include "common.h"

namespace {

struct AttnArgs {
  const float *q, *k, *v, *o, *d_o, *lse_in;
  float *o_out, *lse_out, *p_out, *dq, *dk, *dv, *dsum;
  int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  const uint8_t *key_mask, *full_mask;
  int causal, B, h, Lq, Lk;
  float scale;
  const uint32_t* rng; uint32_t site; float p;
};

constexpr float MASK_FILL = -1e9f;

// -DTVAE_ATTN_STAMPS (tools/debug/attn_stamps.py builds such a library next to the real one): lane 0 of every wave of the
// first 1024 workgroups records the shader clock at a few points of the kernel; tvae_debug_stamps copies them out.
#ifdef TVAE_ATTN_STAMPS
__device__ unsigned long long tvae_stamps[1024 * 4 * 16];
#define STAMP(slot)                                                                                              \
  do {                                                                                                           \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024)                                                            \
      tvae_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_amdgcn_s_memtime();          \
  } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

// exp(x) for softmax weights, x <= 0 (or the -1e9 fill / -inf): one v_mul + one v_exp_f32 instead of expf's twelve
// instructions (range reduction, ldexp, under/overflow selects).  The product x * log2(e) carries a relative error of
// 2^-24 in the ARGUMENT, i.e. |x| 2^-24 relative in the weight: 6e-8 for the keys that dominate a row, 1e-6 for a key whose
// weight is e^-17 of the row maximum -- below fp32 resolution of the row sum either way.  Forward and both backward
// kernels use this one function, so the backward pass recomputes the forward's P bit for bit.
__device__ inline float attn_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// T[128][DK+4] <- src rows 0..nrows-1 (DK floats each), zero beyond.  Split in two so that the loads
// of several tiles are all in flight before the first LDS write waits for them.
template <int DK>
struct TileRegs { float4 v[128 * (DK / 4) / 256]; };

template <int DK>
__device__ inline void tile_issue(TileRegs<DK>& R, const float* __restrict__ src, int ld, int nrows, int tid) {
  constexpr int F4 = DK / 4, NJ = 128 * F4 / 256;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int f = tid + 256 * j, row = f / F4, c = (f % F4) * 4;
    R.v[j] = *reinterpret_cast<const float4*>(src + (size_t)(row < nrows ? row : 0) * ld + c);
  }
}
template <int DK>
__device__ inline void tile_commit(const TileRegs<DK>& R, float* __restrict__ T, int nrows, int tid) {
  constexpr int LS = DK + 4, F4 = DK / 4, NJ = 128 * F4 / 256;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int f = tid + 256 * j, row = f / F4, c = (f % F4) * 4;
    *reinterpret_cast<float4*>(&T[row * LS + c]) = row < nrows ? R.v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// A lane's half row into registers.  Unconditional: the callers clamp the row of a lane beyond the molecule to row 0 (such a
// lane computes a copy of row 0 and stores nothing).  A load under `if (valid)` is a branch whose join makes the compiler
// wait for the load right there -- in the middle of the prologue's load burst, one memory latency per call.
template <int HALF>
__device__ inline void load_row_half(float (&reg)[HALF], const float* __restrict__ rowp) {
#pragma unroll
  for (int s = 0; s < HALF; s += 4) {
    float4 v = *reinterpret_cast<const float4*>(rowp + s);
    reg[s] = v.x; reg[s + 1] = v.y; reg[s + 2] = v.z; reg[s + 3] = v.w;
  }
}

// bf16 arithmetic (dtype TVAE_BF16): the tiles stay fp32 in LDS and in registers and are rounded to bf16 on the way into
// v_mfma_f32_32x32x16_bf16 -- eight consecutive k of the lane's half per instruction, two ds_read_b128 and four
// v_cvt_pk_bf16_f32 per MFMA that does the work of eight fp32 ones.  Softmax, statistics, masks and accumulators are fp32.
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ inline unsigned pk2(float a, float b) {          // one v_cvt_pk_bf16_f32 (round to nearest even)
  bf16x2v v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ inline bf16x8 pack8(const float (&f)[8]) {
  u32x4 w = {pk2(f[0], f[1]), pk2(f[2], f[3]), pk2(f[4], f[5]), pk2(f[6], f[7])};
  return __builtin_bit_cast(bf16x8, w);
}

// acc[i] (+)= sum over this k-tile of A_lds[row][hh*HALF + s] * breg[s]   (32 rows x 32 lanes)
template <int DK, bool BF>
__device__ inline floatx16 mma_rows_x_reg(const float* __restrict__ T, int row, int hh, const float (&breg)[DK / 2]) {
  constexpr int LS = DK + 4, HALF = DK / 2;
  floatx16 acc = zero16();
  const float* p = &T[row * LS + hh * HALF];
  if (BF) {
#pragma unroll
    for (int s = 0; s < HALF; s += 8) {
      const float4 a0 = *reinterpret_cast<const float4*>(p + s), a1 = *reinterpret_cast<const float4*>(p + s + 4);
      const float af[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float bf[8] = {breg[s], breg[s + 1], breg[s + 2], breg[s + 3], breg[s + 4], breg[s + 5], breg[s + 6], breg[s + 7]};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(af), pack8(bf), acc, 0, 0, 0);
    }
    return acc;
  }
#pragma unroll
  for (int s = 0; s < HALF; s += 4) {
    float4 a = *reinterpret_cast<const float4*>(p + s);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, breg[s + 0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, breg[s + 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, breg[s + 2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, breg[s + 3], acc, 0, 0, 0);
  }
  return acc;
}

// out[u] += sum_i  T[(row0 + crow(i,hh))][32u + r] * breg[i]   (contraction over the 32 rows of a tile)
template <int DK, bool BF>
__device__ inline void mma_cols_x_acc(floatx16 (&out)[DK / 32], const float* __restrict__ T, int row0, int hh, int r,
                                      const floatx16& breg) {
  constexpr int LS = DK + 4;
  if (BF) {
    // k slot j of lane half hh in MFMA m is tile row crow(8 m + j, hh), in A and in B alike
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float bf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bf[j] = breg[8 * m + j];
      const bf16x8 B = pack8(bf);
#pragma unroll
      for (int u = 0; u < DK / 32; ++u) {
        float af[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) af[j] = T[(row0 + crow(8 * m + j, hh)) * LS + 32 * u + r];
        out[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(af), B, out[u], 0, 0, 0);
      }
    }
    return;
  }
  // The A operand comes from LDS one dword per MFMA.  Left to itself hipcc issues each pair's ds_read right behind the
  // previous pair's MFMAs and waits for it in front of the next pair: one LDS latency exposed per two MFMAs.  Here the 16
  // values of a half (8 tile rows) are all requested before the half's MFMAs, the second half's before the first half's
  // MFMAs are issued (sched_barrier keeps the compiler from folding the groups back together).
  constexpr int NUc = DK / 32;
  float av[2][8][NUc];
  auto fetch = [&](int half) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float* p = &T[(row0 + crow(8 * half + i, hh)) * LS + r];
#pragma unroll
      for (int u = 0; u < NUc; ++u) av[half][i][u] = p[32 * u];
    }
  };
  auto run = [&](int half) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int u = 0; u < NUc; ++u)
        out[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[half][i][u], breg[8 * half + i], out[u], 0, 0, 0);
  };
  fetch(0);
  __builtin_amdgcn_sched_barrier(0);
  fetch(1);
  __builtin_amdgcn_sched_barrier(0);
  run(0);
  __builtin_amdgcn_sched_barrier(0);
  run(1);
}

template <int DK>
__device__ inline void store_rowmajor(float* __restrict__ rowp, const floatx16 (&acc)[DK / 32], int hh) {
  // acc[u][4g+e] = value for column 32u + 8g + 4hh + e of this lane's row
#pragma unroll
  for (int u = 0; u < DK / 32; ++u)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = make_float4(acc[u][4 * g], acc[u][4 * g + 1], acc[u][4 * g + 2], acc[u][4 * g + 3]);
      *reinterpret_cast<float4*>(rowp + 32 * u + 8 * g + 4 * hh) = v;
    }
}

// Mask of one (query, 32-key tile) as a 32-bit word, bit j = key 32t + j may be attended.
// key_bits[t] holds the (B, Lk) key mask of this molecule (bits past Lk are 0).
template <bool GENERIC>
__device__ inline uint32_t tile_mask(const AttnArgs& a, int b, int qi, int t, const uint32_t* key_bits) {
  uint32_t m = key_bits[t];
  if (a.causal) {
    const int last = qi - 32 * t;                         // keys 32t .. 32t+last are visible
    m &= last < 0 ? 0U : (last >= 31 ? 0xFFFFFFFFU : ((2U << last) - 1U));
  }
  if (GENERIC && a.full_mask) {
    const uint8_t* fm = a.full_mask + ((size_t)b * a.Lq + qi) * a.Lk + 32 * t;
    uint32_t w = 0;
    for (int j = 0; j < 32 && 32 * t + j < a.Lk; ++j) w |= (fm[j] != 0 ? 1U : 0U) << j;
    m &= w;
  }
  return m;
}

// Dead tiles.  A masked score is REPLACED by -1e9, so in a row that has at least one visible key every
// masked key gets the weight exp(-1e9 - max) = 0 exactly: a (32-query wave, 32-key tile) pair whose
// mask words are all zero contributes exact zeros to the row sums, to P V, to dQ, dK and dV, and is
// skipped -- same bits, less work (SMILES rows average 35 of 127 positions; the causal mask kills
// the upper triangle).  The one case where masked keys DO carry weight is a fully masked row
// (uniform attention over all Lk keys, greedy decode with a predicted length of 0): a wave that
// holds such a row skips nothing.
struct Live { bool t[4]; };
__device__ inline Live live_tiles(const uint32_t (&mw)[4]) {
  const bool rows_ok = __ballot((mw[0] | mw[1] | mw[2] | mw[3]) == 0U) == 0ULL;
  Live l;
#pragma unroll
  for (int t = 0; t < 4; ++t) l.t[t] = !rows_ok || __ballot(mw[t] != 0U) != 0ULL;
  return l;
}

// key-mask words of molecule b into LDS (bit j of word t = key 32t+j unmasked and < Lk): wave 0 ballots
__device__ inline void load_key_bits(uint32_t* key_bits, const AttnArgs& a, int b, int tid) {
  // the byte is loaded by every thread from a clamped address (any readable byte when there is no mask): no branch around
  // the load, see load_row_half
  const int kc = min(tid & 127, a.Lk - 1);
  const uint8_t* kp = a.key_mask ? a.key_mask + (size_t)b * a.Lk + kc : reinterpret_cast<const uint8_t*>(a.q);
  const uint8_t kb = *kp;
  if (tid < 128) {
    const bool m = (tid < a.Lk) && (a.key_mask ? kb != 0 : true);
    const unsigned long long bal = __ballot(m);
    if ((tid & 63) == 0) { key_bits[(tid >> 6) * 2] = (uint32_t)bal; key_bits[(tid >> 6) * 2 + 1] = (uint32_t)(bal >> 32); }
  }
}

// Dropout on the attention weights is indexed (row, key) with a row pitch of 128, so the four keys
// 8g + 4hh + {0..3} a lane holds in accumulator registers 4g..4g+3 are one hash group.
__device__ inline uint32_t drop_group(size_t prow, int t, int g, int hh) {
  return (uint32_t)(prow * 32 + 8 * t + 2 * g + hh);
}

// ------------------------------------------------------------------------------- forward
// GENERIC = false: no (B,Lq,Lk) mask and no attention-weight dump -- the train-step instantiation,
// free of the per-element branches those options need.
template <int DK, bool GENERIC, bool BF>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  constexpr int LS = DK + 4, HALF = DK / 2, NU = DK / 32;
  __shared__ __attribute__((aligned(16))) float Ks[128 * LS];
  __shared__ __attribute__((aligned(16))) float Vs[128 * LS];
  __shared__ uint32_t key_bits[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.x / a.h, head = blockIdx.x % a.h;
  const int qi = 32 * wave + r;
  const bool qvalid = qi < a.Lq;
  const int qs = qvalid ? qi : 0;

  float qreg[HALF];
  STAMP(0);
  {
    TileRegs<DK> rk, rv;
    tile_issue<DK>(rk, a.k + (size_t)b * a.Lk * a.ldk + head * DK, a.ldk, a.Lk, tid);
    tile_issue<DK>(rv, a.v + (size_t)b * a.Lk * a.ldv + head * DK, a.ldv, a.Lk, tid);
    load_row_half<HALF>(qreg, a.q + ((size_t)b * a.Lq + qs) * a.ldq + head * DK + hh * HALF);
    load_key_bits(key_bits, a, b, tid);
    STAMP(1);
    tile_commit<DK>(rk, Ks, a.Lk, tid);
    tile_commit<DK>(rv, Vs, a.Lk, tid);
    STAMP(2);
  }
  __syncthreads();
  STAMP(3);

  DropCfg dc = drop_cfg(a.rng, a.site, a.p);
  floatx16 st[4];
  float mx = -INFINITY;
  uint32_t mwf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) mwf[t] = tile_mask<GENERIC>(a, b, qs, t, key_bits);
  const Live live = live_tiles(mwf);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (!live.t[t]) continue;
    st[t] = mma_rows_x_reg<DK, BF>(Ks, 32 * t + r, hh, qreg);
    const uint32_t mw = mwf[t] >> (4 * hh);
    const int nk = a.Lk - 32 * t - 4 * hh;                 // keys of this tile with in-lane offset < nk exist
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int off = crow(i, 0);                          // compile-time
      float s;
      if (off >= nk) s = -INFINITY;
      else s = ((mw >> off) & 1U) ? st[t][i] * a.scale : MASK_FILL;
      st[t][i] = s;
      mx = fmaxf(mx, s);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (!live.t[t]) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float p = attn_exp(st[t][i] - mx);
      st[t][i] = p;
      sum += p;
    }
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  STAMP(4);                                                   // scores and softmax statistics done
  const size_t prow = ((size_t)(b * a.h + head) * a.Lq + qi);
  if (qvalid && hh == 0 && a.lse_out) {
    // (row max, 1 / row sum) kept apart: with the -1e9 fill, max + log(sum) would round the log
    // away in fp32 (a fully masked row would come back as P = 1 instead of 1/Lk), and
    // exp(s - max) * inv reproduces the forward's P bit for bit in the backward kernels
    a.lse_out[2 * prow] = mx;
    a.lse_out[2 * prow + 1] = inv;
  }

  floatx16 oacc[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) oacc[u] = zero16();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (!live.t[t]) {
      if (GENERIC && a.p_out && qvalid) {          // the dump still gets its zeros
        for (int i = 0; i < 16; ++i) {
          const int key = 32 * t + crow(i, hh);
          if (key < a.Lk) a.p_out[prow * a.Lk + key] = 0.f;
        }
      }
      continue;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint32_t keep = 0xFU;
      if (dc.on) keep = drop_keep4(dc, drop_group(prow, t, g, hh));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * g + e;
        float p = st[t][i] * inv;
        if (GENERIC && a.p_out) {
          const int key = 32 * t + crow(i, hh);
          if (qvalid && key < a.Lk) a.p_out[prow * a.Lk + key] = p;
        }
        if (dc.on) p = ((keep >> e) & 1U) ? p * dc.scale : 0.f;
        st[t][i] = p;
      }
    }
    mma_cols_x_acc<DK, BF>(oacc, Vs, 32 * t, hh, r, st[t]);
  }
  STAMP(13);
  if (qvalid) store_rowmajor<DK>(a.o_out + ((size_t)b * a.Lq + qi) * a.ldo + head * DK, oacc, hh);
  STAMP(14);
}

// ------------------------------------------------------------------------------- backward: dQ
// Same orientation as the forward (query on the lane), ONE pass: P and dP of the whole row stay in
// registers (128 of them), dsum = sum_j P_j dP_j / sum_j P_j, dS = P (dP - dsum), dQ^T += K^T dS^T.
// dsum is formed from the SAME P and dP values that enter dS, so sum_j dS_j vanishes to rounding.
// (The flash-attention shortcut dsum = rowsum(dO * O) is the same number mathematically, but its
// rounding error is common to every key of the row and gets multiplied by the MEAN key; with the
// near-uniform source attention of this model that error was 3-4x the reference's own.)
// Also leaves, for the dK/dV kernel: dsum per row and the dropout keep flags of the row as four
// 32-bit words (that kernel holds the key on the lane and would need one hash per element).
template <int DK, bool GENERIC, bool BF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void attn_bwd_dq_kernel(AttnArgs a) {
  constexpr int LS = DK + 4, HALF = DK / 2, NU = DK / 32;
  __shared__ __attribute__((aligned(16))) float Ks[128 * LS];
  __shared__ __attribute__((aligned(16))) float Vs[128 * LS];
  __shared__ uint32_t key_bits[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.x / a.h, head = blockIdx.x % a.h;
  const int qi = 32 * wave + r;
  const bool qvalid = qi < a.Lq;
  const int qs = qvalid ? qi : 0;

  const size_t prow = ((size_t)(b * a.h + head) * a.Lq + qi);
  const size_t prow_s = ((size_t)(b * a.h + head) * a.Lq + qs);   // clamped: a lane beyond Lq uses neither (nk = 0 below)
  const float row_mx = a.lse_in[2 * prow_s];
  const float row_ls = a.lse_in[2 * prow_s + 1];
  DropCfg dc = drop_cfg(a.rng, a.site, a.p);
  floatx16 pt[4], dpt[4];
  uint32_t mwt[4], kwt[4];
  Live live;
  {
    float qreg[HALF], doreg[HALF];
    {
      TileRegs<DK> rk, rv;
      tile_issue<DK>(rk, a.k + (size_t)b * a.Lk * a.ldk + head * DK, a.ldk, a.Lk, tid);
      tile_issue<DK>(rv, a.v + (size_t)b * a.Lk * a.ldv + head * DK, a.ldv, a.Lk, tid);
      load_row_half<HALF>(qreg, a.q + ((size_t)b * a.Lq + qs) * a.ldq + head * DK + hh * HALF);
      load_row_half<HALF>(doreg, a.d_o + ((size_t)b * a.Lq + qs) * a.lddo + head * DK + hh * HALF);
      load_key_bits(key_bits, a, b, tid);
      tile_commit<DK>(rk, Ks, a.Lk, tid);
      tile_commit<DK>(rv, Vs, a.Lk, tid);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) mwt[t] = tile_mask<GENERIC>(a, b, qs, t, key_bits);
    live = live_tiles(mwt);
    {
      // For the dK/dV kernel, which keeps the KEY on the lane: which key tiles of this molecule have a visible key at all,
      // and whether any query row is fully masked (then masked keys carry weight and nothing is dead).  One word per
      // workgroup behind the keep words of the workspace; that kernel reads it first and deals its waves to live tiles.
      const bool wave_full_row = __ballot(qvalid && (mwt[0] | mwt[1] | mwt[2] | mwt[3]) == 0U) != 0ULL;
      const bool any_full_row = __syncthreads_or(wave_full_row);
      if (tid == 0) {
        uint32_t n = 0, list = 0;
        for (int t = 0; t < 4; ++t)
          if (key_bits[t] != 0U) { list |= (uint32_t)t << (2 * n); ++n; }
        const size_t rows_pad_ = ((size_t)a.B * a.h * a.Lq + 3) & ~(size_t)3;
        reinterpret_cast<uint32_t*>(a.dsum)[rows_pad_ + 4 * (size_t)a.B * a.h * a.Lq + blockIdx.x] =
            n | ((any_full_row || (GENERIC && a.full_mask)) ? 0U : 8U) | (list << 4);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (!live.t[t]) continue;
      pt[t] = mma_rows_x_reg<DK, BF>(Ks, 32 * t + r, hh, qreg);
      dpt[t] = mma_rows_x_reg<DK, BF>(Vs, 32 * t + r, hh, doreg);
    }
  }
  float dsum = 0.f, psum = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    kwt[t] = 0U;                     // a dead tile's keep flags only ever multiply exact zeros
    if (!live.t[t]) continue;
    const uint32_t mw = mwt[t] >> (4 * hh);
    mwt[t] = mw;
    const int nk = qvalid ? a.Lk - 32 * t - 4 * hh : 0;
    uint32_t kw = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint32_t keep = 0xFU;
      if (dc.on) keep = drop_keep4(dc, drop_group(prow, t, g, hh));
      kw |= keep << (8 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * g + e, off = crow(i, 0);
        float p = 0.f, dp = 0.f;
        if (off < nk) {
          const float sc = ((mw >> off) & 1U) ? pt[t][i] * a.scale : MASK_FILL;
          p = attn_exp(sc - row_mx) * row_ls;
          dp = ((keep >> e) & 1U) ? dpt[t][i] * dc.scale : 0.f;
          psum += p;
          dsum += p * dp;
        }
        pt[t][i] = p;
        dpt[t][i] = dp;
      }
    }
    kwt[t] = kw << (4 * hh);        // bit (8g + 4hh + e) of word t = keep flag of key 32t + 8g + 4hh + e
  }
  dsum += __shfl_xor(dsum, 32, 64);
  psum += __shfl_xor(psum, 32, 64);
  // normalise by the ACTUAL sum of the P values used below, so that sum_j dS_j = 0 to rounding
  dsum = qvalid ? dsum / psum : 0.f;
  const size_t rows_pad = ((size_t)a.B * a.h * a.Lq + 3) & ~(size_t)3;
  if (qvalid) {
#pragma unroll
    for (int t = 0; t < 4; ++t) kwt[t] |= __shfl_xor(kwt[t], 32, 64);
    if (hh == 0) {
      a.dsum[prow] = dsum;
      uint4 w = make_uint4(kwt[0], kwt[1], kwt[2], kwt[3]);
      *reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(a.dsum) + rows_pad + 4 * prow) = w;
    }
  }

  floatx16 dqacc[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) dqacc[u] = zero16();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (!live.t[t]) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      // masked entries were REPLACED by the fill value in the forward: no gradient through them
      const float ds = ((mwt[t] >> crow(i, 0)) & 1U) ? pt[t][i] * (dpt[t][i] - dsum) * a.scale : 0.f;
      dpt[t][i] = ds;
    }
    mma_cols_x_acc<DK, BF>(dqacc, Ks, 32 * t, hh, r, dpt[t]);
  }
  if (qvalid) store_rowmajor<DK>(a.dq + ((size_t)b * a.Lq + qi) * a.lddq + head * DK, dqacc, hh);
}

// ------------------------------------------------------------------------------- backward: dK, dV
// Key on the lane: S = Q K^T and dP = dO V^T come out as [query regs][key lanes], which is the
// B operand of dV^T = dO^T P and dK^T = Q^T dS directly.  Row statistics, dsum and the dropout keep
// words come from the dQ kernel's workspace (one LDS broadcast read per query).
//
// Waves are DEALT to the live key tiles.  A SMILES row has ~36 of 127 positions, so two of the four key tiles hold no
// visible key at all, and with one wave per key tile two waves of every workgroup had nothing to do while the other two
// walked all four query tiles (the kernel's time was that serial chain).  The dQ kernel leaves one word per workgroup:
// the live key tiles and whether any row is fully masked.  With 2 live key tiles each gets two waves (query tiles 0-1
// and 2-3), with 1 all four waves take one query tile each; the partial dK / dV of the waves that share a key tile are
// added through LDS in a fixed order (Q / dO tiles are dead by then).  3-4 live tiles, a fully masked row or a (B,Lq,Lk)
// mask: one wave per key tile as before.
template <int DK, bool GENERIC, bool BF>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs a) {
  constexpr int LS = DK + 4, HALF = DK / 2, NU = DK / 32;
  __shared__ __attribute__((aligned(16))) float QG[2 * 128 * LS];  // Q tile, then dO tile; afterwards the dK / dV exchange
  float* const Qs = QG;
  float* const Gs = QG + 128 * LS;
  __shared__ __attribute__((aligned(16))) float4 stat_s[128];     // {row max, 1/row sum, dsum, -} per query
  __shared__ __attribute__((aligned(16))) uint32_t keep_s[128][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.x / a.h, head = blockIdx.x % a.h;
  const size_t rows_pad = ((size_t)a.B * a.h * a.Lq + 3) & ~(size_t)3;

  // the workgroup's word from the dQ kernel (uniform: a scalar load), then every vector load of the prologue in one burst
  STAMP(0);
  const uint32_t info = reinterpret_cast<const uint32_t*>(a.dsum)[rows_pad + 4 * (size_t)a.B * a.h * a.Lq + blockIdx.x];
  const int n_live = info & 7;
  const bool skip_ok = (info & 8U) != 0U;
  const int share = (skip_ok && n_live == 2) ? 2 : ((skip_ok && n_live == 1) ? 4 : 1);   // waves per key tile
  int kt = wave, tq0 = 0, tq1 = 4;
  if (share == 2) { kt = (info >> (4 + 2 * (wave >> 1))) & 3; tq0 = 2 * (wave & 1); tq1 = tq0 + 2; }
  if (share == 4) { kt = (info >> 4) & 3; tq0 = wave; tq1 = wave + 1; }
  const int key = 32 * kt + r;
  const bool kvalid = key < a.Lk;
  const int ks = kvalid ? key : 0;

  DropCfg dc = drop_cfg(a.rng, a.site, a.p);
  float kreg[HALF], vreg[HALF];
  TileRegs<DK> rq, rg;
  {
    tile_issue<DK>(rq, a.q + (size_t)b * a.Lq * a.ldq + head * DK, a.ldq, a.Lq, tid);
    tile_issue<DK>(rg, a.d_o + (size_t)b * a.Lq * a.lddo + head * DK, a.lddo, a.Lq, tid);
    load_row_half<HALF>(kreg, a.k + ((size_t)b * a.Lk + ks) * a.ldk + head * DK + hh * HALF);
    load_row_half<HALF>(vreg, a.v + ((size_t)b * a.Lk + ks) * a.ldv + head * DK + hh * HALF);
  }
  // Row statistics, keep words and the key-mask byte: issued with the tiles, from clamped addresses and without branches
  // (they used to follow the tiles as four load -> wait -> load chains: four more memory latencies in every prologue).
  const int srow = min(tid & 127, a.Lq - 1);
  const size_t prow = (size_t)(b * a.h + head) * a.Lq + srow;
  const float st_mx = a.lse_in[2 * prow], st_inv = a.lse_in[2 * prow + 1], st_ds = a.dsum[prow];
  const uint32_t* kwp = dc.on ? reinterpret_cast<const uint32_t*>(a.dsum) + rows_pad + 4 * prow
                              : reinterpret_cast<const uint32_t*>(a.dsum);          // any readable 16 bytes
  const uint4 kw = *reinterpret_cast<const uint4*>(kwp);
  const uint8_t* kmp = a.key_mask ? a.key_mask + (size_t)b * a.Lk + ks : reinterpret_cast<const uint8_t*>(a.q);
  const uint8_t kmb = *kmp;
  STAMP(1);
  tile_commit<DK>(rq, Qs, a.Lq, tid);
  tile_commit<DK>(rg, Gs, a.Lq, tid);
  STAMP(2);
  if (tid < 128) {
    const bool ok = tid < a.Lq;
    stat_s[tid] = ok ? make_float4(st_mx, st_inv, st_ds, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<uint4*>(&keep_s[tid][0]) = (ok && dc.on) ? kw : make_uint4(~0U, ~0U, ~0U, ~0U);
  }
  const bool kbit = kvalid && (a.key_mask ? kmb != 0 : true);
  __syncthreads();
  STAMP(3);

  floatx16 dkacc[NU], dvacc[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) { dkacc[u] = zero16(); dvacc[u] = zero16(); }
#pragma unroll 1
  for (int t = tq0; t < tq1; ++t) {
    // (query tile t, this wave's key tile) is dead when no key of the wave is visible to any query of the tile
    const int q_last = min(32 * t + 31, a.Lq - 1);
    const bool lane_live = kbit && q_last >= 32 * t && (!a.causal || key <= q_last);
    if (skip_ok && __ballot(lane_live) == 0ULL) continue;
    floatx16 s = mma_rows_x_reg<DK, BF>(Qs, 32 * t + r, hh, kreg);
    floatx16 dp = mma_rows_x_reg<DK, BF>(Gs, 32 * t + r, hh, vreg);
    // Per element: the query's {row max, 1 / row sum, dsum} and keep word come from LDS.  They are fetched eight queries at
    // a time BEFORE the arithmetic, which is branch-free (selects): with the reads inside `if (valid)` blocks every element
    // waited for its own LDS round trip (16 exposed latencies per tile, a third of the kernel's time in the stamps).
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 stv[8];
      uint32_t kwv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int qi = 32 * t + crow(8 * half + j, hh);        // < 128: rows past Lq hold zeros / all-ones
        stv[j] = stat_s[qi];
        kwv[j] = keep_s[qi][kt];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = 8 * half + j, qi = 32 * t + crow(i, hh);
        const bool valid = kvalid && qi < a.Lq;
        bool m = kbit && (!a.causal || key <= qi);
        if (GENERIC && a.full_mask) m = m && valid && (a.full_mask[((size_t)b * a.Lq + qi) * a.Lk + key] != 0);
        const float sc = m ? s[i] * a.scale : MASK_FILL;
        const float p = attn_exp(sc - stv[j].x) * stv[j].y;
        const float ksc = ((kwv[j] >> r) & 1U) ? dc.scale : 0.f;
        const float dsv = p * (dp[i] * ksc - stv[j].z) * a.scale;
        s[i] = valid ? p * ksc : 0.f;
        dp[i] = (valid && m) ? dsv : 0.f;
      }
    }
    STAMP(4 + 2 * (t - tq0));            // S, dP and the softmax arithmetic of this query tile are done
    mma_cols_x_acc<DK, BF>(dvacc, Gs, 32 * t, hh, r, s);
    mma_cols_x_acc<DK, BF>(dkacc, Qs, 32 * t, hh, r, dp);
    STAMP(5 + 2 * (t - tq0));            // its dV / dK MFMAs are issued
  }
  STAMP(12);
  if (share > 1) {                       // workgroup-uniform
    // waves that share a key tile: the first of them (the owner) adds the others' partial sums, in wave order
    const int sub = wave % share;        // 0 = owner
    __syncthreads();                     // every wave is done with the Q / dO tiles
    if (sub != 0) {
      float* slot = QG + (size_t)((wave / share) * (share - 1) + (sub - 1)) * (2 * NU * 16 * 64);
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          slot[(u * 16 + i) * 64 + lane] = dkacc[u][i];
          slot[((NU + u) * 16 + i) * 64 + lane] = dvacc[u][i];
        }
    }
    __syncthreads();
    if (sub != 0) {
      // every row of dK / dV is written: the waves without a tile of their own store the zeros of the DEAD key tiles
      // (no visible key: exact zero gradient), the j-th such wave those of the j-th dead tile
      int j = (wave / share) * (share - 1) + (sub - 1), dt = -1;
      for (int t = 0; t < 4; ++t) {
        bool is_live = false;
        for (int l = 0; l < n_live; ++l) is_live = is_live || (int)((info >> (4 + 2 * l)) & 3) == t;
        if (!is_live && j-- == 0) dt = t;
      }
      const int zkey = 32 * dt + r;
      if (dt >= 0 && zkey < a.Lk) {
        floatx16 z[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) z[u] = zero16();
        store_rowmajor<DK>(a.dk + ((size_t)b * a.Lk + zkey) * a.lddk + head * DK, z, hh);
        store_rowmajor<DK>(a.dv + ((size_t)b * a.Lk + zkey) * a.lddv + head * DK, z, hh);
      }
      return;
    }
    for (int o = 0; o < share - 1; ++o) {
      const float* slot = QG + (size_t)((wave / share) * (share - 1) + o) * (2 * NU * 16 * 64);
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          dkacc[u][i] += slot[(u * 16 + i) * 64 + lane];
          dvacc[u][i] += slot[((NU + u) * 16 + i) * 64 + lane];
        }
    }
  }
  STAMP(13);
  if (kvalid) {
    store_rowmajor<DK>(a.dk + ((size_t)b * a.Lk + key) * a.lddk + head * DK, dkacc, hh);
    store_rowmajor<DK>(a.dv + ((size_t)b * a.Lk + key) * a.lddv + head * DK, dvacc, hh);
  }
  STAMP(14);
}

typedef void (*AttnKernel)(AttnArgs);
// (dk == 64, generic, bf16) -> one of the eight instantiations, listed as <64,T,T> <64,T,F> <64,F,T> <64,F,F> <32,...>
AttnKernel pick(bool d64, bool generic, bool bf, AttnKernel a0, AttnKernel a1, AttnKernel a2, AttnKernel a3, AttnKernel b0,
                AttnKernel b1, AttnKernel b2, AttnKernel b3) {
  const AttnKernel t[8] = {a0, a1, a2, a3, b0, b1, b2, b3};
  return t[(d64 ? 0 : 4) + (generic ? 0 : 2) + (bf ? 0 : 1)];
}

int check_common(const char* who, int B, int h, int Lq, int Lk, int dk, const int* lds, int nld, const void* const* ptrs,
                 int nptr, bool dropout) {
  TVAE_CHECK(B > 0 && h > 0 && Lq >= 1 && Lk >= 1 && Lq <= 128 && Lk <= 128, TVAE_E_SHAPE,
             "%s: need 1 <= Lq,Lk <= 128 (got %d,%d), B=%d h=%d", who, Lq, Lk, B, h);
  TVAE_CHECK(dk == 32 || dk == 64, TVAE_E_SHAPE, "%s: dk=%d unsupported (32 or 64)", who, dk);
  for (int i = 0; i < nld; ++i)
    TVAE_CHECK(lds[i] % 4 == 0 && lds[i] >= h * dk, TVAE_E_SHAPE, "%s: leading dimension %d must be a multiple of 4 and >= h*dk", who, lds[i]);
  for (int i = 0; i < nptr; ++i)
    TVAE_CHECK(ptrs[i] && aligned16(ptrs[i]), TVAE_E_NULL, "%s: operand %d is null or not 16-byte aligned", who, i);
  TVAE_CHECK(!dropout || (long)B * h * Lq * Lk < (1L << 32), TVAE_E_SHAPE, "%s: dropout index space exceeds 32 bits", who);
  TVAE_CHECK((long)B * h < (1L << 31), TVAE_E_SHAPE, "%s: B*h exceeds the grid limit", who);
  return TVAE_OK;
}

}  // namespace

#ifdef TVAE_ATTN_STAMPS
extern "C" int tvae_debug_stamps(unsigned long long* host_out, int n) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tvae_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int tvae_attn_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                             float* lse, float* p_out, const uint8_t* key_mask, const uint8_t* full_mask, int causal,
                             int B, int h, int Lq, int Lk, int dk, float scale, const uint32_t* rng, uint32_t site,
                             float p_drop, int dtype, void* stream) {
  int lds[] = {ldq, ldk, ldv, ldo};
  const void* ptrs[] = {q, k, v, o};
  int rc = check_common("tvae_attn_fwd", B, h, Lq, Lk, dk, lds, 4, ptrs, 4, rng && p_drop > 0.f);
  if (rc) return rc;
  TVAE_CHECK(dtype == TVAE_F32 || dtype == TVAE_BF16 || dtype == TVAE_F32X3, TVAE_E_DTYPE, "tvae_attn_fwd: dtype %d", dtype);
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.o_out = o; a.lse_out = lse; a.p_out = p_out;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.key_mask = key_mask; a.full_mask = full_mask; a.causal = causal;
  a.B = B; a.h = h; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.rng = rng; a.site = site; a.p = p_drop;
  hipStream_t s = (hipStream_t)stream;
  const bool generic = full_mask || p_out;
  // dtype BF16: the four products on the bf16 matrix cores; F32 and F32X3 (whose promise is fp32 accuracy): exact fp32 MFMAs
  auto kern = pick(dk == 64, generic, dtype == TVAE_BF16,
                   attn_fwd_kernel<64, true, true>, attn_fwd_kernel<64, true, false>, attn_fwd_kernel<64, false, true>, attn_fwd_kernel<64, false, false>,
                   attn_fwd_kernel<32, true, true>, attn_fwd_kernel<32, true, false>, attn_fwd_kernel<32, false, true>, attn_fwd_kernel<32, false, false>);
  hipLaunchKernelGGL(kern, dim3(B * h), dim3(256), 0, s, a);
  TVAE_LAUNCH_CHECK("tvae_attn_fwd");
  return TVAE_OK;
}

extern "C" int tvae_attn_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                             int ldo, const float* d_o, int lddo, const float* lse, float* dq, int lddq, float* dk_,
                             int lddk, float* dv, int lddv, float* dsum, const uint8_t* key_mask,
                             const uint8_t* full_mask, int causal, int B, int h, int Lq, int Lk, int dk, float scale,
                             const uint32_t* rng, uint32_t site, float p_drop, int dtype, void* stream) {
  int lds[] = {ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv};
  const void* ptrs[] = {q, k, v, o, d_o, dq, dk_, dv};
  int rc = check_common("tvae_attn_bwd", B, h, Lq, Lk, dk, lds, 8, ptrs, 8, rng && p_drop > 0.f);
  if (rc) return rc;
  TVAE_CHECK(lse && dsum && aligned16(dsum), TVAE_E_NULL, "tvae_attn_bwd: lse / dsum workspace missing or not 16-byte aligned");
  TVAE_CHECK(dtype == TVAE_F32 || dtype == TVAE_BF16 || dtype == TVAE_F32X3, TVAE_E_DTYPE, "tvae_attn_bwd: dtype %d", dtype);
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.o = o; a.d_o = d_o; a.lse_in = lse;
  a.dq = dq; a.dk = dk_; a.dv = dv; a.dsum = dsum;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.key_mask = key_mask; a.full_mask = full_mask; a.causal = causal;
  a.B = B; a.h = h; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.rng = rng; a.site = site; a.p = p_drop;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(B * h), block(256);
  const bool bf = dtype == TVAE_BF16, gen = full_mask != nullptr;
  auto kq = pick(dk == 64, gen, bf,
                 attn_bwd_dq_kernel<64, true, true>, attn_bwd_dq_kernel<64, true, false>, attn_bwd_dq_kernel<64, false, true>, attn_bwd_dq_kernel<64, false, false>,
                 attn_bwd_dq_kernel<32, true, true>, attn_bwd_dq_kernel<32, true, false>, attn_bwd_dq_kernel<32, false, true>, attn_bwd_dq_kernel<32, false, false>);
  auto kkv = pick(dk == 64, gen, bf,
                  attn_bwd_dkv_kernel<64, true, true>, attn_bwd_dkv_kernel<64, true, false>, attn_bwd_dkv_kernel<64, false, true>, attn_bwd_dkv_kernel<64, false, false>,
                  attn_bwd_dkv_kernel<32, true, true>, attn_bwd_dkv_kernel<32, true, false>, attn_bwd_dkv_kernel<32, false, true>, attn_bwd_dkv_kernel<32, false, false>);
  hipLaunchKernelGGL(kq, grid, block, 0, s, a);
  hipLaunchKernelGGL(kkv, grid, block, 0, s, a);
  TVAE_LAUNCH_CHECK("tvae_attn_bwd");
  return TVAE_OK;
}
