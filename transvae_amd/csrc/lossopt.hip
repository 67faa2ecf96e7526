// trans_vae_loss fused forward+backward, multi-tensor Adam, and the single-token kernels of the
// KV-cached greedy decoder.  All HBM-bound.
#include "common.h"
#include <mutex>

namespace {

__device__ inline float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}

struct LossArgs {
  const int64_t* tokens; const float* logits; int ldl; const float* weights; const float* pred_len;
  const float* mu; const float* logvar; const float* pred_prop; const float* true_prop; const int64_t* true_len; float beta;
  int B, L, n_cls, n_len, d_latent, pad_idx, n_groups;
  float *out5, *dlogits, *dpred_len, *dmu, *dlogvar, *dpred_prop, *scratch;
};


// The reference accumulates the gradients of BATCH_CHUNKS separately-normalised chunk losses
// (trans_models.py:202-236).  One launch handles all chunks: blockIdx.y selects the chunk
// ("group"), every pointer is shifted to that chunk's molecules and B becomes the chunk size.
__device__ inline LossArgs for_group(LossArgs a, int g) {
  const int Bg = a.B / a.n_groups;
  const size_t m0 = (size_t)g * Bg;
  a.tokens += m0 * (a.L + 1);
  a.logits += m0 * a.L * a.ldl;
  if (a.pred_len) a.pred_len += m0 * a.n_len;
  a.mu += m0 * a.d_latent;
  a.logvar += m0 * a.d_latent;
  if (a.pred_prop) { a.pred_prop += m0; a.true_prop += m0; }
  if (a.true_len) a.true_len += m0;
  a.out5 += 5 * g;
  a.scratch += 8 * g;
  if (a.dlogits) a.dlogits += m0 * a.L * a.ldl;
  if (a.dpred_len) a.dpred_len += m0 * a.n_len;
  if (a.dmu) { a.dmu += m0 * a.d_latent; a.dlogvar += m0 * a.d_latent; }
  if (a.dpred_prop) a.dpred_prop += m0;
  a.B = Bg;
  return a;
}

// The 256 logit rows of a workgroup are one contiguous run of 256 * ldl floats (ldl = n_cls = 25 in the model): they are
// copied into LDS with whole-wave coalesced loads and every thread then walks ITS row there (row pitch 25 or any odd
// number of floats: conflict-free) -- a thread walking 25 floats of a 100-byte-strided row in global memory touched 64
// different cache lines per load instruction and the two loss kernels ran at 5-9 % of the HBM rate.
constexpr int LOSS_STAGE_MAX = 33;          // row pitches up to this many floats are staged (256 * 33 * 4 = 33 KB)
__device__ inline const float* stage_logit_rows(const LossArgs& a, float* tile, int row0, int rows, int tid) {
  const int nrow = min(256, rows - row0);
  // only rows without padding columns are staged: with ldl > n_cls the contiguous run would read past the last row's
  // n_cls floats and the write-back would touch padding columns of dlogits (a column view of a wider matrix)
  if (a.ldl != a.n_cls || a.ldl > LOSS_STAGE_MAX || nrow <= 0) return nullptr;
  const float* src = a.logits + (size_t)row0 * a.ldl;
  const int total = nrow * a.ldl;
  for (int i = tid; i < total; i += 256) tile[i] = src[i];
  __syncthreads();
  return tile;
}

// scratch: [0] sum w[y]  [1] sum w[y]*nll  [2] sum nll_len  [3] sum (1+lv-mu^2-e^lv)  [4] sum (pp-tp)^2
__global__ __launch_bounds__(256) void loss_reduce_kernel(LossArgs a0) {
  const LossArgs a = for_group(a0, blockIdx.y);
  __shared__ float red[4];
  __shared__ float tile[256 * LOSS_STAGE_MAX];
  const int tid = threadIdx.x, lane = tid & 63;
  float den = 0.f, num = 0.f;
  const int rows = a.B * a.L;
  const int row = blockIdx.x * 256 + tid;
  const float* staged = stage_logit_rows(a, tile, blockIdx.x * 256, rows, tid);
  if (row < rows) {
    int b = row / a.L, t = row % a.L;
    int y = (int)a.tokens[(size_t)b * (a.L + 1) + t + 1] - 1;
    if (y >= 0 && y < a.n_cls) {
      const float* lg = staged ? staged + tid * a.ldl : a.logits + (size_t)row * a.ldl;
      float mx = -INFINITY;
      for (int c = 0; c < a.n_cls; ++c) mx = fmaxf(mx, lg[c]);
      float se = 0.f;
      for (int c = 0; c < a.n_cls; ++c) se += expf(lg[c] - mx);
      float w = a.weights[y];
      den = w;
      num = w * (mx + logf(se) - lg[y]);
    }
  }
  den = block_sum(den, red);
  num = block_sum(num, red);
  // length-prediction CE: one wave per molecule, molecules strided over the grid's waves
  float nl = 0.f;
  const int gw = blockIdx.x * 4 + (tid >> 6), nw = gridDim.x * 4;
  for (int b = gw; a.pred_len && b < a.B; b += nw) {     // no length head (vae_loss of the RNN models): the term is 0
    int cnt = 0;
    if (a.true_len) {
      cnt = (int)a.true_len[b];
    } else {
      for (int t = lane; t < a.L + 1; t += 64) cnt += (a.tokens[(size_t)b * (a.L + 1) + t] != a.pad_idx);
      cnt = (int)wave_sum((float)cnt);
    }
    const float* pl = a.pred_len + (size_t)b * a.n_len;
    float mx = -INFINITY;
    for (int c = lane; c < a.n_len; c += 64) mx = fmaxf(mx, pl[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < a.n_len; c += 64) se += expf(pl[c] - mx);
    se = wave_sum(se);
    int cls = cnt < 0 ? 0 : (cnt < a.n_len ? cnt : a.n_len - 1);
    if (lane == 0) nl += mx + logf(se) - pl[cls];
  }
  nl = block_sum(nl, red);
  float kl = 0.f, ms = 0.f;
  const int nz = a.B * a.d_latent;
  for (int i = blockIdx.x * 256 + tid; i < nz; i += gridDim.x * 256) {
    float m = a.mu[i], lv = a.logvar[i];
    kl += 1.f + lv - m * m - expf(lv);
  }
  kl = block_sum(kl, red);
  if (a.pred_prop)
    for (int i = blockIdx.x * 256 + tid; i < a.B; i += gridDim.x * 256) {
      float d = a.pred_prop[i] - a.true_prop[i];
      ms += d * d;
    }
  ms = block_sum(ms, red);
  if (tid == 0) {
    atomicAdd(&a.scratch[0], den);
    atomicAdd(&a.scratch[1], num);
    atomicAdd(&a.scratch[2], nl);
    atomicAdd(&a.scratch[3], kl);
    atomicAdd(&a.scratch[4], ms);
  }
}

__global__ __launch_bounds__(256) void loss_grad_kernel(LossArgs a0) {
  const LossArgs a = for_group(a0, blockIdx.y);
  const int tid = threadIdx.x, lane = tid & 63;
  const float den = a.scratch[0];
  const float bce = a.scratch[1] / den;
  const float bmask = a.scratch[2] / a.B;
  const int nz = a.B * a.d_latent;
  float kld = a.beta * -0.5f * (a.scratch[3] / nz);
  const bool kl_nan = isnan(kld);
  if (kl_nan) kld = 0.f;
  const float mse = a.pred_prop ? a.scratch[4] / a.B : 0.f;
  if (blockIdx.x == 0 && tid == 0) {
    a.out5[0] = bce + bmask + kld + mse;
    a.out5[1] = bce; a.out5[2] = bmask; a.out5[3] = kld; a.out5[4] = mse;
  }
  const int rows = a.B * a.L;
  const int row = blockIdx.x * 256 + tid;
  __shared__ float tile[256 * LOSS_STAGE_MAX];
  float* staged = a.dlogits ? const_cast<float*>(stage_logit_rows(a, tile, blockIdx.x * 256, rows, tid)) : nullptr;
  if (row < rows && a.dlogits) {
    int b = row / a.L, t = row % a.L;
    int y = (int)a.tokens[(size_t)b * (a.L + 1) + t + 1] - 1;
    const float* lg = staged ? staged + tid * a.ldl : a.logits + (size_t)row * a.ldl;
    float* dl = staged ? staged + tid * a.ldl : a.dlogits + (size_t)row * a.ldl;    // staged: the gradient replaces the row in LDS
    if (y >= 0 && y < a.n_cls) {
      float mx = -INFINITY;
      for (int c = 0; c < a.n_cls; ++c) mx = fmaxf(mx, lg[c]);
      float se = 0.f;
      for (int c = 0; c < a.n_cls; ++c) se += expf(lg[c] - mx);
      float coef = a.weights[y] / den, inv = 1.f / se;
      for (int c = 0; c < a.n_cls; ++c) dl[c] = coef * (expf(lg[c] - mx) * inv - (c == y ? 1.f : 0.f));
    } else {
      for (int c = 0; c < a.n_cls; ++c) dl[c] = 0.f;
    }
  }
  if (staged) {                                       // the block's gradient rows leave as one coalesced run
    __syncthreads();
    const int row0 = blockIdx.x * 256, total = min(256, rows - row0) * a.ldl;
    float* dst = a.dlogits + (size_t)row0 * a.ldl;
    for (int i = tid; i < total; i += 256) dst[i] = staged[i];
  }
  const int gw = blockIdx.x * 4 + (tid >> 6), nw = gridDim.x * 4;
  if (a.dpred_len && a.pred_len)
    for (int b = gw; b < a.B; b += nw) {
      int cnt = 0;
      if (a.true_len) {
        cnt = (int)a.true_len[b];
      } else {
        for (int t = lane; t < a.L + 1; t += 64) cnt += (a.tokens[(size_t)b * (a.L + 1) + t] != a.pad_idx);
        cnt = (int)wave_sum((float)cnt);
      }
      int cls = cnt < 0 ? 0 : (cnt < a.n_len ? cnt : a.n_len - 1);   // same clamp as loss_reduce_kernel
      const float* pl = a.pred_len + (size_t)b * a.n_len;
      float mx = -INFINITY;
      for (int c = lane; c < a.n_len; c += 64) mx = fmaxf(mx, pl[c]);
      mx = wave_max(mx);
      float se = 0.f;
      for (int c = lane; c < a.n_len; c += 64) se += expf(pl[c] - mx);
      se = wave_sum(se);
      float inv = 1.f / (se * a.B);
      for (int c = lane; c < a.n_len; c += 64)
        a.dpred_len[(size_t)b * a.n_len + c] = expf(pl[c] - mx) * inv - (c == cls ? 1.f / a.B : 0.f);
    }
  if (a.dmu)
    for (int i = blockIdx.x * 256 + tid; i < nz; i += gridDim.x * 256) {
      float m = a.mu[i], lv = a.logvar[i];
      a.dmu[i] = kl_nan ? 0.f : a.beta * m / nz;
      a.dlogvar[i] = kl_nan ? 0.f : a.beta * -0.5f * (1.f - expf(lv)) / nz;
    }
  if (a.pred_prop && a.dpred_prop)
    for (int i = blockIdx.x * 256 + tid; i < a.B; i += gridDim.x * 256)
      a.dpred_prop[i] = 2.f * (a.pred_prop[i] - a.true_prop[i]) / a.B;
}

// ------------------------------------------------------------------------------- Adam
__global__ __launch_bounds__(256) void adam_kernel(void* const* ptrs, const int64_t* sizes, const int32_t* chunk_tensor,
                                                   const int64_t* chunk_start, int n_tensors, const float* hyper,
                                                   float lr_v, float bc1_v, float bc2_v, float gs_v,
                                                   float beta2, float omb1, float omb2, float eps, int zero_grad) {
  const int ti = chunk_tensor[blockIdx.x];
  const int64_t start = chunk_start[blockIdx.x];
  const int64_t n = sizes[ti];
  float* p = (float*)ptrs[ti];
  float* g = (float*)ptrs[n_tensors + ti];
  float* m = (float*)ptrs[2 * n_tensors + ti];
  float* v = (float*)ptrs[3 * n_tensors + ti];
  const float lr = hyper ? hyper[0] : lr_v, bc1 = hyper ? hyper[1] : bc1_v, bc2 = hyper ? hyper[2] : bc2_v,
              gs = hyper ? hyper[3] : gs_v;
  const float step_size = lr / bc1, bc2s = sqrtf(bc2);
  const int64_t end = (start + 4096 < n) ? start + 4096 : n;
  for (int64_t i = start + threadIdx.x; i < end; i += 256) {
    float gi = g[i] * gs, mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * omb1;                    // exp_avg.lerp_(grad, 1-beta1)
    vi = vi * beta2 + omb2 * gi * gi;              // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
    float denom = sqrtf(vi) / bc2s + eps;
    p[i] = p[i] - step_size * (mi / denom);        // param.addcdiv_(exp_avg, denom, -step_size)
    m[i] = mi; v[i] = vi;
    if (zero_grad) g[i] = 0.f;
  }
}

// ------------------------------------------------------------------------------- decode helpers
constexpr float MASK_FILL = -1e9f;

// One wave per (molecule, head): q (dk) against n_keys cached rows.  HBM-bound (the K/V rows are
// read once per step), so the point is to read them in whole cache lines and to read nothing that
// cannot matter:
//   scores: 8 lanes share a key row in whole 128-byte lines (8 keys per instruction), the partial dot
//           products meet in three DPP adds;
//   masked keys are not read at all: their score is the -1e9 fill, and unless EVERY key of the row
//           is masked exp(-1e9 - max) is exactly 0 in fp32, so their V rows are skipped too (an
//           all-masked row attends uniformly to all n_keys, as in the reference, and reads them);
//   P.V: dk/4 lanes cover one V row with float4 loads, 64/(dk/4) rows in flight per instruction.
template <int DK>
__global__ __launch_bounds__(256) void decode_attn_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ kc,
                                                          const float* __restrict__ vc, int ldc, int Lcache,
                                                          float* __restrict__ o, int ldo, const uint8_t* __restrict__ key_mask,
                                                          int ld_mask, int n_keys, int n, int h, float scale) {
  // scores: LPK lanes share a key row and every load instruction takes WHOLE 128-byte lines of it (dk = 64: 8 lanes x
  // 16 bytes, twice; dk = 32: 8 lanes once) -- 64-byte pieces of 16 rows per instruction kept the address path twice as
  // busy for the same bytes
  constexpr int LPK = 8;                  // lanes per key row in the score phase
  constexpr int NF4 = DK / (4 * LPK);     // float4 per lane and key (2 | 1)
  constexpr int KPI = 64 / LPK;           // keys per load instruction (8)
  constexpr int NG = 128 / KPI;           // key groups (16)
  constexpr int LPR = DK / 4;             // lanes per V row
  constexpr int KG = 64 / LPR;            // V rows per wave instruction
  __shared__ float ps[4][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = blockIdx.x * 4 + wave;
  const bool active = pair < n * h;
  const int b = active ? pair / h : 0, head = active ? pair % h : 0;
  const int part = lane & (LPK - 1), kl = lane / LPK;
  float4 qr[NF4];
#pragma unroll
  for (int it = 0; it < NF4; ++it)
    qr[it] = *reinterpret_cast<const float4*>(q + (size_t)b * ldq + head * DK + 32 * it + 4 * part);
  const float* kbase = kc + (size_t)b * Lcache * ldc + head * DK + 4 * part;
  const uint8_t* mrow = key_mask ? key_mask + (size_t)b * ld_mask : nullptr;
  float sc[NG];
  float mx = -INFINITY;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int key = KPI * g + kl;
    float s = -INFINITY;
    if (KPI * g < n_keys) {                         // wave-uniform
      const bool valid = key < n_keys;
      const bool m = valid && (mrow ? mrow[key] != 0 : true);
      float acc = 0.f;
      if (m) {
        const float* kr = kbase + (size_t)key * ldc;
#pragma unroll
        for (int it = 0; it < NF4; ++it) {
          float4 kv = *reinterpret_cast<const float4*>(kr + 32 * it);
          acc += kv.x * qr[it].x + kv.y * qr[it].y + kv.z * qr[it].z + kv.w * qr[it].w;
        }
      }
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      acc += __shfl_xor(acc, 4, 64);
      if (valid) s = m ? acc * scale : MASK_FILL;
    }
    sc[g] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float loc = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    sc[g] = expf(sc[g] - mx);                       // exp(-inf) = 0 for keys past n_keys
    loc += sc[g];
  }
  const float inv = 1.f / wave_sum(part == 0 ? loc : 0.f);
  if (part == 0) {
#pragma unroll
    for (int g = 0; g < NG; ++g) ps[wave][KPI * g + kl] = sc[g] * inv;
  }
  __syncthreads();
  const int c4 = lane % LPR, kg = lane / LPR;
  const float* vbase = vc + (size_t)b * Lcache * ldc + head * DK + 4 * c4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = kg; j < n_keys; j += KG) {
    const float p = ps[wave][j];
    if (p != 0.f) {
      const float4 v = *reinterpret_cast<const float4*>(vbase + (size_t)j * ldc);
      acc.x += p * v.x; acc.y += p * v.y; acc.z += p * v.z; acc.w += p * v.w;
    }
  }
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
    acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
  }
  if (active && kg == 0) *reinterpret_cast<float4*>(o + (size_t)b * ldo + head * DK + 4 * c4) = acc;
}

__global__ void argmax_kernel(const float* __restrict__ logits, int ldl, int n_cls, int64_t* ids, int ld_ids, int col,
                              int32_t* out32, int n, int plus) {
  int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  const float* lg = logits + (size_t)b * ldl;
  int best = 0;
  float bv = lg[0];
  for (int c = 1; c < n_cls; ++c)
    if (lg[c] > bv) { bv = lg[c]; best = c; }   // strict > keeps the FIRST maximum (torch.max rule)
  if (ids) ids[(size_t)b * ld_ids + col] = best + plus;
  if (out32) out32[b] = best + plus;
}

__global__ void scatter_rows_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int Ldst,
                                    int t, int n, int width) {
  const int w4 = width >> 2;
  size_t total = (size_t)n * w4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int b = (int)(i / w4), c = (int)(i % w4) * 4;
    *reinterpret_cast<float4*>(&dst[((size_t)b * Ldst + t) * ldd + c]) =
        *reinterpret_cast<const float4*>(&src[(size_t)b * lds + c]);
  }
}

__global__ void embed_step_kernel(const int64_t* __restrict__ ids, int ld_ids, int col, const float* __restrict__ lut,
                                  const float* __restrict__ pe, float* __restrict__ out, int n, int d, int vocab, float sq) {
  const int d4 = d >> 2;
  size_t total = (size_t)n * d4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int b = (int)(i / d4), c = (int)(i % d4) * 4;
    int64_t id = ids[(size_t)b * ld_ids + col];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    float4 w = *reinterpret_cast<const float4*>(&lut[(size_t)id * d + c]);
    float4 p = *reinterpret_cast<const float4*>(&pe[(size_t)col * d + c]);
    *reinterpret_cast<float4*>(&out[(size_t)b * d + c]) =
        make_float4(w.x * sq + p.x, w.y * sq + p.y, w.z * sq + p.z, w.w * sq + p.w);
  }
}

inline int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

// Both entry points below run the same two kernels; they differ only in where the 8 floats of
// partial sums per chunk live.
static int loss_launch(const int64_t* tokens, const float* logits, int ldl, const float* weights, const float* pred_len,
                       const float* mu, const float* logvar, const float* pred_prop, const float* true_prop,
                       const int64_t* true_len, float beta, int B, int L, int n_cls, int n_len, int d_latent, int pad_idx,
                       int n_groups, float* out5, float* dlogits, float* dpred_len, float* dmu, float* dlogvar,
                       float* dpred_prop, float* sums, hipStream_t s, const char* who) {
  TVAE_CHECK(tokens && logits && weights && mu && logvar && out5 && sums, TVAE_E_NULL, "%s: null pointer", who);
  // pred_len may be NULL (with n_len 0): the loss without the length-prediction term, vae_loss of the reference (loss.py:8-21)
  TVAE_CHECK((pred_len != nullptr) == (n_len > 0) && (pred_len || !dpred_len), TVAE_E_SHAPE, "%s: pred_len and n_len %d disagree", who, n_len);
  TVAE_CHECK(B > 0 && L > 0 && n_cls > 0 && n_cls <= ldl && n_len >= 0 && d_latent > 0, TVAE_E_SHAPE,
             "%s: bad shape B=%d L=%d n_cls=%d ldl=%d", who, B, L, n_cls, ldl);
  TVAE_CHECK(!pred_prop || true_prop, TVAE_E_NULL, "%s: pred_prop without true_prop", who);
  TVAE_CHECK((dmu == nullptr) == (dlogvar == nullptr), TVAE_E_NULL, "%s: dmu/dlogvar must come together", who);
  TVAE_CHECK(n_groups >= 1 && B % n_groups == 0, TVAE_E_SHAPE, "%s: B=%d not divisible into %d groups", who, B, n_groups);
  if (hipMemsetAsync(sums, 0, 8 * (size_t)n_groups * sizeof(float), s) != hipSuccess) {
    tvae_set_error("%s: memset failed", who);
    return TVAE_E_LAUNCH;
  }
  LossArgs a = {tokens, logits, ldl, weights, pred_len, mu, logvar, pred_prop, true_prop, true_len, beta,
                B, L, n_cls, n_len, d_latent, pad_idx, n_groups, out5, dlogits, dpred_len, dmu, dlogvar, dpred_prop, sums};
  dim3 grid(cdiv((long)(B / n_groups) * L, 256), n_groups);
  hipLaunchKernelGGL(loss_reduce_kernel, grid, dim3(256), 0, s, a);
  hipLaunchKernelGGL(loss_grad_kernel, grid, dim3(256), 0, s, a);
  TVAE_LAUNCH_CHECK(who);
  return TVAE_OK;
}

extern "C" int tvae_loss_fwd_bwd(const int64_t* tokens, const float* logits, int ldl, const float* weights,
                                 const float* pred_len, const float* mu, const float* logvar, const float* pred_prop,
                                 const float* true_prop, const int64_t* true_len, float beta, int B, int L, int n_cls, int n_len, int d_latent,
                                 int pad_idx, int n_groups, float* out5, float* dlogits, float* dpred_len, float* dmu, float* dlogvar,
                                 float* dpred_prop, float* scratch, void* stream) {
  return loss_launch(tokens, logits, ldl, weights, pred_len, mu, logvar, pred_prop, true_prop, true_len, beta, B, L, n_cls,
                     n_len, d_latent, pad_idx, n_groups, out5, dlogits, dpred_len, dmu, dlogvar, dpred_prop, scratch,
                     (hipStream_t)stream, "tvae_loss_fwd_bwd");
}

// ---- the same call for callers that cannot allocate (HIP graph capture, serving loops): the partial sums live in
// a device array that is part of the code object, TVAE_LOSS_STATIC_MAX_GROUPS chunks at most (32 KB per GPU, held
// until the process ends).  `scratch` is a host-side stand-in for that array, NOT a device address: its address on
// each GPU comes from hipGetSymbolAddress with that GPU current, once per device, and only that pointer is ever
// given to a memset or a kernel.
static __device__ __align__(256) float scratch[8 * TVAE_LOSS_STATIC_MAX_GROUPS];
static constexpr int MAX_DEVICES = 64;
static float* g_static_sums[MAX_DEVICES];          // per device: where `scratch` lives on it
static std::mutex g_static_sums_lock;

static int static_sums_for_current_device(float** out, const char* who) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
    tvae_set_error("%s: no current HIP device (or device index beyond %d)", who, MAX_DEVICES - 1);
    return TVAE_E_LAUNCH;
  }
  std::lock_guard<std::mutex> hold(g_static_sums_lock);
  if (!g_static_sums[dev]) {
    float* p = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) {
      tvae_set_error("%s: hipGetSymbolAddress(scratch) on device %d failed: %s", who, dev, hipGetErrorString(e));
      return TVAE_E_LAUNCH;
    }
    g_static_sums[dev] = p;
  }
  *out = g_static_sums[dev];
  return TVAE_OK;
}

extern "C" int tvae_loss_fwd_bwd_static(const int64_t* tokens, const float* logits, int ldl, const float* weights,
                                        const float* pred_len, const float* mu, const float* logvar, const float* pred_prop,
                                        const float* true_prop, const int64_t* true_len, float beta, int B, int L, int n_cls,
                                        int n_len, int d_latent, int pad_idx, int n_groups, float* out5, float* dlogits,
                                        float* dpred_len, float* dmu, float* dlogvar, float* dpred_prop, void* stream) {
  TVAE_CHECK(n_groups >= 1 && n_groups <= TVAE_LOSS_STATIC_MAX_GROUPS, TVAE_E_SHAPE,
             "tvae_loss_fwd_bwd_static: n_groups=%d outside 1..%d (the built-in workspace holds %d chunks; use "
             "tvae_loss_fwd_bwd with a caller-owned workspace beyond that)", n_groups, TVAE_LOSS_STATIC_MAX_GROUPS,
             TVAE_LOSS_STATIC_MAX_GROUPS);
  float* sums = nullptr;
  int rc = static_sums_for_current_device(&sums, "tvae_loss_fwd_bwd_static");
  if (rc != TVAE_OK) return rc;
  return loss_launch(tokens, logits, ldl, weights, pred_len, mu, logvar, pred_prop, true_prop, true_len, beta, B, L, n_cls,
                     n_len, d_latent, pad_idx, n_groups, out5, dlogits, dpred_len, dmu, dlogvar, dpred_prop, sums,
                     (hipStream_t)stream, "tvae_loss_fwd_bwd_static");
}

extern "C" int tvae_adam_multi(void* const* ptrs, const int64_t* sizes, const int32_t* chunk_tensor,
                               const int64_t* chunk_start, int n_tensors, int n_chunks, const float* hyper,
                               double lr, double bias_correction1, double bias_correction2, double grad_scale,
                               double beta1, double beta2, double eps, int zero_grad, void* stream) {
  TVAE_CHECK(ptrs && sizes && chunk_tensor && chunk_start, TVAE_E_NULL, "tvae_adam_multi: null pointer");
  TVAE_CHECK(hyper || (bias_correction1 > 0.0 && bias_correction2 > 0.0), TVAE_E_SHAPE,
             "tvae_adam_multi: bias corrections must be positive (step >= 1), got %g %g", bias_correction1, bias_correction2);
  TVAE_CHECK(n_tensors > 0 && n_chunks > 0, TVAE_E_SHAPE, "tvae_adam_multi: empty parameter list");
  float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
  hipLaunchKernelGGL(adam_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, ptrs, sizes, chunk_tensor,
                     chunk_start, n_tensors, hyper, (float)lr, (float)bias_correction1, (float)bias_correction2,
                     (float)grad_scale, (float)beta2, omb1, omb2, (float)eps, zero_grad);
  TVAE_LAUNCH_CHECK("tvae_adam_multi");
  return TVAE_OK;
}

extern "C" int tvae_decode_attn(const float* q, int ldq, const float* kc, const float* vc, int ldc, int Lcache, float* o,
                                int ldo, const uint8_t* key_mask, int ld_mask, int n_keys, int n, int h, int dk,
                                float scale, void* stream) {
  TVAE_CHECK(q && kc && vc && o, TVAE_E_NULL, "tvae_decode_attn: null pointer");
  TVAE_CHECK(n > 0 && h > 0 && n_keys >= 1 && n_keys <= 128 && n_keys <= Lcache && (dk == 32 || dk == 64) &&
                 ldc % 4 == 0 && ldq % 4 == 0 && ldo % 4 == 0 && aligned16(kc) && aligned16(vc) && aligned16(q) && aligned16(o),
             TVAE_E_SHAPE, "tvae_decode_attn: need 1<=n_keys<=128, dk in {32,64}, 16-byte aligned rows (n_keys=%d dk=%d)",
             n_keys, dk);
  dim3 grid(cdiv((long)n * h, 4));
  if (dk == 64)
    hipLaunchKernelGGL(decode_attn_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, q, ldq, kc, vc, ldc, Lcache, o,
                       ldo, key_mask, ld_mask, n_keys, n, h, scale);
  else
    hipLaunchKernelGGL(decode_attn_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, q, ldq, kc, vc, ldc, Lcache, o,
                       ldo, key_mask, ld_mask, n_keys, n, h, scale);
  TVAE_LAUNCH_CHECK("tvae_decode_attn");
  return TVAE_OK;
}

extern "C" int tvae_argmax_next(const float* logits, int ldl, int n_cls, int64_t* ids, int ld_ids, int col, int n,
                                void* stream) {
  TVAE_CHECK(logits && ids && n > 0 && n_cls > 0, TVAE_E_NULL, "tvae_argmax_next: bad arguments");
  hipLaunchKernelGGL(argmax_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, logits, ldl, n_cls, ids,
                     ld_ids, col, (int32_t*)nullptr, n, 1);
  TVAE_LAUNCH_CHECK("tvae_argmax_next");
  return TVAE_OK;
}
extern "C" int tvae_argmax_rows(const float* logits, int ldl, int n_cls, int32_t* out, int n, void* stream) {
  TVAE_CHECK(logits && out && n > 0 && n_cls > 0, TVAE_E_NULL, "tvae_argmax_rows: bad arguments");
  hipLaunchKernelGGL(argmax_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, logits, ldl, n_cls,
                     (int64_t*)nullptr, 0, 0, out, n, 0);
  TVAE_LAUNCH_CHECK("tvae_argmax_rows");
  return TVAE_OK;
}
extern "C" int tvae_scatter_rows(const float* src, int lds, float* dst, int ldd, int Ldst, int t, int n, int width,
                                 void* stream) {
  TVAE_CHECK(src && dst, TVAE_E_NULL, "tvae_scatter_rows: null pointer");
  TVAE_CHECK(n > 0 && width % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && t >= 0 && t < Ldst && aligned16(src) && aligned16(dst),
             TVAE_E_SHAPE, "tvae_scatter_rows: bad shape/alignment");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((size_t)n * width / 4)), dim3(256), 0, (hipStream_t)stream, src,
                     lds, dst, ldd, Ldst, t, n, width);
  TVAE_LAUNCH_CHECK("tvae_scatter_rows");
  return TVAE_OK;
}
extern "C" int tvae_embed_step(const int64_t* ids, int ld_ids, int col, const float* lut, const float* pe, float* out,
                               int n, int d, int vocab, void* stream) {
  TVAE_CHECK(ids && lut && pe && out, TVAE_E_NULL, "tvae_embed_step: null pointer");
  TVAE_CHECK(n > 0 && d % 4 == 0 && col >= 0 && col < ld_ids, TVAE_E_SHAPE, "tvae_embed_step: bad shape");
  hipLaunchKernelGGL(embed_step_kernel, dim3(grid_for((size_t)n * d / 4)), dim3(256), 0, (hipStream_t)stream, ids,
                     ld_ids, col, lut, pe, out, n, d, vocab, sqrtf((float)d));
  TVAE_LAUNCH_CHECK("tvae_embed_step");
  return TVAE_OK;
}

// ------------------------------------------------------------------------------- error string
static thread_local char g_err[512] = "";
void tvae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* tvae_last_error(void) { return g_err; }
extern "C" const char* tvae_version(void) { return "tvae-hip 0.1 (gfx950)"; }
