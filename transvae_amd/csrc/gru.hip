// GRU recurrence of the RNN / RNNAttn models (reference transvae/rnn_models.py:225,241,284,301,320,333,362,378: torch.nn.GRU)
// and the row softmax of RNNAttnEncoder's attention weights (rnn_models.py:246).
//
// A GRU layer is  gi_t = x_t W_ih^T + b_ih  (no recurrence: ONE large GEMM over all time steps, done by the caller with
// tvae_gemm) followed, per time step, by
//     gh_t = h_{t-1} W_hh^T + b_hh                      r = sigmoid(gi_r + gh_r)    z = sigmoid(gi_z + gh_z)
//     n = tanh(gi_n + r * gh_n)                          h_t = (1 - z) * n + z * h_{t-1}
// The recurrent part is what this file does: one launch per time step that fuses the [B, H] x [H, 3H] product with the gate
// arithmetic.  Batch rows are independent and the hidden units of one step are independent, so a step is a grid of
// 32-row x 16-unit tiles; the four waves of a workgroup split K (each lane reads whole 64-byte runs of its operand rows
// straight into MFMA operand registers -- v_mfma_f32_16x16x4_f32 takes any k order as long as A and B agree), add their
// partial tiles through LDS, and every thread then finishes one (row, unit).  The backward step is the same product shape
// against W_hh^T (K = 3H) fused with the gate derivatives; weight gradients are left to two large GEMMs over all steps.
#include "common.h"

namespace {

struct GruArgs {
  const float* gi;      // [B, T, 3H]  input projections (+ b_ih), row (b, t) at (b * sb + t * st) * 3H
  const float* w;       // fwd: W_hh [3H, H]      bwd: W_hh^T [H, 3H]
  const float* b_hh;    // [3H]
  const float* h0;      // [B, H] or null (zeros)
  float* hseq;          // [B, T, H]   h_t
  float* gates;         // [B, T, 4H]  r, z, n, gh_n + b_hn   (null: inference)
  const float* dhseq;   // bwd: [B, T, H] gradient arriving at every h_t (null: none)
  const float* dh_last; // bwd: [B, H] extra gradient of the final state (null: none)
  float* dgi;           // bwd: [B, T, 3H] gradient of gi  (dr_pre, dz_pre, dn_pre)
  float* dgh;           // bwd: [B, T, 3H] gradient of gh  (dr_pre, dz_pre, dn_pre * r)
  float* carry;         // bwd: [B, H] dh_{t+1} * z_{t+1}
  int T, B, H, sb, st, t;
  int continues;        // bwd: the step after the last one of this call exists (its dgh rows and the carry are valid)
};

__device__ inline float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

constexpr int RS = 2;                    // 16-row sub-tiles per workgroup: a step's tile is 32 rows x 16 units

// partial[s][g][r] of this wave: rows 16 s + 4 * (lane / 16) + r, unit lane % 16, over the wave's quarter of K.
// A rows are `lda` floats apart, rows of W `ldw`; NG accumulators per sub-tile use W rows u0 + g * gstride + j.
// A step is a chain of latencies around 1.25 us of MFMA time, so every operand byte of a batch of CH float4 per row is
// requested before the first MFMA of the batch is issued (one memory round trip per batch, not one per float4), and a
// workgroup covers 32 rows with one read of its W rows.  CH divides the number of float4 per lane (the host picks it):
// with a range test per chunk the compiler peeled the loop into load -> wait -> MFMA links, one round trip each; the
// scheduling barrier keeps it from sinking loads below MFMAs to save registers (a step runs one wave per SIMD anyway).
template <int NG, int CH>
__device__ inline void tile_product(const float* __restrict__ A, int lda, int rows_valid, const float* __restrict__ W, int ldw,
                                    int gstride, int K, int wave, int lane, floatx4 (&acc)[RS][NG]) {
  const int i = lane & 15, kq = lane >> 4;
  const int kw = K >> 2;                 // this wave's share of K
  const int kl = kw >> 2;                // floats per lane: a contiguous run
  const int nch = kl >> 2;               // float4 per lane
  const float* ap[RS];
#pragma unroll
  // k order: float4 c of lane-quarter kq is k = 16 c + 4 kq .. + 3 of the wave's share, so that ONE load instruction reads a
  // contiguous 64-byte run per row (16 runs per instruction; with each lane walking its own run it touched 32-64 cache lines
  // per instruction, 16 bytes from each, and the texture path's line rate bounded the step)
  for (int s = 0; s < RS; ++s) ap[s] = A + (16 * s + i < rows_valid ? 16 * s + i : 0) * lda + wave * kw + 4 * kq;
  const float* wp = W + i * ldw + wave * kw + 4 * kq;
  for (int c0 = 0; c0 < nch; c0 += CH) {
    float4 a[RS][CH], b[NG][CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int s = 0; s < RS; ++s) a[s][c] = *reinterpret_cast<const float4*>(ap[s] + 16 * (c0 + c));
#pragma unroll
      for (int g = 0; g < NG; ++g) b[g][c] = *reinterpret_cast<const float4*>(wp + g * gstride * ldw + 16 * (c0 + c));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int s = 0; s < RS; ++s)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          acc[s][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][c].x, b[g][c].x, acc[s][g], 0, 0, 0);
          acc[s][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][c].y, b[g][c].y, acc[s][g], 0, 0, 0);
          acc[s][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][c].z, b[g][c].z, acc[s][g], 0, 0, 0);
          acc[s][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][c].w, b[g][c].w, acc[s][g], 0, 0, 0);
        }
    }
  }
}

// sum over the four waves of accumulator (s, g) for thread tid's rows 16 s + tid / 16, unit tid % 16
template <int NG>
__device__ inline void reduce_tiles(floatx4 (&acc)[RS][NG], float* red, int wave, int lane, int tid, float (&out)[RS][NG]) {
#pragma unroll
  for (int s = 0; s < RS; ++s)
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(((wave * RS + s) * NG + g) * 4 + r) * 64 + lane] = acc[s][g][r];
  __syncthreads();
  const int i = tid >> 4, j = tid & 15;
  const int src = 16 * (i >> 2) + j, r = i & 3;
#pragma unroll
  for (int s = 0; s < RS; ++s)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += red[(((w * RS + s) * NG + g) * 4 + r) * 64 + src];
      out[s][g] = t;
    }
}

// Per-thread addresses are a uniform 64-bit base (scalar arithmetic) plus a 32-bit offset inside the workgroup's 32 rows
// (the host checks that 32 rows of the widest buffer stay below 2^31 floats): with 64-bit per-thread products the register
// allocator paired the high word of an address with a register that a pending load was to fill, and the address
// arithmetic of the NEXT load waited for that load -- a memory round trip in the middle of the prologue.
template <int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gru_fwd_step(GruArgs a) {
  __shared__ float red[4 * RS * 3 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * 16 * RS, u0 = blockIdx.y * 16;
  const int H = a.H, t = a.t;
  const float* hprev_base = t > 0 ? a.hseq + ((long)b0 * a.sb + (long)(t - 1) * a.st) * H : (a.h0 ? a.h0 + (long)b0 * H : nullptr);
  const int hprev_ld = t > 0 ? a.sb * H : H;
  const long row0 = (long)b0 * a.sb + (long)t * a.st;          // buffer row of the workgroup's first batch row
  // this thread's (rows, unit) and everything its gate arithmetic reads: requested before the product, used after it
  const int i = tid >> 4, j = tid & 15, u = u0 + j;
  const float br = a.b_hh[u], bz = a.b_hh[H + u], bn = a.b_hh[2 * H + u];
  const float* gi0 = a.gi + row0 * 3 * H;
  const float* hp0 = hprev_base ? hprev_base : a.gi;           // no initial state: any readable address, the value is dropped
  bool live[RS];
  int rel[RS];                                                 // row of this thread relative to row0, in buffer rows
  float gir[RS], giz[RS], gin[RS], hp[RS];
#pragma unroll
  for (int s = 0; s < RS; ++s) {
    live[s] = b0 + 16 * s + i < a.B;
    const int lr = live[s] ? 16 * s + i : 0;
    rel[s] = lr * a.sb;
    const float* gi = gi0 + rel[s] * 3 * H;
    gir[s] = gi[u]; giz[s] = gi[H + u]; gin[s] = gi[2 * H + u];
    hp[s] = hp0[lr * hprev_ld + u];
  }
  floatx4 acc[RS][3];
#pragma unroll
  for (int s = 0; s < RS; ++s)
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[s][g] = floatx4{0.f, 0.f, 0.f, 0.f};
  if (hprev_base) tile_product<3, CH>(hprev_base, hprev_ld, a.B - b0, a.w + (long)u0 * H, H, H, H, wave, lane, acc);
  float gh[RS][3];
  reduce_tiles<3>(acc, red, wave, lane, tid, gh);
  float* hs0 = a.hseq + row0 * H;
  float* G0 = a.gates ? a.gates + row0 * 4 * H : nullptr;
#pragma unroll
  for (int s = 0; s < RS; ++s) {
    if (!live[s]) continue;
    const float hr = gh[s][0] + br, hz = gh[s][1] + bz, hn = gh[s][2] + bn;
    const float r = sigmoidf_(gir[s] + hr), z = sigmoidf_(giz[s] + hz);
    const float n = tanhf(gin[s] + r * hn);
    hs0[rel[s] * H + u] = (1.f - z) * n + z * (hprev_base ? hp[s] : 0.f);
    if (G0) {
      float* G = G0 + rel[s] * 4 * H;
      G[u] = r; G[H + u] = z; G[2 * H + u] = n; G[3 * H + u] = hn;
    }
  }
}

template <int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gru_bwd_step(GruArgs a) {
  __shared__ float red[4 * RS * 1 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * 16 * RS, u0 = blockIdx.y * 16;
  const int H = a.H, t = a.t;
  const bool last = t == a.T - 1 && !a.continues;
  const int i = tid >> 4, j = tid & 15, u = u0 + j;
  const long row0 = (long)b0 * a.sb + (long)t * a.st;
  // the gate derivatives' inputs: requested before the product, every load unconditional (an operand that is absent is read
  // from some readable address and dropped by a select: no branch whose join would wait for the loads before it)
  const float* G0 = a.gates + row0 * 4 * H;
  const float* dhs0 = a.dhseq ? a.dhseq + row0 * H : G0;
  const float* add0 = last ? (a.dh_last ? a.dh_last + (long)b0 * H : G0) : a.carry + (long)b0 * H;   // [B, H] rows
  const bool has_add = !last || a.dh_last;
  const float* hp0 = t > 0 ? a.hseq + (row0 - a.st) * H : (a.h0 ? a.h0 + (long)b0 * H : G0);
  const int hp_ld = t > 0 ? a.sb * H : H;
  const bool has_hp = t > 0 || a.h0;
  bool live[RS];
  int rel[RS], lr[RS];
  float dh[RS], r[RS], z[RS], n[RS], hn[RS], hp[RS];
#pragma unroll
  for (int s = 0; s < RS; ++s) {
    live[s] = b0 + 16 * s + i < a.B;
    lr[s] = live[s] ? 16 * s + i : 0;
    rel[s] = lr[s] * a.sb;
    const float d1 = dhs0[rel[s] * H + u], d2 = add0[lr[s] * H + u];
    dh[s] = (a.dhseq ? d1 : 0.f) + (has_add ? d2 : 0.f);
    const float* G = G0 + rel[s] * 4 * H;
    r[s] = G[u]; z[s] = G[H + u]; n[s] = G[2 * H + u]; hn[s] = G[3 * H + u];
    const float h1 = hp0[lr[s] * hp_ld + u];
    hp[s] = has_hp ? h1 : 0.f;
  }
  floatx4 acc[RS][1];
#pragma unroll
  for (int s = 0; s < RS; ++s) acc[s][0] = floatx4{0.f, 0.f, 0.f, 0.f};
  if (!last) {
    const float* dgh_next = a.dgh + ((long)b0 * a.sb + (long)(t + 1) * a.st) * 3 * H;
    tile_product<1, CH>(dgh_next, a.sb * 3 * H, a.B - b0, a.w + (long)u0 * 3 * H, 3 * H, 0, 3 * H, wave, lane, acc);
  }
  float back[RS][1];
  reduce_tiles<1>(acc, red, wave, lane, tid, back);
  float* dgi0 = a.dgi + row0 * 3 * H;
  float* dgh0 = a.dgh + row0 * 3 * H;
  float* carry0 = a.carry + (long)b0 * H;
#pragma unroll
  for (int s = 0; s < RS; ++s) {
    if (!live[s]) continue;
    const float d = last ? dh[s] : dh[s] + back[s][0];
    const float dn = d * (1.f - z[s]), dz = d * (hp[s] - n[s]);
    const float dnp = dn * (1.f - n[s] * n[s]);
    const float drp = dnp * hn[s] * r[s] * (1.f - r[s]), dzp = dz * z[s] * (1.f - z[s]);
    float* dgi = dgi0 + rel[s] * 3 * H;
    float* dgh = dgh0 + rel[s] * 3 * H;
    dgi[u] = drp; dgi[H + u] = dzp; dgi[2 * H + u] = dnp;
    dgh[u] = drp; dgh[H + u] = dzp; dgh[2 * H + u] = dnp * r[s];
    carry0[lr[s] * H + u] = d * z[s];
  }
}

// one wave per row: y = softmax(x) / dx = y * (dy - sum(y dy))
__global__ __launch_bounds__(256) void softmax_rows_fwd(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int rows, int n) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  float mx = -INFINITY;
  for (int c = lane; c < n; c += 64) mx = fmaxf(mx, xr[c]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < n; c += 64) se += expf(xr[c] - mx);
  se = wave_sum(se);
  const float inv = 1.f / se;
  for (int c = lane; c < n; c += 64) y[(long)row * ldy + c] = expf(xr[c] - mx) * inv;
}

__global__ __launch_bounds__(256) void softmax_rows_bwd(const float* __restrict__ y, int ldy, const float* __restrict__ dy, int lddy,
                                                        float* __restrict__ dx, int lddx, int rows, int n) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* yr = y + (long)row * ldy;
  const float* dr = dy + (long)row * lddy;
  float s = 0.f;
  for (int c = lane; c < n; c += 64) s += yr[c] * dr[c];
  s = wave_sum(s);
  for (int c = lane; c < n; c += 64) dx[(long)row * lddx + c] = yr[c] * (dr[c] - s);
}

int check_gru(const char* who, int T, int B, int H, int time_major) {
  TVAE_CHECK(T > 0 && B > 0 && H > 0, TVAE_E_SHAPE, "%s: T, B, H must be positive (%d, %d, %d)", who, T, B, H);
  TVAE_CHECK(H % 64 == 0, TVAE_E_SHAPE, "%s: hidden size %d must be a multiple of 64", who, H);
  TVAE_CHECK(time_major == 0 || time_major == 1, TVAE_E_SHAPE, "%s: time_major must be 0 or 1", who);
  TVAE_CHECK((long)B * T * 4 * H < (1L << 40), TVAE_E_SHAPE, "%s: sequence buffer too large", who);
  // 32-bit offsets inside a workgroup's 32 rows of the widest buffer ([.., 4H], rows T apart when batch-major), and in W_hh
  TVAE_CHECK(32L * T * 4 * H < (1L << 31) && 3L * H * H < (1L << 31), TVAE_E_SHAPE, "%s: T * H = %ld exceeds the kernels' 32-bit row offsets", who, (long)T * H);
  return TVAE_OK;
}

}  // namespace

extern "C" int tvae_gru_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h0, float* hseq, float* gates,
                            int T, int B, int H, int time_major, void* stream) {
  TVAE_CHECK(gi && w_hh && b_hh && hseq, TVAE_E_NULL, "tvae_gru_fwd: null pointer");
  if (int e = check_gru("tvae_gru_fwd", T, B, H, time_major)) return e;
  TVAE_CHECK(aligned16(w_hh) && aligned16(hseq) && (!h0 || aligned16(h0)), TVAE_E_SHAPE, "tvae_gru_fwd: operands must be 16-byte aligned");
  GruArgs a = {};
  a.gi = gi; a.w = w_hh; a.b_hh = b_hh; a.h0 = h0; a.hseq = hseq; a.gates = gates;
  a.T = T; a.B = B; a.H = H; a.sb = time_major ? 1 : T; a.st = time_major ? B : 1;
  const dim3 grid(cdiv(B, 16 * RS), H / 16);
  const int nch = H / 64;              // float4 per lane and wave in the h W_hh^T product: chunks in flight = its largest divisor <= 4
  auto kern = nch % 4 == 0 ? gru_fwd_step<4> : nch % 3 == 0 ? gru_fwd_step<3> : nch % 2 == 0 ? gru_fwd_step<2> : gru_fwd_step<1>;
  for (int t = 0; t < T; ++t) {
    a.t = t;
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  TVAE_LAUNCH_CHECK("tvae_gru_fwd");
  return TVAE_OK;
}

extern "C" int tvae_gru_bwd(const float* dhseq, const float* dh_last, const float* hseq, const float* gates, const float* h0,
                            const float* w_hh_t, float* dgi, float* dgh, float* carry, int T, int B, int H, int time_major,
                            int continues, void* stream) {
  TVAE_CHECK(hseq && gates && w_hh_t && dgi && dgh && carry, TVAE_E_NULL, "tvae_gru_bwd: null pointer");
  TVAE_CHECK(dhseq || dh_last || continues, TVAE_E_NULL, "tvae_gru_bwd: neither dhseq nor dh_last given");
  TVAE_CHECK(!continues || (time_major && !dh_last), TVAE_E_SHAPE,
             "tvae_gru_bwd: a call that continues a later one needs time-major buffers (the next step's rows follow this call's) and no dh_last");
  if (int e = check_gru("tvae_gru_bwd", T, B, H, time_major)) return e;
  TVAE_CHECK(aligned16(w_hh_t) && aligned16(dgh), TVAE_E_SHAPE, "tvae_gru_bwd: operands must be 16-byte aligned");
  GruArgs a = {};
  a.w = w_hh_t; a.h0 = h0; a.hseq = const_cast<float*>(hseq); a.gates = const_cast<float*>(gates);
  a.dhseq = dhseq; a.dh_last = dh_last; a.dgi = dgi; a.dgh = dgh; a.carry = carry;
  a.T = T; a.B = B; a.H = H; a.sb = time_major ? 1 : T; a.st = time_major ? B : 1;
  a.continues = continues;
  const dim3 grid(cdiv(B, 16 * RS), H / 16);
  const int nch = 3 * H / 64;          // float4 per lane and wave in the dgh W_hh product (a multiple of 3)
  auto kern = nch % 12 == 0 ? gru_bwd_step<12> : nch % 6 == 0 ? gru_bwd_step<6> : gru_bwd_step<3>;
  for (int t = T - 1; t >= 0; --t) {
    a.t = t;
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  TVAE_LAUNCH_CHECK("tvae_gru_bwd");
  return TVAE_OK;
}

extern "C" int tvae_softmax_rows_fwd(const float* x, int ldx, float* y, int ldy, int rows, int n, void* stream) {
  TVAE_CHECK(x && y, TVAE_E_NULL, "tvae_softmax_rows_fwd: null pointer");
  TVAE_CHECK(rows > 0 && n > 0 && ldx >= n && ldy >= n, TVAE_E_SHAPE, "tvae_softmax_rows_fwd: bad shape (rows %d n %d ldx %d ldy %d)", rows, n, ldx, ldy);
  hipLaunchKernelGGL(softmax_rows_fwd, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, n);
  TVAE_LAUNCH_CHECK("tvae_softmax_rows_fwd");
  return TVAE_OK;
}

extern "C" int tvae_softmax_rows_bwd(const float* y, int ldy, const float* dy, int lddy, float* dx, int lddx, int rows, int n,
                                     void* stream) {
  TVAE_CHECK(y && dy && dx, TVAE_E_NULL, "tvae_softmax_rows_bwd: null pointer");
  TVAE_CHECK(rows > 0 && n > 0 && ldy >= n && lddy >= n && lddx >= n, TVAE_E_SHAPE, "tvae_softmax_rows_bwd: bad shape");
  hipLaunchKernelGGL(softmax_rows_bwd, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, y, ldy, dy, lddy, dx, lddx, rows, n);
  TVAE_LAUNCH_CHECK("tvae_softmax_rows_bwd");
  return TVAE_OK;
}
