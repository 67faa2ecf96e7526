// HBM-bound row kernels: LayerNorm fwd/bwd, column sums, embedding + positional encoding,
// conv/deconv bottleneck helpers, reparameterisation.  One wave (64 lanes) per row wherever a
// row is a feature vector; all loads are 8/16 bytes per lane and coalesced.
#include "common.h"

namespace {

// contiguous per-lane run of NPL floats, moved with the widest aligned vector the run allows
template <int NPL>
__device__ inline void load_run(float (&v)[NPL], const float* __restrict__ p) {
  if constexpr (NPL % 4 == 0) {
#pragma unroll
    for (int i = 0; i < NPL; i += 4) {
      float4 t = *reinterpret_cast<const float4*>(p + i);
      v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
    }
  } else if constexpr (NPL % 2 == 0) {
#pragma unroll
    for (int i = 0; i < NPL; i += 2) {
      float2 t = *reinterpret_cast<const float2*>(p + i);
      v[i] = t.x; v[i + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NPL; ++i) v[i] = p[i];
  }
}
template <int NPL>
__device__ inline void store_run(float* __restrict__ p, const float (&v)[NPL]) {
  if constexpr (NPL % 4 == 0) {
#pragma unroll
    for (int i = 0; i < NPL; i += 4) *reinterpret_cast<float4*>(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  } else if constexpr (NPL % 2 == 0) {
#pragma unroll
    for (int i = 0; i < NPL; i += 2) *reinterpret_cast<float2*>(p + i) = make_float2(v[i], v[i + 1]);
  } else {
#pragma unroll
    for (int i = 0; i < NPL; ++i) p[i] = v[i];
  }
}

// ---------------------------------------------------------------------------- LayerNorm
// Reference semantics (transvae/trans_models.py:1084-1087): mean over the last dim, UNBIASED
// std (divisor d-1), eps added to std:  y = a * (x - mean) / (std + eps) + b.
template <int NPL>  // elements per lane, d = 64 * NPL, lane owns a contiguous run
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                     const float* __restrict__ b, float* __restrict__ y,
                                                     int rows, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  constexpr int d = 64 * NPL;
  float v[NPL], av[NPL], bv[NPL];
  load_run<NPL>(v, x + (size_t)row * d + lane * NPL);
  load_run<NPL>(av, a + lane * NPL);
  load_run<NPL>(bv, b + lane * NPL);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) s += v[i];
  const float mean = wave_sum(s) * (1.f / d);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) { v[i] -= mean; ss += v[i] * v[i]; }
  const float sd = sqrtf(wave_sum(ss) * (1.f / (d - 1)));
  const float rinv = 1.f / (sd + eps);
#pragma unroll
  for (int i = 0; i < NPL; ++i) v[i] = av[i] * v[i] * rinv + bv[i];
  store_run<NPL>(y + (size_t)row * d + lane * NPL, v);
}

// dx_i = r*(g_i - mean(g)) - xc_i * r^2 * sum_j(g_j xc_j) / ((d-1) sd),  g = dy*a, r = 1/(sd+eps)
// da += dy * xc * r ; db += dy.  Each workgroup (up to 16 waves) walks rows_per_wg rows, keeps da/db
// partials in registers, reduces its waves through LDS and issues one atomic per column.
template <int NPL>
__global__ __launch_bounds__(512) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                     const float* __restrict__ dy, const float* __restrict__ add_to,
                                                     float* __restrict__ dx, float* __restrict__ da,
                                                     float* __restrict__ db, int rows, int rows_per_wg, float eps,
                                                     float* __restrict__ dxm, const uint32_t* __restrict__ rng,
                                                     uint32_t site, float p_drop) {
  constexpr int d = 64 * NPL;
  __shared__ float red[2][8][d];
  // dxm (optional): dx times the keep-scale of a dropout site indexed (row, column) with pitch d -- the residual dropout of
  // the sublayer whose OUTPUT this LayerNorm read.  dx is that sublayer's incoming gradient, and its two gradient GEMMs
  // need it masked: written here it costs one more store stream; regenerating the mask while staging costs both GEMMs
  // a hash per float4 inside their k-loops (26 us on the d x d weight gradient, 28 us on the FFN's).
  const DropCfg dcm = drop_cfg(dxm ? rng : nullptr, site, p_drop);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float av[NPL], pa[NPL], pb[NPL];
  load_run<NPL>(av, a + lane * NPL);
#pragma unroll
  for (int i = 0; i < NPL; ++i) { pa[i] = 0.f; pb[i] = 0.f; }
  const int rbeg = blockIdx.x * rows_per_wg;
  const int rend = min(rows, rbeg + rows_per_wg);
  // two rows per trip: all six loads are in flight before the first reduction, which is what hides the HBM
  // latency of this three-stream kernel at 8 waves per SIMD
  for (int row0 = rbeg + wave; row0 < rend; row0 += 2 * nw) {
    const bool two = row0 + nw < rend;
    const size_t off[2] = {(size_t)row0 * d + lane * NPL, (size_t)(two ? row0 + nw : row0) * d + lane * NPL};
    float v[2][NPL], g[2][NPL], res[2][NPL];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      load_run<NPL>(v[r], x + off[r]);
      load_run<NPL>(g[r], dy + off[r]);
      if (add_to) load_run<NPL>(res[r], add_to + off[r]);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (r == 1 && !two) break;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NPL; ++i) s += v[r][i];
      const float mean = wave_sum(s) * (1.f / d);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < NPL; ++i) { v[r][i] -= mean; ss += v[r][i] * v[r][i]; }
      const float sd = sqrtf(wave_sum(ss) * (1.f / (d - 1)));
      const float rinv = 1.f / (sd + eps);
      float sg = 0.f, sgx = 0.f;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        pb[i] += g[r][i];
        pa[i] += g[r][i] * v[r][i] * rinv;
        g[r][i] *= av[i];
        sg += g[r][i];
        sgx += g[r][i] * v[r][i];
      }
      sg = wave_sum(sg) * (1.f / d);
      sgx = wave_sum(sgx);
      const float c2 = rinv * rinv * sgx / ((d - 1) * sd);
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        float o = rinv * (g[r][i] - sg) - v[r][i] * c2;
        if (add_to) o += res[r][i];
        g[r][i] = o;
      }
      store_run<NPL>(dx + off[r], g[r]);
      if (dxm) {
        const uint32_t e0 = (uint32_t)(off[r]);            // element index = row * d + column (host: rows * d < 2^32)
        if constexpr (NPL % 4 == 0) {
#pragma unroll
          for (int i = 0; i < NPL; i += 4) {
            float ks[4];
            drop_scale4(dcm, e0 + i, ks);
            g[r][i] *= ks[0]; g[r][i + 1] *= ks[1]; g[r][i + 2] *= ks[2]; g[r][i + 3] *= ks[3];
          }
        } else {
#pragma unroll
          for (int i = 0; i < NPL; ++i) g[r][i] *= drop_scale(dcm, e0 + i);
        }
        store_run<NPL>(dxm + off[r], g[r]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) { red[0][wave][lane * NPL + i] = pa[i]; red[1][wave][lane * NPL + i] = pb[i]; }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float sa = 0.f, sb = 0.f;
    for (int w = 0; w < nw; ++w) { sa += red[0][w][c]; sb += red[1][w][c]; }
    atomicAdd(&da[c], sa);
    atomicAdd(&db[c], sb);
  }
}

// ---------------------------------------------------------------------------- column sums
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int ld, float* __restrict__ out,
                                                     int M, int N, int rows_per_wg, const uint32_t* rng,
                                                     uint32_t site, float p) {
  // blockDim (64, 4): x = column within a 64-wide strip, y = row phase
  __shared__ float red[4][64];
  DropCfg dc = drop_cfg(rng, site, p);
  const int col = blockIdx.x * 64 + threadIdx.x;
  const int rbeg = blockIdx.y * rows_per_wg, rend = min(M, rbeg + rows_per_wg);
  float s = 0.f;
  if (col < N)
    for (int r = rbeg + threadIdx.y; r < rend; r += 4)
      s += X[(size_t)r * ld + col] * drop_scale(dc, (uint32_t)r * (uint32_t)N + (uint32_t)col);
  red[threadIdx.y][threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.y == 0 && col < N)
    atomicAdd(&out[col], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------- embedding + PE
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ lut,
                                                        const float* __restrict__ pe, float* __restrict__ out,
                                                        int rows, int L, int d, int vocab, float sq,
                                                        const uint32_t* rng, uint32_t site, float p) {
  DropCfg dc = drop_cfg(rng, site, p);
  const int d4 = d >> 2;
  const size_t total = (size_t)rows * d4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int row = (int)(i / d4), c = (int)(i % d4) * 4;
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    int t = row % L;
    float4 w = *reinterpret_cast<const float4*>(&lut[(size_t)id * d + c]);
    float4 q = *reinterpret_cast<const float4*>(&pe[(size_t)t * d + c]);
    uint32_t base = (uint32_t)row * (uint32_t)d + (uint32_t)c;
    float ks[4];
    drop_scale4(dc, base, ks);
    float4 o;
    o.x = (w.x * sq + q.x) * ks[0];
    o.y = (w.y * sq + q.y) * ks[1];
    o.z = (w.z * sq + q.z) * ks[2];
    o.w = (w.w * sq + q.w) * ks[3];
    *reinterpret_cast<float4*>(&out[(size_t)row * d + c]) = o;
  }
}

// dlut[id] += sqrt(d) * keep * dout: every workgroup reduces its slab of rows into an LDS copy
// of the (vocab x d) table first, then adds the table to global memory once.
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                        float* __restrict__ dlut, int rows, int rows_per_wg, int d,
                                                        int vocab, float sq, const uint32_t* rng, uint32_t site, float p) {
  extern __shared__ float tab[];   // vocab * d
  DropCfg dc = drop_cfg(rng, site, p);
  for (int i = threadIdx.x; i < vocab * d; i += 256) tab[i] = 0.f;
  __syncthreads();
  const int rbeg = blockIdx.x * rows_per_wg, rend = min(rows, rbeg + rows_per_wg);
  // one thread per column: no intra-workgroup race.  Eight rows per trip: their id and gradient loads are all
  // issued before the first LDS add (a row at a time, every row paid a full HBM round trip: 73 us per launch)
  for (int c = threadIdx.x; c < d; c += 256) {
    for (int row0 = rbeg; row0 < rend; row0 += 8) {
      int64_t id[8];
      float g[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = min(row0 + u, rend - 1);
        id[u] = ids[row];
        g[u] = dout[(size_t)row * d + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = row0 + u;
        if (row < rend) {
          const int64_t t = id[u] < 0 ? 0 : (id[u] >= vocab ? vocab - 1 : id[u]);
          tab[t * d + c] += g[u] * drop_scale(dc, (uint32_t)row * (uint32_t)d + (uint32_t)c) * sq;
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < vocab * d; i += 256) {
    float v = tab[i];
    if (v != 0.f) atomicAdd(&dlut[i], v);
  }
}

// ---------------------------------------------------------------------------- conv helpers
// (Cout, Cin, k) -> (Cout, k*Cin) with wp[co][j*Cin + ci] = w[co][ci][j]          (Conv1d)
// transposed_flip: w is (Cin, Cout, k) (ConvTranspose1d) and the taps are flipped:
//                  wp[co][j*Cin + ci] = w[ci][co][k-1-j]
__global__ void conv_w_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int k,
                                   int tf, int unpack_grad) {
  const int total = Cout * Cin * k;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int co = i / (k * Cin), rem = i % (k * Cin), j = rem / Cin, ci = rem % Cin;
    size_t src = tf ? ((size_t)ci * Cout + co) * k + (k - 1 - j) : ((size_t)co * Cin + ci) * k + j;
    if (unpack_grad == 2) wp[src] = w[(size_t)rem * Cout + co];   // dwp arrives TRANSPOSED, (k*Cin, Cout)
    else if (unpack_grad) wp[src] = w[i];   // here `w` is dwp (packed layout) and `wp` is dw
    else wp[i] = w[src];
  }
}

// ---- ConvTranspose1d as a POLYPHASE product (no zero-stuffed input) ----------------------------------------------
// out[b, co, t] = sum_{ci, i, j : t + pad = i s + j} x[b, ci, i] w[ci, co, j].  With t + pad = m s + r (r < s) only the
// taps j = r + s q (q < Q = ceil(k / s)) reach phase r, from inputs i = m - q: per phase a Q-tap convolution of the
// UN-upsampled input, and all s phases together ONE product
//   C[(b, m), (r, co)] = sum_{qq < Q, ci} xp[(b, m + qq), ci] . wp[(r, co), (qq, ci)]
// over the input padded with Q - 1 leading zero rows per molecule (window row qq holds input i = m - (Q - 1 - qq)), whose
// output rows (b, m) x (r, co) ARE the token-major rows (b, m s + r) x co.  FLOPs = those of the transposed convolution
// itself (x s / k less than the zero-stuffed form).  Modes:
//   0  wp (s Cout, Q Cin):   wp[(r, co), (qq, ci)] = w[ci, co, r + s (Q - 1 - qq)]   (0 where that tap is >= k); bias tiled s times
//   1  wd (Cin, Q s Cout):   wd[ci, (q, r, co)]    = w[ci, co, r + s q]              the data-gradient product's weight
//   2  dw (Cin, Cout, k) from dwp laid out as mode 0, db[co] = sum_r dbp[(r, co)]
__global__ void deconv_poly_pack_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ bin,
                                        float* __restrict__ bout, int Cin, int Cout, int k, int s, int Q, int mode) {
  const int gtid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
  if (mode == 2) {
    const int total = Cin * Cout * k;
    for (int i = gtid; i < total; i += gsz) {
      const int j = i % k, co = (i / k) % Cout, ci = i / (k * Cout);
      const int r = j % s, qq = Q - 1 - j / s;
      out[i] = in[((size_t)r * Cout + co) * (Q * Cin) + qq * Cin + ci];
    }
    if (bin && bout)
      for (int co = gtid; co < Cout; co += gsz) {
        float t = 0.f;
        for (int r = 0; r < s; ++r) t += bin[r * Cout + co];
        bout[co] = t;
      }
    return;
  }
  const int total = s * Cout * Q * Cin;
  for (int i = gtid; i < total; i += gsz) {
    int ci, co, j;
    if (mode == 0) {
      ci = i % Cin;
      const int qq = (i / Cin) % Q, rc = i / (Cin * Q);
      co = rc % Cout;
      j = rc / Cout + s * (Q - 1 - qq);
    } else {
      co = i % Cout;
      const int qr = (i / Cout) % (Q * s);        // q s + r = the tap itself
      ci = i / (Cout * Q * s);
      j = qr;
    }
    out[i] = j < k ? in[((size_t)ci * Cout + co) * k + j] : 0.f;
  }
  if (mode == 0 && bin && bout)
    for (int i = gtid; i < s * Cout; i += gsz) bout[i] = bin[i % Cout];
}

__global__ __launch_bounds__(256) void pool_relu_fwd_kernel(const float* __restrict__ conv, float* __restrict__ y,
                                                            int B, int Lin, int Lp, int C, int out_cm) {
  const int c4n = C >> 2;
  const size_t total = (size_t)B * Lp * c4n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int c = (int)(i % c4n) * 4;
    size_t bt = i / c4n;
    int t = (int)(bt % Lp), b = (int)(bt / Lp);
    const float* p0 = conv + ((size_t)b * Lin + 2 * t) * C + c;
    float4 u = *reinterpret_cast<const float4*>(p0);
    float4 v = *reinterpret_cast<const float4*>(p0 + C);
    float4 o = make_float4(fmaxf(fmaxf(u.x, v.x), 0.f), fmaxf(fmaxf(u.y, v.y), 0.f),
                           fmaxf(fmaxf(u.z, v.z), 0.f), fmaxf(fmaxf(u.w, v.w), 0.f));
    if (!out_cm) {
      *reinterpret_cast<float4*>(&y[((size_t)b * Lp + t) * C + c]) = o;
    } else {   // channel-major (B, C, Lp): the layout the reference flattens for z_means / z_var
      float* yb = y + (size_t)b * C * Lp + t;
      yb[(size_t)(c + 0) * Lp] = o.x; yb[(size_t)(c + 1) * Lp] = o.y;
      yb[(size_t)(c + 2) * Lp] = o.z; yb[(size_t)(c + 3) * Lp] = o.w;
    }
  }
}

// dconv over ALL Lin rows of each molecule: pooled pairs route dy to the first maximal element
// (MaxPool1d keeps the first index on ties) when the pooled value is > 0; every other row is 0.
__global__ __launch_bounds__(256) void pool_relu_bwd_kernel(const float* __restrict__ conv, const float* __restrict__ dy,
                                                            float* __restrict__ dconv, int B, int Lin, int Lp, int C,
                                                            int dy_cm) {
  const size_t total = (size_t)B * Lin * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int c = (int)(i % C);
    size_t bt = i / C;
    int t = (int)(bt % Lin), b = (int)(bt / Lin);
    float o = 0.f;
    int tp = t >> 1;
    if (tp < Lp) {
      const float* p0 = conv + ((size_t)b * Lin + 2 * tp) * C + c;
      float u = p0[0], v = p0[C];
      float m = fmaxf(u, v);
      bool first = (u >= v);
      if (m > 0.f && ((t & 1) == 0) == first)
        o = dy_cm ? dy[((size_t)b * C + c) * Lp + tp] : dy[((size_t)b * Lp + tp) * C + c];
    }
    dconv[i] = o;
  }
}

// up[b, lpad + stride*i, c] = src[b, i, c] (zero elsewhere).  src has src_rows rows per molecule
// (token-major) or is channel-major (B, C, Lin).  stride 1 / lpad 0 is a plain row compaction.
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ src, int src_rows, int src_cm,
                                                           float* __restrict__ up, int B, int Lin, int C, int stride,
                                                           int lpad, int Lup) {
  const size_t total = (size_t)B * Lup * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int c = (int)(i % C);
    size_t bt = i / C;
    int t = (int)(bt % Lup), b = (int)(bt / Lup);
    int u = t - lpad;
    float o = 0.f;
    if (u >= 0 && u % stride == 0 && u / stride < Lin) {
      int si = u / stride;
      o = src_cm ? src[((size_t)b * C + c) * Lin + si] : src[((size_t)b * src_rows + si) * C + c];
    }
    up[i] = o;
  }
}

// dx[b, t, c] = (t < Lin && (y == NULL || y[b,t,c] > 0)) ? dup[b, lpad + stride*t, c] : 0 for
// t < dst_rows (token-major) or channel-major (B, C, Lin); y shares dx's layout (ReLU mask).
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dup, const float* __restrict__ y,
                                                           float* __restrict__ dx, int dst_rows, int dst_cm, int B,
                                                           int Lin, int C, int stride, int lpad, int Lup) {
  const size_t total = (size_t)B * dst_rows * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int c, t, b;
    if (!dst_cm) { c = (int)(i % C); size_t bt = i / C; t = (int)(bt % dst_rows); b = (int)(bt / dst_rows); }
    else { t = (int)(i % dst_rows); size_t bc = i / dst_rows; c = (int)(bc % C); b = (int)(bc / C); }
    float o = 0.f;
    if (t < Lin && (y == nullptr || y[i] > 0.f)) o = dup[((size_t)b * Lup + lpad + (size_t)t * stride) * C + c];
    dx[i] = o;
  }
}

__global__ void reparam_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ eps,
                               float* __restrict__ z, int n, float es) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) z[i] = mu[i] + eps[i] * es * expf(0.5f * logvar[i]);
}
__global__ void reparam_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ logvar,
                                   const float* __restrict__ eps, float* __restrict__ dmu, float* __restrict__ dlogvar,
                                   int n, float es) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    dmu[i] += dz[i];
    dlogvar[i] += dz[i] * eps[i] * es * 0.5f * expf(0.5f * logvar[i]);
  }
}

// pre_add == 0: out = (res ? res : 0) + x * keep      (generic SublayerConnection / nn.Dropout)
// pre_add == 1: out = (x + res[(row % res_rows), :]) * keep   (PositionalEncoding: res = pe table)
__global__ __launch_bounds__(256) void residual_dropout_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                               float* __restrict__ out, int rows, int N, int res_rows,
                                                               int pre_add, const uint32_t* rng, uint32_t site, float p) {
  DropCfg dc = drop_cfg(rng, site, p);
  const size_t total = (size_t)rows * N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    float ks = drop_scale(dc, (uint32_t)i);
    float v = x[i];
    if (pre_add) {
      size_t row = i / N, c = i % N;
      out[i] = (v + res[(row % res_rows) * N + c]) * ks;
    } else {
      out[i] = (res ? res[i] : 0.f) + v * ks;
    }
  }
}

// float4 form of the above for N % 4 == 0 and 16-byte aligned buffers: one hash group per float4
__global__ __launch_bounds__(256) void residual_dropout_vec_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                                   float* __restrict__ out, int rows, int N, int res_rows,
                                                                   int pre_add, const uint32_t* rng, uint32_t site, float p) {
  DropCfg dc = drop_cfg(rng, site, p);
  const size_t total4 = (size_t)rows * N / 4;
  for (size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (size_t)gridDim.x * 256) {
    const size_t i = i4 * 4;
    float ks[4];
    drop_scale4(dc, (uint32_t)i, ks);
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    float4 o;
    if (pre_add) {
      const size_t row = i / N, c = i % N;
      const float4 t = *reinterpret_cast<const float4*>(res + (row % res_rows) * N + c);
      o = make_float4((v.x + t.x) * ks[0], (v.y + t.y) * ks[1], (v.z + t.z) * ks[2], (v.w + t.w) * ks[3]);
    } else {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (res) t = *reinterpret_cast<const float4*>(res + i);
      o = make_float4(t.x + v.x * ks[0], t.y + v.y * ks[1], t.z + v.z * ks[2], t.w + v.w * ks[3]);
    }
    *reinterpret_cast<float4*>(out + i) = o;
  }
}

__global__ void rng_advance_kernel(uint32_t* rng) {
  if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += 1;   // vector store path (global_store_dword)
}

inline int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

#define LN_DISPATCH(NPL_, KERNEL, ...)                                          \
  case NPL_: hipLaunchKernelGGL((KERNEL<NPL_>), grid, dim3(256), 0, s, __VA_ARGS__); break;

#define LN_DISPATCH_B(NPL_, KERNEL, ...)                                        \
  case NPL_: hipLaunchKernelGGL((KERNEL<NPL_>), grid, block, 0, s, __VA_ARGS__); break;

extern "C" int tvae_layernorm_fwd(const float* x, const float* a, const float* b, float* y, int rows, int d,
                                  float eps, void* stream) {
  TVAE_CHECK(x && a && b && y, TVAE_E_NULL, "tvae_layernorm_fwd: null pointer");
  TVAE_CHECK(rows > 0 && d % 64 == 0 && d >= 64 && d <= 512, TVAE_E_SHAPE,
             "tvae_layernorm_fwd: d=%d must be a multiple of 64 in [64,512], rows=%d", d, rows);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(rows, 4));
  switch (d / 64) {
    LN_DISPATCH(1, ln_fwd_kernel, x, a, b, y, rows, eps)
    LN_DISPATCH(2, ln_fwd_kernel, x, a, b, y, rows, eps)
    LN_DISPATCH(3, ln_fwd_kernel, x, a, b, y, rows, eps)
    LN_DISPATCH(4, ln_fwd_kernel, x, a, b, y, rows, eps)
    LN_DISPATCH(5, ln_fwd_kernel, x, a, b, y, rows, eps)
    LN_DISPATCH(6, ln_fwd_kernel, x, a, b, y, rows, eps)
    LN_DISPATCH(7, ln_fwd_kernel, x, a, b, y, rows, eps)
    LN_DISPATCH(8, ln_fwd_kernel, x, a, b, y, rows, eps)
  }
  TVAE_LAUNCH_CHECK("tvae_layernorm_fwd");
  return TVAE_OK;
}

static int layernorm_bwd_impl(const float* x, const float* a, const float* dy, const float* add_to, float* dx,
                              float* da, float* db, int rows, int d, float eps, float* dxm, const uint32_t* rng,
                              uint32_t site, float p_drop, void* stream) {
  TVAE_CHECK(x && a && dy && dx && da && db, TVAE_E_NULL, "tvae_layernorm_bwd: null pointer");
  TVAE_CHECK(rows > 0 && d % 64 == 0 && d >= 64 && d <= 512, TVAE_E_SHAPE,
             "tvae_layernorm_bwd: d=%d must be a multiple of 64 in [64,512], rows=%d", d, rows);
  TVAE_CHECK(!dxm || (long)rows * d < (1L << 32), TVAE_E_SHAPE, "tvae_layernorm_bwd_masked: dropout index space exceeds 32 bits");
  hipStream_t s = (hipStream_t)stream;
  // The column sums da, db leave every workgroup as 2*d same-address atomics, and those, not the four
  // streams, were the kernel's limit (68 us with ~2000 workgroups of 4 waves, 47 us with the atomics
  // removed): ~1000 workgroups of 8 waves keep the waves per CU and halve the atomics -> 51 us.
  int rpw = (rows / 1024 + 15) / 16 * 16;
  rpw = rpw < 16 ? 16 : (rpw > 256 ? 256 : rpw);
  const int threads = 512;
  dim3 grid(cdiv(rows, rpw)), block(threads);
  switch (d / 64) {
    LN_DISPATCH_B(1, ln_bwd_kernel, x, a, dy, add_to, dx, da, db, rows, rpw, eps, dxm, rng, site, p_drop)
    LN_DISPATCH_B(2, ln_bwd_kernel, x, a, dy, add_to, dx, da, db, rows, rpw, eps, dxm, rng, site, p_drop)
    LN_DISPATCH_B(3, ln_bwd_kernel, x, a, dy, add_to, dx, da, db, rows, rpw, eps, dxm, rng, site, p_drop)
    LN_DISPATCH_B(4, ln_bwd_kernel, x, a, dy, add_to, dx, da, db, rows, rpw, eps, dxm, rng, site, p_drop)
    LN_DISPATCH_B(5, ln_bwd_kernel, x, a, dy, add_to, dx, da, db, rows, rpw, eps, dxm, rng, site, p_drop)
    LN_DISPATCH_B(6, ln_bwd_kernel, x, a, dy, add_to, dx, da, db, rows, rpw, eps, dxm, rng, site, p_drop)
    LN_DISPATCH_B(7, ln_bwd_kernel, x, a, dy, add_to, dx, da, db, rows, rpw, eps, dxm, rng, site, p_drop)
    LN_DISPATCH_B(8, ln_bwd_kernel, x, a, dy, add_to, dx, da, db, rows, rpw, eps, dxm, rng, site, p_drop)
  }
  TVAE_LAUNCH_CHECK("tvae_layernorm_bwd");
  return TVAE_OK;
}

extern "C" int tvae_layernorm_bwd(const float* x, const float* a, const float* dy, const float* add_to, float* dx,
                                  float* da, float* db, int rows, int d, float eps, void* stream) {
  return layernorm_bwd_impl(x, a, dy, add_to, dx, da, db, rows, d, eps, nullptr, nullptr, 0, 0.f, stream);
}

extern "C" int tvae_layernorm_bwd_masked(const float* x, const float* a, const float* dy, const float* add_to, float* dx,
                                         float* dx_masked, float* da, float* db, int rows, int d, float eps,
                                         const uint32_t* rng, uint32_t site, float p_drop, void* stream) {
  TVAE_CHECK(dx_masked && rng && p_drop > 0.f, TVAE_E_NULL, "tvae_layernorm_bwd_masked: needs dx_masked and an active dropout site");
  return layernorm_bwd_impl(x, a, dy, add_to, dx, da, db, rows, d, eps, dx_masked, rng, site, p_drop, stream);
}

extern "C" int tvae_colsum(const float* X, int ld, float* out, int M, int N, int accumulate, const uint32_t* rng,
                           uint32_t site, float p_drop, void* stream) {
  TVAE_CHECK(X && out, TVAE_E_NULL, "tvae_colsum: null pointer");
  TVAE_CHECK(M > 0 && N > 0 && ld >= N, TVAE_E_SHAPE, "tvae_colsum: bad shape M=%d N=%d ld=%d", M, N, ld);
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    if (hipMemsetAsync(out, 0, sizeof(float) * N, s) != hipSuccess) { tvae_set_error("tvae_colsum: memset failed"); return TVAE_E_LAUNCH; }
  }
  int rpw = 256;
  dim3 grid(cdiv(N, 64), cdiv(M, rpw));
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(64, 4), 0, s, X, ld, out, M, N, rpw, rng, site, p_drop);
  TVAE_LAUNCH_CHECK("tvae_colsum");
  return TVAE_OK;
}

extern "C" int tvae_embed_fwd(const int64_t* ids, const float* lut, const float* pe, float* out, int B, int L, int d,
                              int vocab, const uint32_t* rng, uint32_t site, float p_drop, void* stream) {
  TVAE_CHECK(ids && lut && pe && out, TVAE_E_NULL, "tvae_embed_fwd: null pointer");
  TVAE_CHECK(B > 0 && L > 0 && d % 4 == 0 && vocab > 0, TVAE_E_SHAPE, "tvae_embed_fwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  size_t total = (size_t)B * L * (d / 4);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, ids, lut, pe, out, B * L, L, d, vocab,
                     sqrtf((float)d), rng, site, p_drop);
  TVAE_LAUNCH_CHECK("tvae_embed_fwd");
  return TVAE_OK;
}

extern "C" int tvae_embed_bwd(const int64_t* ids, const float* dout, float* dlut, int B, int L, int d, int vocab,
                              const uint32_t* rng, uint32_t site, float p_drop, void* stream) {
  TVAE_CHECK(ids && dout && dlut, TVAE_E_NULL, "tvae_embed_bwd: null pointer");
  size_t lds = (size_t)vocab * d * sizeof(float);
  TVAE_CHECK(B > 0 && L > 0 && d > 0 && lds <= 64 * 1024, TVAE_E_SHAPE,
             "tvae_embed_bwd: vocab*d*4 = %zu bytes exceeds the 64 KiB LDS table", lds);
  hipStream_t s = (hipStream_t)stream;
  int rows = B * L, rpw = 128;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(cdiv(rows, rpw)), dim3(256), lds, s, ids, dout, dlut, rows, rpw, d, vocab,
                     sqrtf((float)d), rng, site, p_drop);
  TVAE_LAUNCH_CHECK("tvae_embed_bwd");
  return TVAE_OK;
}

extern "C" int tvae_conv_w_pack(const float* w, float* wp, int Cout, int Cin, int k, int transposed_flip, void* stream) {
  TVAE_CHECK(w && wp && Cout > 0 && Cin > 0 && k > 0, TVAE_E_SHAPE, "tvae_conv_w_pack: bad arguments");
  hipLaunchKernelGGL(conv_w_pack_kernel, dim3(grid_for((size_t)Cout * Cin * k)), dim3(256), 0, (hipStream_t)stream, w, wp,
                     Cout, Cin, k, transposed_flip, 0);
  TVAE_LAUNCH_CHECK("tvae_conv_w_pack");
  return TVAE_OK;
}
extern "C" int tvae_conv_w_unpack_grad(const float* dwp, float* dw, int Cout, int Cin, int k, int transposed_flip,
                                       void* stream) {
  TVAE_CHECK(dwp && dw && Cout > 0 && Cin > 0 && k > 0, TVAE_E_SHAPE, "tvae_conv_w_unpack_grad: bad arguments");
  hipLaunchKernelGGL(conv_w_pack_kernel, dim3(grid_for((size_t)Cout * Cin * k)), dim3(256), 0, (hipStream_t)stream, dwp,
                     dw, Cout, Cin, k, transposed_flip & 1, (transposed_flip & 2) ? 2 : 1);
  TVAE_LAUNCH_CHECK("tvae_conv_w_unpack_grad");
  return TVAE_OK;
}

extern "C" int tvae_deconv_poly_pack(const float* in, float* out, const float* bias_in, float* bias_out, int Cin, int Cout,
                                     int k, int stride, int mode, void* stream) {
  TVAE_CHECK(in && out, TVAE_E_NULL, "tvae_deconv_poly_pack: null pointer");
  TVAE_CHECK(Cin > 0 && Cout > 0 && k > 0 && stride >= 1 && mode >= 0 && mode <= 2, TVAE_E_SHAPE,
             "tvae_deconv_poly_pack: bad arguments (Cin %d Cout %d k %d stride %d mode %d)", Cin, Cout, k, stride, mode);
  const int Q = (k + stride - 1) / stride;
  hipLaunchKernelGGL(deconv_poly_pack_kernel, dim3(grid_for((size_t)stride * Cout * Q * Cin)), dim3(256), 0, (hipStream_t)stream,
                     in, out, bias_in, bias_out, Cin, Cout, k, stride, Q, mode);
  TVAE_LAUNCH_CHECK("tvae_deconv_poly_pack");
  return TVAE_OK;
}

extern "C" int tvae_pool_relu_fwd(const float* conv, float* y, int B, int Lin, int Lconv, int C, int out_cm,
                                  void* stream) {
  TVAE_CHECK(conv && y, TVAE_E_NULL, "tvae_pool_relu_fwd: null pointer");
  TVAE_CHECK(B > 0 && Lconv >= 2 && Lconv <= Lin && C % 4 == 0, TVAE_E_SHAPE, "tvae_pool_relu_fwd: bad shape");
  int Lp = Lconv / 2;
  hipLaunchKernelGGL(pool_relu_fwd_kernel, dim3(grid_for((size_t)B * Lp * C / 4)), dim3(256), 0, (hipStream_t)stream,
                     conv, y, B, Lin, Lp, C, out_cm);
  TVAE_LAUNCH_CHECK("tvae_pool_relu_fwd");
  return TVAE_OK;
}
extern "C" int tvae_pool_relu_bwd(const float* conv, const float* dy, float* dconv, int B, int Lin, int Lconv, int C,
                                  int dy_cm, void* stream) {
  TVAE_CHECK(conv && dy && dconv, TVAE_E_NULL, "tvae_pool_relu_bwd: null pointer");
  TVAE_CHECK(B > 0 && Lconv >= 2 && Lconv <= Lin, TVAE_E_SHAPE, "tvae_pool_relu_bwd: bad shape");
  hipLaunchKernelGGL(pool_relu_bwd_kernel, dim3(grid_for((size_t)B * Lin * C)), dim3(256), 0, (hipStream_t)stream, conv,
                     dy, dconv, B, Lin, Lconv / 2, C, dy_cm);
  TVAE_LAUNCH_CHECK("tvae_pool_relu_bwd");
  return TVAE_OK;
}

extern "C" int tvae_upsample_fwd(const float* x, int src_rows, int src_cm, float* up, int B, int Lin, int C,
                                 int stride, int lpad, int Lup, void* stream) {
  TVAE_CHECK(x && up, TVAE_E_NULL, "tvae_upsample_fwd: null pointer");
  TVAE_CHECK(B > 0 && stride >= 1 && lpad >= 0 && Lup >= lpad + (Lin - 1) * stride + 1 && (src_cm || src_rows >= Lin),
             TVAE_E_SHAPE, "tvae_upsample_fwd: Lup %d / src_rows %d too small", Lup, src_rows);
  hipLaunchKernelGGL(upsample_fwd_kernel, dim3(grid_for((size_t)B * Lup * C)), dim3(256), 0, (hipStream_t)stream, x,
                     src_rows, src_cm, up, B, Lin, C, stride, lpad, Lup);
  TVAE_LAUNCH_CHECK("tvae_upsample_fwd");
  return TVAE_OK;
}
extern "C" int tvae_upsample_bwd(const float* dup, const float* y, float* dx, int dst_rows, int dst_cm, int B,
                                 int Lin, int C, int stride, int lpad, int Lup, void* stream) {
  TVAE_CHECK(dup && dx, TVAE_E_NULL, "tvae_upsample_bwd: null pointer");
  TVAE_CHECK(B > 0 && stride >= 1 && lpad >= 0 && Lup >= lpad + (Lin - 1) * stride + 1 && dst_rows >= Lin &&
                 (!dst_cm || dst_rows == Lin),
             TVAE_E_SHAPE, "tvae_upsample_bwd: Lup %d / dst_rows %d inconsistent", Lup, dst_rows);
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for((size_t)B * dst_rows * C)), dim3(256), 0, (hipStream_t)stream,
                     dup, y, dx, dst_rows, dst_cm, B, Lin, C, stride, lpad, Lup);
  TVAE_LAUNCH_CHECK("tvae_upsample_bwd");
  return TVAE_OK;
}

extern "C" int tvae_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, int n,
                                float eps_scale, void* stream) {
  TVAE_CHECK(mu && logvar && eps && z && n > 0, TVAE_E_NULL, "tvae_reparam_fwd: bad arguments");
  hipLaunchKernelGGL(reparam_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, z, n,
                     eps_scale);
  TVAE_LAUNCH_CHECK("tvae_reparam_fwd");
  return TVAE_OK;
}
extern "C" int tvae_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmu, float* dlogvar,
                                int n, float eps_scale, void* stream) {
  TVAE_CHECK(dz && logvar && eps && dmu && dlogvar && n > 0, TVAE_E_NULL, "tvae_reparam_bwd: bad arguments");
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dz, logvar, eps, dmu,
                     dlogvar, n, eps_scale);
  TVAE_LAUNCH_CHECK("tvae_reparam_bwd");
  return TVAE_OK;
}

extern "C" int tvae_rng_advance(uint32_t* rng, void* stream) {
  TVAE_CHECK(rng, TVAE_E_NULL, "tvae_rng_advance: null pointer");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rng);
  TVAE_LAUNCH_CHECK("tvae_rng_advance");
  return TVAE_OK;
}

extern "C" int tvae_residual_dropout(const float* x, const float* res, float* out, int rows, int N, int res_rows,
                                     int pre_add, const uint32_t* rng, uint32_t site, float p_drop, void* stream) {
  TVAE_CHECK(x && out && rows > 0 && N > 0, TVAE_E_NULL, "tvae_residual_dropout: bad arguments");
  TVAE_CHECK(!pre_add || (res && res_rows > 0), TVAE_E_NULL, "tvae_residual_dropout: pre_add needs a table");
  TVAE_CHECK((long)rows * N < (1L << 32), TVAE_E_SHAPE, "tvae_residual_dropout: index space exceeds 32 bits");
  if (N % 4 == 0 && aligned16(x) && aligned16(out) && (!res || aligned16(res)))
    hipLaunchKernelGGL(residual_dropout_vec_kernel, dim3(grid_for((size_t)rows * N / 4)), dim3(256), 0, (hipStream_t)stream,
                       x, res, out, rows, N, res_rows, pre_add, rng, site, p_drop);
  else
    hipLaunchKernelGGL(residual_dropout_kernel, dim3(grid_for((size_t)rows * N)), dim3(256), 0, (hipStream_t)stream, x, res,
                       out, rows, N, res_rows, pre_add, rng, site, p_drop);
  TVAE_LAUNCH_CHECK("tvae_residual_dropout");
  return TVAE_OK;
}
