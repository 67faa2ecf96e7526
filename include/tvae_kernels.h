/*
 * tvae_kernels.h -- C ABI of libtvae_hip.so: the MI355X (gfx950) kernels under the
 * TransVAE transformer-VAE train step and greedy decode.
 *
 * The reference (oriondollar/TransVAE) has no FFI / plugin interface: its hot path is a
 * chain of stock ATen ops called from nn.Module.forward.  Each entry
 * point below therefore cites the reference Python call site (file:line under
 * /root/reference) whose arithmetic it replaces.  The Python host (transvae_amd/) binds
 * these with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless
 *     a parameter is documented as host;
 *   - every function is asynchronous on `stream` (a hipStream_t passed as void*), never
 *     allocates and never synchronises;
 *   - return 0 on success, a negative TVAE_E_* code otherwise; tvae_last_error() returns a
 *     thread-local message for the last failure;
 *   - matrices are row-major fp32; "ld" is the row stride in elements;
 *   - `dtype`: TVAE_F32 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32), TVAE_BF16 = operands
 *     rounded to bf16 on the way into LDS, fp32 accumulate (v_mfma_f32_32x32x16_bf16);
 *     TVAE_F32X3 = every fp32 operand split exactly into hi+mid+lo bf16 and the six partial
 *     products >= 2^-24 accumulated in fp32 on the bf16 matrix cores (error at fp32's own
 *     rounding level; GEMM only, the other kernels compute in fp32); tensors in HBM stay fp32
 *     in all modes;
 *   - dropout: `rng` points at two device uint32 {seed, step}; `site` names the call site;
 *     the keep-mask is a counter-based hash of (seed, step, site, element index), so the
 *     backward kernels regenerate it instead of storing it.  p == 0 or rng == NULL: off.
 *     p is quantised to 1/65536 (0.1 -> 6554/65536) and kept elements are scaled by exactly
 *     1 / P(keep).
 */
#ifndef TVAE_KERNELS_H
#define TVAE_KERNELS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TVAE_OK 0
#define TVAE_E_SHAPE (-1)   /* unsupported or inconsistent shape / stride / alignment */
#define TVAE_E_DTYPE (-2)
#define TVAE_E_LAUNCH (-3)  /* hipGetLastError() after the launch was not hipSuccess */
#define TVAE_E_NULL (-4)
#define TVAE_E_DATA (-5)    /* tvae_encode_smiles: an input row cannot be encoded (see TVAE_TOK_*) */

/* why a SMILES row was refused (tvae_encode_smiles) and what the reference raises for it */
#define TVAE_TOK_UNMATCHED 1 /* text outside the tokenizer's forms: AssertionError, tvae_util.py:102 */
#define TVAE_TOK_UNKNOWN 2   /* token not in CHAR_DICT: KeyError, tvae_util.py:121 */
#define TVAE_TOK_TOO_LONG 3  /* more than max_len tokens: the row assignment fails, data.py:31 */

#define TVAE_F32 0
#define TVAE_BF16 1
#define TVAE_F32X3 2   /* fp32-accurate GEMM from six bf16 MFMAs on exact 3-way bf16 splits of the operands */

/* gemm flags */
#define TVAE_GEMM_RELU 1        /* v = max(v, 0) after bias */
#define TVAE_GEMM_ACCUM 2       /* C += v instead of C = v (plain read-modify-write) */
#define TVAE_GEMM_A_KMAJOR 4    /* A(m,k) = A[k*lda + m]  (default A[m*lda + k]) */
#define TVAE_GEMM_B_KMAJOR 8    /* B(k,n) = B[k*ldb + n]  (default B[n*ldb + k], i.e. nn.Linear weight) */
#define TVAE_GEMM_SPLITK 16     /* split K over grid.z, fp32 atomicAdd into C (C must be initialised); bias / alpha only: ReLU, residual and output dropout are refused */
#define TVAE_GEMM_RELUMASK 64   /* v = residual[m,n] > 0 ? v : 0 (`residual` is the saved activation, not added) */
#define TVAE_GEMM_B_SPLIT3 128  /* dtype F32X3 only: B points at bf16 planes written by tvae_split3, ldb = 3*Kb in bf16 elements */
#define TVAE_GEMM_B_BF16 256    /* dtype BF16 only: B points at bf16 values (N, ldb) written by tvae_cast_bf16; row-major A, K % 32 == 0 */
#define TVAE_GEMM_A_DROPMASK 32 /* multiply A(m,k) by the dropout keep-scale of (rng, site, m*K+k): dgrad/wgrad of a dropped-out output */

const char* tvae_last_error(void);
const char* tvae_version(void);

/*
 * C[M,N] = epilogue( alpha * sum_k A(m,k) B(k,n) ), epilogue: + bias[n] -> ReLU -> dropout(p)
 * -> + residual[m*ldr+n].  Replaces every nn.Linear on the path
 * (transvae/trans_models.py:747, 790, 792-793, 852, 929-930, 941, 952, 1030) together with
 * F.relu / nn.Dropout / the residual add of SublayerConnection (:1101), their backward
 * data- and weight-gradient products, and, via overlapping rows (lda < K), the Conv1d /
 * ConvTranspose1d contractions of ConvBottleneck / DeconvBottleneck (:980-983, :1006-1009).
 * Unaligned operands (ld or K not a multiple of 4) take a scalar-load path.
 * a_rowsum (k-major A only, else NULL): a_rowsum[m] += sum_k A(m,k) with the A dropout scale applied --
 * the bias gradient sum_rows dY, produced by the weight-gradient launch that reads dY anyway.
 */
int tvae_gemm(const float* A, int lda, const float* B, int ldb, float* C, int ldc,
              int M, int N, int K, const float* bias, const float* residual, int ldr,
              float alpha, int flags, const uint32_t* rng, uint32_t site, float p_drop,
              float* a_rowsum, int dtype, void* stream);

/*
 * `batch` independent products of one shape in one launch: C_i = epilogue(op(A_i) op(B_i)), i = 0 .. batch-1, the
 * matrices of consecutive entries strideA / strideB / strideC (/ strideR) elements apart; flags as tvae_gemm without
 * split-K, dropout or pre-converted operands; exact fp32.  Replaces torch.bmm(attn_weights, mem) of RNNAttnEncoder
 * (transvae/rnn_models.py:247) and the two products of its backward.
 */
int tvae_gemm_batched(const float* A, int lda, long strideA, const float* B, int ldb, long strideB,
                      float* C, int ldc, long strideC, int M, int N, int K, int batch,
                      const float* bias, const float* residual, int ldr, long strideR,
                      float alpha, int flags, void* stream);

/*
 * The recurrence of one torch.nn.GRU layer (transvae/rnn_models.py:225/241, 284/301, 320/333, 362/378), gate order
 * r, z, n as in torch:   gh = h_{t-1} W_hh^T + b_hh,  r = sigmoid(gi_r + gh_r),  z = sigmoid(gi_z + gh_z),
 * n = tanh(gi_n + r * gh_n),  h_t = (1 - z) n + z h_{t-1}.   gi = x W_ih^T + b_ih for ALL time steps is one tvae_gemm
 * made by the caller.  Buffers are [B, T, width] (time_major 0: row b * T + t) or [T, B, width] (time_major 1).
 *   fwd: gi (.., 3H), w_hh (3H, H), b_hh (3H), h0 (B, H) or NULL = zeros -> hseq (.., H); gates (.., 4H) receives
 *        r, z, n, gh_n for the backward pass (NULL: inference).  T launches, one per time step.
 *   bwd: dhseq (.., H) gradient arriving at every h_t and/or dh_last (B, H) at the final state (either may be NULL);
 *        w_hh_t = W_hh^T (H, 3H) -> dgi (.., 3H) gradient of gi, dgh (.., 3H) gradient of gh; carry (B, H) is workspace.
 *        The caller finishes with plain GEMMs: dx = dgi W_ih, dW_ih = dgi^T x, db_ih = colsum dgi,
 *        dW_hh = dgh[:, 1:]^T hseq[:, :-1] (+ dgh[:, 0]^T h0), db_hh = colsum dgh.  The gradient of h0 is not produced
 *        (the reference always starts from zeros, rnn_models.py:262-263).
 * A sequence may be processed in time CHUNKS (time-major buffers, pointers advanced by t0 * B rows, h0 = the rows of step
 * t0 - 1): forward chunks in ascending order; backward chunks in descending order with `continues` = 1 on every chunk but
 * the one that holds the last time step (the call then takes the step after its last from the dgh rows that follow its
 * own and from `carry`, which the later chunk's call left).  That is what lets the layers of a stack run as a pipeline on
 * separate streams (ops.GRUStackFn).
 * H must be a multiple of 64.
 */
int tvae_gru_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h0, float* hseq, float* gates,
                 int T, int B, int H, int time_major, void* stream);
int tvae_gru_bwd(const float* dhseq, const float* dh_last, const float* hseq, const float* gates, const float* h0,
                 const float* w_hh_t, float* dgi, float* dgh, float* carry, int T, int B, int H, int time_major,
                 int continues, void* stream);

/*
 * Row softmax y = softmax(x) over n columns and its backward dx = y * (dy - sum(y dy)): the attention weights of
 * RNNAttnEncoder, F.softmax(self.attn(cat(x, mem)), dim=2) (transvae/rnn_models.py:246).
 */
int tvae_softmax_rows_fwd(const float* x, int ldx, float* y, int ldy, int rows, int n, void* stream);
int tvae_softmax_rows_bwd(const float* y, int ldy, const float* dy, int lddy, float* dx, int lddx, int rows, int n,
                          void* stream);

/*
 * Three-way bf16 split of an fp32 matrix w (rows x cols, row stride ldw) for the F32X3 arithmetic:
 * w = hi + mid + lo exactly to 2^-26, each a bf16.  out_n = the planes of w (rows n = matrix rows, contraction index
 * k = columns; cols % 16 == 0), out_t = the planes of w^T (rows % 16 == 0); either may be NULL.  Layout, tile-major
 * (csrc/gemm3_split.h): blocks of 64 rows x 16 k, block (n / 64, k / 16) at ((n / 64) * (K / 16) + k / 16) * 3072 elements,
 * inside a block six pieces of 512 elements [k-half (k / 8) % 2][plane hi, mid, lo][row n % 64][8 values] -- one piece is
 * what one LDS-DMA instruction of the streamed kernel moves.  Rows are padded with zeros to a multiple of 64, written by
 * this call: out_n holds ceil64(rows) * 3 * cols elements, out_t ceil64(cols) * 3 * rows.  Lets tvae_gemm take an nn.Linear
 * weight (forward: out_n; data gradient: out_t) as a pre-split B operand (TVAE_GEMM_B_SPLIT3, ldb = 3 * K of the split), so
 * that the kernel only splits the activation operand.  Weights change once per optimizer step.
 */
int tvae_split3(const float* w, int ldw, uint16_t* out_n, uint16_t* out_t, int rows, int cols, void* stream);

/*
 * bf16 copies of an fp32 matrix w (rows x cols, row stride ldw), round-to-nearest-even: out_n (rows, cols) and out_t
 * (cols, rows) = w^T; either may be NULL.  Lets tvae_gemm in dtype BF16 take an nn.Linear weight as a ready bf16 B
 * operand (TVAE_GEMM_B_BF16: forward x W^T against out_n, data gradient dY W against out_t, both in the streamed
 * NT kernel that moves its tiles global -> LDS without touching registers).  Weights change once per optimizer step.
 */
int tvae_cast_bf16(const float* w, int ldw, uint16_t* out_n, uint16_t* out_t, int rows, int cols, void* stream);

/* out[n] (+)= sum_m X[m*ld + n] (* dropout keep-scale if rng): bias gradients. */
int tvae_colsum(const float* X, int ld, float* out, int M, int N, int accumulate,
                const uint32_t* rng, uint32_t site, float p_drop, void* stream);

/*
 * LayerNorm of transvae/trans_models.py:1084-1087: y = a*(x-mean)/(std_unbiased+eps)+b.
 * bwd: dx = dL/dx (+ add_to, if given: the residual-stream gradient of SublayerConnection,
 * :1101) ; da, db are ACCUMULATED with atomics (caller zeroes them).
 */
int tvae_layernorm_fwd(const float* x, const float* a, const float* b, float* y,
                       int rows, int d, float eps, void* stream);
int tvae_layernorm_bwd(const float* x, const float* a, const float* dy, const float* add_to,
                       float* dx, float* da, float* db, int rows, int d, float eps,
                       void* stream);
/* The same, also writing dx_masked = dx * keep-scale of the dropout site (rng, site, p_drop) indexed (row, column) with
 * pitch d: dx is the gradient arriving at the OUTPUT of the SublayerConnection in front (x + dropout(sublayer(norm(x))),
 * transvae/trans_models.py:1101), whose two gradient products need it masked by that sublayer's residual dropout. */
int tvae_layernorm_bwd_masked(const float* x, const float* a, const float* dy, const float* add_to, float* dx,
                              float* dx_masked, float* da, float* db, int rows, int d, float eps,
                              const uint32_t* rng, uint32_t site, float p_drop, void* stream);

/*
 * Fused multi-head attention, one workgroup per (molecule, head), score tile kept on chip.
 * Replaces attention() (transvae/tvae_util.py:39-48) as called from
 * MultiHeadedAttention.forward (transvae/trans_models.py:929-937): softmax(QK^T*scale with
 * masked entries filled by -1e9) -> dropout -> PV, heads concatenated.
 *   q: row (b, i) at q + (b*Lq + i)*ldq + head*dk ; k, v likewise with Lk ; o with ldo.
 *   key_mask  (B, Lk) uint8 or NULL     -- the (B,1,Lk) src_mask of the reference
 *   full_mask (B, Lq, Lk) uint8 or NULL -- an arbitrary (B,Lq,Lk) mask
 *   causal != 0: additionally mask k > i (subsequent_mask, transvae/tvae_util.py:32-37)
 *   lse (B, h, Lq, 2): {row max, 1 / row sum} of each score row, consumed by the backward
 *   p_out (B, h, Lq, Lk) or NULL: the softmax weights before dropout (MultiHeadedAttention.attn,
 *   forward_w_attn: transvae/trans_models.py:796-813, 866-880)
 *   dtype: TVAE_F32 and TVAE_F32X3 run the products as exact fp32 MFMAs; TVAE_BF16 rounds their operands
 *   to bf16 on the way into the bf16 matrix cores (softmax, statistics and masks stay fp32; a backward
 *   call must use the dtype of its forward: it recomputes the forward's P).
 * Limits: Lq, Lk <= 128, dk in {32, 64}.
 */
int tvae_attn_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                  float* o, int ldo, float* lse, float* p_out,
                  const uint8_t* key_mask, const uint8_t* full_mask, int causal,
                  int B, int h, int Lq, int Lk, int dk, float scale,
                  const uint32_t* rng, uint32_t site, float p_drop, int dtype, void* stream);
/* Backward of the above.  dq/dk/dv use the same addressing as q/k/v (may be strided views);
 * every element of the addressed head slices is written.  dsum is workspace, 16-byte aligned, of
 * R4 + 4*R + B*h floats with R = B*h*Lq and R4 = R rounded up to a multiple of 4: the per-row sum_j P_j dP_j,
 * four 32-bit dropout keep words per row and one word per (molecule, head) naming its live key tiles (written by the
 * dQ kernel, read by the dK/dV kernel, which deals its waves to the key tiles that hold a visible key). */
int tvae_attn_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                  const float* o, int ldo, const float* d_o, int lddo, const float* lse,
                  float* dq, int lddq, float* dk_, int lddk, float* dv, int lddv, float* dsum,
                  const uint8_t* key_mask, const uint8_t* full_mask, int causal,
                  int B, int h, int Lq, int Lk, int dk, float scale,
                  const uint32_t* rng, uint32_t site, float p_drop, int dtype, void* stream);

/*
 * Embeddings + PositionalEncoding (+dropout): out[b,t,:] = lut[ids[b,t]]*sqrt(d) + pe[t,:]
 * (transvae/trans_models.py:1042-1043, 1060-1063).  ids are int64 (torch .long()).
 * bwd: dlut[id] += sqrt(d) * dout (atomics; caller zeroes dlut (vocab,d)).
 */
int tvae_embed_fwd(const int64_t* ids, const float* lut, const float* pe, float* out,
                   int B, int L, int d, int vocab, const uint32_t* rng, uint32_t site,
                   float p_drop, void* stream);
int tvae_embed_bwd(const int64_t* ids, const float* dout, float* dlut, int B, int L, int d,
                   int vocab, const uint32_t* rng, uint32_t site, float p_drop, void* stream);

/*
 * ConvBottleneck helpers (transvae/trans_models.py:976-983).  Activations are TOKEN-major
 * (B, L, C) so that the im2col row of output position t is the contiguous slice
 * x[b, t:t+k, :] and Conv1d becomes one tvae_gemm with lda = C, K = k*C over all B*L rows
 * (rows whose window crosses a molecule boundary are computed and ignored).
 *   conv weight (Cout, Cin, k) -> GEMM weight (Cout, k*Cin)           [tvae_conv_w_pack]
 *   and back for its gradient [tvae_conv_w_unpack_grad]; there bit 1 of transposed_flip says that dwp arrives
 *   transposed, (k*Cin, Cout): the weight-gradient product with its operands swapped, so that N = Cout (148 / 160
 *   channels) lies on the 160-wide tile of tvae_gemm instead of M on two 128-row tiles
 *   pool: y[b,t,c] = relu(max(conv[b,2t,c], conv[b,2t+1,c])), t < Lp = Lconv/2 ; out_cm: y is
 *         written channel-major (B, C, Lp), the layout trans_models.py:789 flattens
 *   pool_bwd: dconv[b,t',c] = dy routed to the arg-max element when y > 0, zero elsewhere,
 *             for ALL rows t' < Lin (rows >= Lconv are the ignored rows: zero).
 */
int tvae_conv_w_pack(const float* w, float* wp, int Cout, int Cin, int k, int transposed_flip,
                     void* stream);
int tvae_conv_w_unpack_grad(const float* dwp, float* dw, int Cout, int Cin, int k,
                            int transposed_flip, void* stream);
int tvae_pool_relu_fwd(const float* conv, float* y, int B, int Lin, int Lconv, int C, int out_cm,
                       void* stream);
int tvae_pool_relu_bwd(const float* conv, const float* dy, float* dconv, int B, int Lin,
                       int Lconv, int C, int dy_cm, void* stream);
/*
 * DeconvBottleneck helpers (transvae/trans_models.py:1001-1009).  ConvTranspose1d(k, stride s,
 * padding p) == Conv1d with the flipped kernel over the input with s-1 zeros inserted between
 * samples and k-1-p zeros at both ends:
 *   up[b, lpad + s*i, c] = x[b,i,c] (lpad = k-1-p), everything else 0,
 *   Lup = (Lin-1)*s + 1 + 2*lpad
 * followed by tvae_gemm (weights packed with transposed_flip=1) with fused bias+ReLU.
 *   x is token-major with src_rows rows per molecule, or channel-major (B, C, Lin) if src_cm
 *   (the (B,64,9) view of decoder.linear's output, trans_models.py:853).  stride=1, lpad=0
 *   compacts rows (src_rows -> Lup rows per molecule).
 *   upsample_bwd: dx[b,t,c] = (t < Lin && (y == NULL || y[b,t,c] > 0)) ? dup[b, lpad+s*t, c] : 0
 *   for t < dst_rows; y (ReLU mask source) and dx share a layout (token-major, or channel-major
 *   if dst_cm).  Rows t >= Lin are zeroed: they are the rows a conv-as-GEMM ignores.
 */
int tvae_upsample_fwd(const float* x, int src_rows, int src_cm, float* up, int B, int Lin, int C,
                      int stride, int lpad, int Lup, void* stream);
int tvae_upsample_bwd(const float* dup, const float* y, float* dx, int dst_rows, int dst_cm, int B,
                      int Lin, int C, int stride, int lpad, int Lup, void* stream);
/*
 * ConvTranspose1d with stride s > 1 (transvae/trans_models.py:1001-1009, nn.ConvTranspose1d at :995-998) as a POLYPHASE
 * product: output position t + pad = m s + r only receives the taps j = r + s q (q < Q = ceil(k/s)) from the inputs
 * i = m - q, so all s phases are ONE tvae_gemm over the un-upsampled input (Q - 1 leading zero rows per molecule,
 * tvae_upsample_fwd with stride 1 / lpad Q-1): A rows overlap (lda = Cin, K = Q Cin), N = s Cout, and the output rows
 * (b, m) x (r, co) are the token-major rows (b, m s + r) x co.  The FLOPs are those of the transposed convolution itself
 * (the zero-stuffed form multiplies s - 1 zero rows out of s).  The data gradient is the same shape of product over the
 * gradient rows (lda = s Cout, K = Q s Cout, N = Cin), the weight gradient a split-K product into the mode-0 layout.
 *   mode 0: w (Cin, Cout, k) -> out (s Cout, Q Cin), out[(r,co),(qq,ci)] = w[ci,co, r + s (Q-1-qq)] (0 if >= k);
 *           bias_out (s Cout) = bias_in tiled s times (either may be NULL)
 *   mode 1: w -> out (Cin, Q s Cout), out[ci,(q,r,co)] = w[ci,co, r + s q] (0 if >= k): weight of the data gradient
 *   mode 2: in = weight gradient in the mode-0 layout -> out (Cin, Cout, k); bias_out[co] = sum_r bias_in[(r,co)]
 */
int tvae_deconv_poly_pack(const float* in, float* out, const float* bias_in, float* bias_out, int Cin, int Cout,
                          int k, int stride, int mode, void* stream);

/* z = mu + eps*eps_scale*exp(0.5*logvar) (transvae/trans_models.py:771-775) and its backward
 * (dmu += dz ; dlogvar += dz*eps*eps_scale*0.5*exp(0.5*logvar)). */
int tvae_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z,
                     int n, float eps_scale, void* stream);
int tvae_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmu,
                     float* dlogvar, int n, float eps_scale, void* stream);

/*
 * trans_vae_loss (transvae/loss.py:23-38) forward AND backward in one pass.
 *   tokens (B, 127) int64: targets y = tokens[:,1:]-1
 *   true_len (B) int64 or NULL: length class per molecule; NULL = #(tokens != pad_idx), which
 *   is what the reference passes (src_mask.sum(-1), trans_models.py:219)
 *   logits (B*126, ldl >= 25) ; weights (25) ; pred_len (B,128) ; mu/logvar (B,dl)
 *   pred_prop (B) / true_prop (B) or NULL
 *   n_groups: the batch is n_groups consecutive chunks, each normalised on its own exactly like one
 *   chunk of VAEShell.train (trans_models.py:202-236); gradients are those of the SUM of the chunk losses
 *   out5 (n_groups, 5) = {total, bce_mol, bce_mask, kld, mse} per chunk (written, not accumulated)
 *   dlogits (B*126, ldl), dpred_len, dmu, dlogvar, dpred_prop: gradients of `total`
 *   scratch: >= 8*n_groups floats of device workspace (zeroed by the call)
 * KLD NaN guard (loss.py:36-37): if the KLD term is NaN it contributes 0 and no gradient.
 * pred_len NULL with n_len 0 (and dpred_len NULL): no length head -- vae_loss of the RNN models (loss.py:8-21),
 * bce_mask is reported as 0.
 */
int tvae_loss_fwd_bwd(const int64_t* tokens, const float* logits, int ldl, const float* weights,
                      const float* pred_len, const float* mu, const float* logvar,
                      const float* pred_prop, const float* true_prop, const int64_t* true_len,
                      float beta,
                      int B, int L, int n_cls, int n_len, int d_latent, int pad_idx, int n_groups,
                      float* out5, float* dlogits, float* dpred_len, float* dmu, float* dlogvar,
                      float* dpred_prop, float* scratch, void* stream);

/*
 * The same computation for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE: it exists only for those callers.  The
 * partial sums that tvae_loss_fwd_bwd keeps in the caller's `scratch` live here in a device array built into
 * the library (8 floats per chunk, TVAE_LOSS_STATIC_MAX_GROUPS chunks = 32 KB on every GPU the library runs
 * on, held until the process ends); the call allocates nothing and frees nothing, n_groups beyond the maximum
 * is refused with TVAE_E_SHAPE.  Every call on a GPU shares that GPU's array (it is cleared on `stream` at the
 * start of each call), so two calls that overlap on ONE GPU -- different streams, or different threads -- race;
 * calls on one stream, or on different GPUs, do not.  The device that is current at the call must be the one
 * `stream` and the buffers belong to.  The array's address is looked up on the FIRST call on each device (a symbol
 * lookup, which may load the code object): make one ordinary, un-captured call per device before capturing a graph
 * that contains this entry point.
 */
#define TVAE_LOSS_STATIC_MAX_GROUPS 1024
int tvae_loss_fwd_bwd_static(const int64_t* tokens, const float* logits, int ldl, const float* weights,
                             const float* pred_len, const float* mu, const float* logvar,
                             const float* pred_prop, const float* true_prop, const int64_t* true_len,
                             float beta,
                             int B, int L, int n_cls, int n_len, int d_latent, int pad_idx, int n_groups,
                             float* out5, float* dlogits, float* dpred_len, float* dmu, float* dlogvar,
                             float* dpred_prop, void* stream);

/*
 * Fused multi-tensor Adam (torch.optim.Adam as built at transvae/trans_models.py:703-705,
 * driven by NoamOpt.step transvae/opt.py:21-30 or AdamOpt :47-49).
 *   ptrs: device array of 4*n_tensors pointers laid out [p_0..][g_0..][m_0..][v_0..]
 *   sizes: device int64[n_tensors]; chunk_tensor/chunk_start: device int32 tables, one entry
 *   per 4096-element chunk (built once by the host)
 *   beta1, beta2, eps and the four numbers below are DOUBLES: 1 - beta is formed in double before it is rounded to
 *   the kernel's fp32 (float(0.999) is 0.99900001, and 1 - that is off by 1.3e-5 relative -- visible in the update).
 *   lr, bias_correction1 (1 - beta1^step), bias_correction2 (1 - beta2^step), grad_scale (every gradient is
 *   multiplied by it first: 1/world under data parallelism): BY VALUE, so that a host that queues steps ahead
 *   of the GPU cannot overwrite the numbers of a step that has not run yet.
 *   hyper: NULL, or device float[4] = {lr, bias_correction1, bias_correction2, grad_scale} that REPLACES the
 *   four by-value arguments (for a captured graph that is replayed with new numbers; the caller then owns the
 *   ordering of its writes to that buffer against the launches that read it)
 *   zero_grad != 0: g is zeroed after use (model.zero_grad(), trans_models.py:238)
 */
int tvae_adam_multi(void* const* ptrs, const int64_t* sizes, const int32_t* chunk_tensor,
                    const int64_t* chunk_start, int n_tensors, int n_chunks,
                    const float* hyper, double lr, double bias_correction1, double bias_correction2,
                    double grad_scale, double beta1, double beta2, double eps, int zero_grad,
                    void* stream);

/*
 * KV-cached greedy decode step kernels (replace the full-recompute loop of
 * VAEShell.greedy_decode, transvae/trans_models.py:418-434).
 *   decode_attn: one query row per (molecule, head) against n_keys cached rows
 *     q (n, ldq) ; kc, vc: row (b, j) at kc + (b*Lcache + j)*ldc + head*dk ;
 *     key_mask (n, ld_mask) uint8 or NULL: keys with a 0 byte are masked (-1e9 fill, so an
 *     all-masked row attends uniformly exactly like the reference)
 *   argmax_next: next[b] = 1 + first arg max_c logits[b, c] ; written to ids[b*ld_ids + col]
 *     as int64 (torch.max tie rule, trans_models.py:429-431)
 */
int tvae_decode_attn(const float* q, int ldq, const float* kc, const float* vc, int ldc,
                     int Lcache, float* o, int ldo, const uint8_t* key_mask, int ld_mask,
                     int n_keys, int n, int h, int dk, float scale, void* stream);
int tvae_argmax_next(const float* logits, int ldl, int n_cls, int64_t* ids, int ld_ids,
                     int col, int n, void* stream);
/* copy rows: dst[(b*Ldst + t)*ldd + c] = src[b*lds + c], c < width (append K/V to the cache) */
int tvae_scatter_rows(const float* src, int lds, float* dst, int ldd, int Ldst, int t,
                      int n, int width, void* stream);
/* mask_len[b] = first arg max of logits (n, n_cls) as int32 (predict_mask_length, :763-769) */
int tvae_argmax_rows(const float* logits, int ldl, int n_cls, int32_t* out, int n, void* stream);
/* out[b,:] = lut[ids[b*ld_ids + col]]*sqrt(d) + pe[col,:]  (tgt_embed of the newest token) */
int tvae_embed_step(const int64_t* ids, int ld_ids, int col, const float* lut, const float* pe,
                    float* out, int n, int d, int vocab, void* stream);

/*
 * Generic residual / dropout glue for callers that use the reference's modules one by one
 * (SublayerConnection.forward with an arbitrary sublayer, trans_models.py:1099-1101;
 * PositionalEncoding.forward, :1060-1063):
 *   pre_add == 0: out = (res ? res : 0) + x * keep
 *   pre_add == 1: out = (x + res[row % res_rows, :]) * keep      (res = positional table)
 */
int tvae_residual_dropout(const float* x, const float* res, float* out, int rows, int N,
                          int res_rows, int pre_add, const uint32_t* rng, uint32_t site,
                          float p_drop, void* stream);

/*
 * Data pipeline in front of the train step.
 *
 * tvae_encode_smiles -- HOST function, every pointer is host memory, synchronous, uses up to n_threads
 * threads (0 = one per core, at most 16).  Does what vae_data_gen does per molecule
 * (transvae/data.py:26-31): tokenizer() (tvae_util.py:97-103, the regex alternatives in order), then
 * encode_smiles() (tvae_util.py:114-122: '<end>' once, then '_' up to max_len; nothing appended when the
 * SMILES already has max_len tokens), then the leading start id.  One byte per token instead of the
 * reference's float32.
 *   text/offsets: string i is text[offsets[i] .. offsets[i+1]) (offsets has n+1 entries)
 *   vocab/vocab_offsets/vocab_ids: CHAR_DICT as n_vocab byte strings and their ids (ids < 256)
 *   out: n rows of ld >= max_len+1 bytes: [start_id, ids..., end_id, pad_id ...]
 *   returns TVAE_E_DATA if a row cannot be encoded; bad_row / bad_code (may be NULL) name the row the
 *   reference would have failed on first and why (TVAE_TOK_*): an untokenizable string anywhere comes before
 *   unknown tokens and over-long rows because the reference tokenizes the whole column first.
 *
 * tvae_gather_batch -- DEVICE: out[b, t] = float(ids[r_b*ld_ids + t]) for t < L, out[b, L] = props[r_b]
 * (0 when props is NULL), r_b = index[b] or first + b when index is NULL: the (B, L+1) float32 batch the
 * reference's DataLoader collates from the float tensor of vae_data_gen (trans_models.py:176-183), built
 * from a uint8 dataset that stays resident in HBM.  An index outside [0, n_rows) yields a row of zeros.
 */
int tvae_encode_smiles(const char* text, const int64_t* offsets, int64_t n, const char* vocab,
                       const int32_t* vocab_offsets, const int32_t* vocab_ids, int n_vocab,
                       int start_id, int end_id, int pad_id, int max_len, uint8_t* out, int64_t ld,
                       int n_threads, int64_t* bad_row, int* bad_code);
int tvae_gather_batch(const uint8_t* ids, int64_t ld_ids, const float* props, const int64_t* index,
                      int64_t first, int64_t n_rows, int B, int L, float* out, int ldo, void* stream);

/* bump the dropout step counter: rng[1] += 1 (one tiny launch per train step, graph-safe) */
int tvae_rng_advance(uint32_t* rng, void* stream);

#ifdef __cplusplus
}
#endif
#endif
