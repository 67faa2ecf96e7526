"""Per-kernel parity: every C-ABI entry point against a float64 CPU restatement of the same op
(torch CPU ops / oracle functions) on seeded inputs.  Needs the MI355X.

Tolerances: fp32 kernels are exact-fp32 MFMA chains, so outputs are compared at ~1e-5 relative
L2 (the north-star bound for loss/gradients); integer outputs are bit-exact.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from helpers import rel_l2  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale)


def g32(t):
    return t.to(torch.float32).to(dev()).contiguous()


TOL = 2e-6


# ------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (77, 25, 30), (1000, 256, 1232), (129, 131, 36), (5, 7, 4)])
def test_gemm_nt_epilogues(M, N, K):
    from transvae_amd import ops
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    y = ops.linear_fwd(g32(x), g32(w), g32(b))
    assert rel_l2(y.cpu(), x @ w.T + b) < TOL
    y = ops.linear_fwd(g32(x), g32(w), g32(b), relu=True, residual=g32(r))
    assert rel_l2(y.cpu(), torch.relu(x @ w.T + b) + r) < TOL


@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (77, 25, 32), (1000, 1024, 256)])
def test_gemm_dgrad_wgrad(M, N, K):
    from transvae_amd import ops
    dy, w, x = rnd(M, N, seed=5), rnd(N, K, seed=6), rnd(M, K, seed=7)
    dx = ops.linear_dgrad(g32(dy), g32(w))
    assert rel_l2(dx.cpu(), dy @ w) < TOL
    base = rnd(M, K, seed=8)
    acc = g32(base)
    ops.linear_dgrad(g32(dy), g32(w), out=acc, accumulate=True)
    assert rel_l2(acc.cpu(), base + dy @ w) < TOL
    dW, db = ops.linear_wgrad(g32(dy), g32(x))
    assert rel_l2(dW.cpu(), dy.T @ x) < 5e-6
    assert rel_l2(db.cpu(), dy.sum(0)) < 5e-6
    mask = rnd(M, K, seed=9)
    dm = ops.linear_dgrad(g32(dy), g32(w), relumask=g32(mask), alpha=1.25)
    assert rel_l2(dm.cpu(), (dy @ w) * 1.25 * (mask > 0)) < TOL


@pytest.mark.parametrize("M,N,K", [(1000, 160, 512), (333, 148, 768), (128, 160, 256), (2000, 132, 1040)])
def test_gemm_wide_tile_nt(M, N, K, monkeypatch):
    """N in (128, 160] goes to the 128 x 160 tile kernel (gemm_wide.hip): bias / ReLU / alpha, ragged M and N, and
    overlapping A rows (lda < K: the conv bottleneck's im2col-free operand)"""
    from transvae_amd import ops
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    y = ops.linear_fwd(g32(x), g32(w), g32(b), relu=True)
    assert rel_l2(y.cpu(), torch.relu(x @ w.T + b)) < TOL
    y = ops.linear_fwd(g32(x), g32(w), None)
    assert rel_l2(y.cpu(), x @ w.T) < TOL
    # overlapping rows: row r of A is flat[r*64 : r*64 + K]
    flat = rnd(M * 64 + K, seed=4)
    A = torch.stack([flat[r * 64:r * 64 + K] for r in range(M)])
    out = torch.empty((M, N), device=dev(), dtype=torch.float32)
    fg = g32(flat)
    ops.gemm(fg, 64, g32(w), K, out, N, M, N, K, bias=g32(b), alpha=0.5)
    assert rel_l2(out.cpu(), 0.5 * (A @ w.T) + b) < TOL


@pytest.mark.parametrize("M,N,K", [(1000, 148, 512), (300, 200, 1024), (640, 160, 768)])
def test_gemm_output_dropout_survives_every_dispatch(M, N, K):
    """Output dropout is signalled by (rng, p) alone, not by a flag.  Launches that would otherwise take the wide-tile kernel
    (N in (128, 160]) or the tail-balance K split (a pure product of few tiles with K >= 512) -- neither can drop -- must
    stay on the kernel that does: ~10 % of the outputs are zero, the rest are the plain product times 1 / P(keep), and split-K
    with output dropout is refused."""
    from transvae_amd import ops, _lib
    x, w = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=0.1)
    drop = ops.Drop(0.1, 17, ops.rng_state(dev()))
    plain = ops.linear_fwd(g32(x), g32(w), None)
    dropped = ops.linear_fwd(g32(x), g32(w), None, drop=drop)
    kept = dropped != 0
    frac = 1.0 - float(kept.float().mean())
    assert 0.08 < frac < 0.12, frac
    scale = float((dropped[kept] / plain[kept]).median())     # 1 / P(keep), P(keep) a multiple of 2^-16: 1.111119
    assert abs(scale * 0.9 - 1.0) < 2e-5, scale
    assert rel_l2(dropped[kept].cpu(), (plain[kept] * scale).cpu()) < TOL
    out = torch.zeros((M, N), device=dev(), dtype=torch.float32)
    with pytest.raises(_lib.TvaeError):
        ops.gemm(g32(x), K, g32(w), K, out, N, M, N, K, flags=ops.GEMM_SPLITK, drop=drop)


@pytest.mark.parametrize("M,N,K", [(2304, 160, 9999), (260, 148, 4100), (128, 160, 300)])
def test_gemm_wide_tile_tn_splitk(M, N, K):
    "the weight-gradient form on the wide tile: both operands k-major, split-K atomics, a partial last k-step"
    from transvae_amd import ops
    a, b = rnd(K, M, seed=5), rnd(K, N, seed=6)
    out = torch.zeros((M, N), device=dev(), dtype=torch.float32)
    ops.gemm(g32(a), M, g32(b), N, out, N, M, N, K, flags=ops.GEMM_A_KMAJOR | ops.GEMM_B_KMAJOR | ops.GEMM_SPLITK)
    assert rel_l2(out.cpu(), a.T @ b) < 5e-6
    out2 = torch.empty((M, N), device=dev(), dtype=torch.float32)
    ops.gemm(g32(a), M, g32(b), N, out2, N, M, N, K, flags=ops.GEMM_A_KMAJOR | ops.GEMM_B_KMAJOR)
    assert rel_l2(out2.cpu(), a.T @ b) < 5e-6


def test_gemm_splitk_large_k():
    from transvae_amd import ops
    M, N, K = 20000, 160, 96          # wgrad shape: reduction over 20000 rows
    dy, x = rnd(M, N, seed=10, scale=0.1), rnd(M, K, seed=11)
    dW, db = ops.linear_wgrad(g32(dy), g32(x))
    assert rel_l2(dW.cpu(), dy.T @ x) < 5e-6


def test_gemm_bf16_mode():
    from transvae_amd import ops, _lib
    M, N, K = 512, 256, 256
    x, w = rnd(M, K, seed=12), rnd(N, K, seed=13)
    out = torch.empty((M, N), device=dev(), dtype=torch.float32)
    ops.gemm(g32(x), K, g32(w), K, out, N, M, N, K, dtype=_lib.BF16)
    assert rel_l2(out.cpu(), x @ w.T) < 1e-2
    xb, wb = x.float().bfloat16().double(), w.float().bfloat16().double()
    assert rel_l2(out.cpu(), xb @ wb.T) < 1e-5      # exact products of the rounded operands
    dy = rnd(M, N, seed=14)
    dxo = torch.empty((M, K), device=dev(), dtype=torch.float32)
    ops.gemm(g32(dy), N, g32(w), K, dxo, K, M, K, N, flags=_lib.GEMM_B_KMAJOR, dtype=_lib.BF16)
    assert rel_l2(dxo.cpu(), dy.float().bfloat16().double() @ wb) < 1e-5
    dW = torch.zeros((N, K), device=dev(), dtype=torch.float32)
    ops.gemm(g32(dy), N, g32(x), K, dW, K, N, K, M,
             flags=_lib.GEMM_A_KMAJOR | _lib.GEMM_B_KMAJOR | _lib.GEMM_SPLITK, dtype=_lib.BF16)
    assert rel_l2(dW.cpu(), dy.float().bfloat16().double().T @ xb) < 1e-5


@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (77, 25, 30), (1000, 131, 36), (63, 5, 129)])
def test_gemm_bf16_kmajor_ragged(M, N, K):
    """bf16 mode, k-major operands (dgrad / wgrad forms: [k][row] staging + transposed LDS reads) on
    shapes that leave partial tiles in every dimension and unaligned rows."""
    from transvae_amd import ops, _lib
    dy, w, x = rnd(M, N, seed=15), rnd(N, K, seed=16), rnd(M, K, seed=17)
    bf = lambda t: t.float().bfloat16().double()
    dxo = torch.empty((M, K), device=dev(), dtype=torch.float32)
    ops.gemm(g32(dy), N, g32(w), K, dxo, K, M, K, N, flags=_lib.GEMM_B_KMAJOR, dtype=_lib.BF16)
    assert rel_l2(dxo.cpu(), bf(dy) @ bf(w)) < 1e-5
    dW = torch.zeros((N, K), device=dev(), dtype=torch.float32)
    db = torch.zeros((N,), device=dev(), dtype=torch.float32)
    ops.gemm(g32(dy), N, g32(x), K, dW, K, N, K, M,
             flags=_lib.GEMM_A_KMAJOR | _lib.GEMM_B_KMAJOR | _lib.GEMM_SPLITK, dtype=_lib.BF16, a_rowsum=db)
    assert rel_l2(dW.cpu(), bf(dy).T @ bf(x)) < 1e-5
    assert rel_l2(db.cpu(), dy.sum(0)) < 5e-6           # the fused bias gradient sums the unrounded values


def test_gemm_rejects_bad_shapes():
    from transvae_amd import ops, _lib
    a = torch.zeros((4, 4), device=dev())
    with pytest.raises(_lib.TvaeError):
        ops.gemm(a, 4, a, 4, a, 2, 4, 4, 4)        # ldc < N
    with pytest.raises(_lib.TvaeError):
        ops.gemm(a, 4, a, 4, a, 4, 0, 4, 4)        # empty


# ------------------------------------------------------------------------------------ LayerNorm
def _ln_ref(x, a, b, eps=1e-6):
    return a * (x - x.mean(-1, keepdim=True)) / (x.std(-1, keepdim=True) + eps) + b


@pytest.mark.parametrize("d", [128, 256, 192])
def test_layernorm_fwd_bwd(d):
    from transvae_amd import ops
    rows = 1003
    x, a, b, dy, res = rnd(rows, d, seed=1, scale=2.0), 1 + 0.1 * rnd(d, seed=2), rnd(d, seed=3), rnd(rows, d, seed=4), rnd(rows, d, seed=5)
    y = ops.layernorm_fwd(g32(x), g32(a), g32(b))
    assert rel_l2(y.cpu(), _ln_ref(x, a, b)) < TOL
    xr, ar, br = x.clone().requires_grad_(True), a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    _ln_ref(xr, ar, br).backward(dy)
    dx, da, db = ops.layernorm_bwd(g32(x), g32(a), g32(dy), add_to=g32(res))
    assert rel_l2(dx.cpu(), xr.grad + res) < 3e-6
    assert rel_l2(da.cpu(), ar.grad) < 5e-6
    assert rel_l2(db.cpu(), br.grad) < 5e-6


# ------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, h, mask):
    B, Lq, d = q.shape
    dk = d // h
    qh, kh, vh = (t.view(B, -1, h, dk).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-2, -1) / math.sqrt(dk)
    if mask is not None:
        s = s.masked_fill(mask.unsqueeze(1) == 0, -1e9)
    p = F.softmax(s, dim=-1)
    return (p @ vh).transpose(1, 2).contiguous().view(B, Lq, d), p


def _masks(kind, B, Lq, Lk, seed=0):
    from transvae_amd import ops
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(0, Lk + 1, (B,), generator=g)
    lens[0] = 0            # fully masked molecule -> uniform attention (reference's -1e9 fill)
    lens[1] = Lk
    km = (torch.arange(Lk)[None, :] < lens[:, None])
    if kind == "none":
        return None, ops.MaskSpec()
    if kind == "key":
        return km.unsqueeze(1), ops.MaskSpec(key_mask=km.to(torch.uint8).to(dev()))
    if kind == "causal":
        m = km.unsqueeze(1) & torch.tril(torch.ones(1, Lq, Lk, dtype=torch.bool))
        return m, ops.MaskSpec(key_mask=km.to(torch.uint8).to(dev()), causal=True)
    if kind == "full":
        m = torch.rand(B, Lq, Lk, generator=g) > 0.3
        m[2, 5, :] = False
        return m, ops.MaskSpec(full=m.to(torch.uint8).to(dev()))
    raise ValueError(kind)


@pytest.mark.parametrize("dk", [32, 64])
@pytest.mark.parametrize("kind,Lq,Lk", [("none", 127, 127), ("key", 127, 127), ("causal", 126, 126),
                                        ("key", 126, 127), ("full", 50, 77), ("causal", 1, 1), ("key", 128, 128)])
def test_attention_fwd_bwd(dk, kind, Lq, Lk):
    from transvae_amd import ops
    B, h = 5, 4
    d = h * dk
    q, k, v, do = rnd(B, Lq, d, seed=1), rnd(B, Lk, d, seed=2), rnd(B, Lk, d, seed=3), rnd(B, Lq, d, seed=4)
    mask, spec = _masks(kind, B, Lq, Lk)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    o_ref, p_ref = _attn_ref(qr, kr, vr, h, mask)
    o_ref.backward(do)
    o, lse, p = ops.attn_fwd(g32(q).view(-1, d), g32(k).view(-1, d), g32(v).view(-1, d), B, h, Lq, Lk, dk, spec, want_p=True)
    assert rel_l2(o.cpu(), o_ref.detach().reshape(-1, d)) < TOL
    assert rel_l2(p.cpu(), p_ref.detach()) < TOL
    dq, dkk, dv = ops.attn_bwd(g32(q).view(-1, d), g32(k).view(-1, d), g32(v).view(-1, d), o, g32(do).view(-1, d), lse,
                               B, h, Lq, Lk, dk, spec)
    assert rel_l2(dq.cpu(), qr.grad.reshape(-1, d)) < 5e-6
    assert rel_l2(dkk.cpu(), kr.grad.reshape(-1, d)) < 5e-6
    assert rel_l2(dv.cpu(), vr.grad.reshape(-1, d)) < 5e-6


@pytest.mark.parametrize("dk", [32, 64])
@pytest.mark.parametrize("kind,Lq,Lk", [("none", 127, 127), ("key", 127, 127), ("causal", 126, 126), ("full", 50, 77), ("key", 128, 128)])
def test_attention_bf16_mode(dk, kind, Lq, Lk):
    """dtype TVAE_BF16: QK^T, PV, dP and the dQ / dK / dV products on v_mfma_f32_32x32x16_bf16 with operands rounded to bf16
    on the way in; softmax, statistics and the -1e9 semantics stay fp32.  north_star's bf16 tolerance: 1e-2."""
    from transvae_amd import ops
    B, h = 5, 4
    d = h * dk
    q, k, v, do = rnd(B, Lq, d, seed=1), rnd(B, Lk, d, seed=2), rnd(B, Lk, d, seed=3), rnd(B, Lq, d, seed=4)
    mask, spec = _masks(kind, B, Lq, Lk)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    o_ref, p_ref = _attn_ref(qr, kr, vr, h, mask)
    o_ref.backward(do)
    prev = ops.compute_dtype()
    ops.set_compute_dtype("bf16")
    try:
        o, lse, p = ops.attn_fwd(g32(q).view(-1, d), g32(k).view(-1, d), g32(v).view(-1, d), B, h, Lq, Lk, dk, spec, want_p=True)
        dq, dkk, dv = ops.attn_bwd(g32(q).view(-1, d), g32(k).view(-1, d), g32(v).view(-1, d), o, g32(do).view(-1, d), lse,
                                   B, h, Lq, Lk, dk, spec)
    finally:
        ops.set_compute_dtype(prev)
    errs = {"o": rel_l2(o.cpu(), o_ref.detach().reshape(-1, d)), "p": rel_l2(p.cpu(), p_ref.detach()),
            "dq": rel_l2(dq.cpu(), qr.grad.reshape(-1, d)), "dk": rel_l2(dkk.cpu(), kr.grad.reshape(-1, d)),
            "dv": rel_l2(dv.cpu(), vr.grad.reshape(-1, d))}
    print({n: "%.1e" % e for n, e in errs.items()})
    assert all(1e-5 < e < 1e-2 for e in errs.values()), errs      # above 1e-5: the bf16 kernels did run


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("causal", [False, True])
def test_attention_dead_tiles_exact(causal, mode):
    """Key tiles whose mask words are all zero are skipped (their weights are exact zeros).  Lengths on
    every 32-key tile boundary, with and without the causal mask, dropout on: (a) against the oracle
    math; (b) the values of masked keys must not matter at all -- bit-identical outputs and gradients
    when their K / V rows are replaced by large finite junk; (c) dK / dV equal to the
    full-mask instantiation's, whose dK/dV kernel skips nothing (bit-identical where the waves are not re-dealt).  Molecule 0 is fully masked (uniform
    attention over masked keys: nothing may be skipped there)."""
    from transvae_amd import ops
    prev = ops.compute_dtype()
    ops.set_compute_dtype(mode)
    try:
        _dead_tiles_body(causal, TOL if mode == "fp32" else 1e-2, 5e-6 if mode == "fp32" else 1e-2)
    finally:
        ops.set_compute_dtype(prev)


def _dead_tiles_body(causal, tol_o, tol_g):
    from transvae_amd import ops
    B, h, dk, L = 8, 4, 64, 127
    d = h * dk
    lens = torch.tensor([0, L, 1, 3, 32, 33, 64, 97])
    km = (torch.arange(L)[None, :] < lens[:, None])
    q, k, v, do = rnd(B, L, d, seed=21), rnd(B, L, d, seed=22), rnd(B, L, d, seed=23), rnd(B, L, d, seed=24)
    mask = km.unsqueeze(1) & (torch.tril(torch.ones(1, L, L, dtype=torch.bool)) if causal else torch.ones(1, L, L, dtype=torch.bool))
    spec = ops.MaskSpec(key_mask=km.to(torch.uint8).to(dev()), causal=causal)
    spec_full = ops.MaskSpec(full=mask.expand(B, L, L).contiguous().to(torch.uint8).to(dev()))
    flat = lambda t: g32(t).view(-1, d)

    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    o_ref, _ = _attn_ref(qr, kr, vr, h, mask)
    o_ref.backward(do)
    o, lse, _ = ops.attn_fwd(flat(q), flat(k), flat(v), B, h, L, L, dk, spec)
    dq, dkk, dv = ops.attn_bwd(flat(q), flat(k), flat(v), o, flat(do), lse, B, h, L, L, dk, spec)
    assert rel_l2(o.cpu(), o_ref.detach().reshape(-1, d)) < tol_o
    assert rel_l2(dq.cpu(), qr.grad.reshape(-1, d)) < tol_g
    assert rel_l2(dkk.cpu(), kr.grad.reshape(-1, d)) < tol_g
    assert rel_l2(dv.cpu(), vr.grad.reshape(-1, d)) < tol_g

    of, lsef, _ = ops.attn_fwd(flat(q), flat(k), flat(v), B, h, L, L, dk, spec_full)
    dqf, dkf, dvf = ops.attn_bwd(flat(q), flat(k), flat(v), of, flat(do), lsef, B, h, L, L, dk, spec_full)
    assert torch.equal(o, of) and torch.equal(dq, dqf)
    # dK / dV: with 1 or 2 live key tiles the waves of the key-masked instantiation share a key tile and add their partial
    # sums over the query tiles in another grouping than the full-mask instantiation's single chain: same terms, rounding apart
    assert rel_l2(dkk.cpu(), dkf.cpu()) < (1e-6 if tol_g < 1e-3 else 1e-2) and rel_l2(dv.cpu(), dvf.cpu()) < (1e-6 if tol_g < 1e-3 else 1e-2)
    full_rows = (lens[:, None] > 64).expand(B, L).reshape(-1).to(dev())        # molecules with 3-4 live tiles: one chain, same bits
    assert torch.equal(dkk[full_rows], dkf[full_rows]) and torch.equal(dv[full_rows], dvf[full_rows])

    # junk in the masked keys (molecule 0 excluded: its masked keys DO carry weight), dropout on
    ops.seed_rng(dev(), 5)
    drop = ops.Drop(0.1, 9, ops.rng_state(dev()))
    junk = (~km)[:, :, None].expand(B, L, d).clone()
    junk[0] = False
    k2, v2 = torch.where(junk, torch.full_like(k, 3.0e4), k), torch.where(junk, torch.full_like(v, -7.0e3), v)
    oa, lsea, _ = ops.attn_fwd(flat(q), flat(k), flat(v), B, h, L, L, dk, spec, drop)
    ob, lseb, _ = ops.attn_fwd(flat(q), flat(k2), flat(v2), B, h, L, L, dk, spec, drop)
    assert torch.equal(oa, ob)
    ga = ops.attn_bwd(flat(q), flat(k), flat(v), oa, flat(do), lsea, B, h, L, L, dk, spec, drop)
    gb = ops.attn_bwd(flat(q), flat(k2), flat(v2), ob, flat(do), lseb, B, h, L, L, dk, spec, drop)
    assert torch.equal(ga[0], gb[0])
    live = (~junk).reshape(-1, d).to(dev())
    assert torch.equal(ga[1] * live, gb[1] * live) and torch.equal(ga[2] * live, gb[2] * live)


def test_attention_strided_qkv():
    """q, k, v as column blocks of one (rows, 3d) buffer (fused QKV projection layout)."""
    from transvae_amd import ops
    B, h, dk, L = 3, 4, 64, 127
    d = h * dk
    qkv = rnd(B * L, 3 * d, seed=7)
    buf = g32(qkv)
    spec = ops.MaskSpec()
    o, _, _ = ops.attn_fwd(buf[:, :d], buf[:, d:2 * d], buf[:, 2 * d:], B, h, L, L, dk, spec)
    o_ref, _ = _attn_ref(qkv[:, :d].reshape(B, L, d), qkv[:, d:2 * d].reshape(B, L, d), qkv[:, 2 * d:].reshape(B, L, d), h, None)
    assert rel_l2(o.cpu(), o_ref.reshape(-1, d)) < TOL


def test_attention_dropout_statistics_and_backward_consistency():
    """Dropout cannot match torch's Philox stream; check (a) keep-rate, (b) that the backward
    regenerates exactly the forward's mask: with v = identity-like probes the output reveals the mask."""
    from transvae_amd import ops
    B, h, dk, L = 4, 4, 32, 64
    d = h * dk
    ops.seed_rng(dev(), 123)
    drop = ops.Drop(0.1, 77, ops.rng_state(dev()))
    q, k = rnd(B, L, d, seed=1, scale=0.01), rnd(B, L, d, seed=2, scale=0.01)
    v = torch.ones(B, L, d, dtype=torch.float64)
    spec = ops.MaskSpec()
    o, lse, _ = ops.attn_fwd(g32(q).view(-1, d), g32(k).view(-1, d), g32(v).view(-1, d), B, h, L, L, dk, spec, drop)
    # rows of P sum to 1, v = 1  =>  o = (#kept / L) / 0.9 approximately (P nearly uniform)
    keep = o.mean().item() * 0.9
    assert abs(keep - 0.9) < 0.01
    o2, _, _ = ops.attn_fwd(g32(q).view(-1, d), g32(k).view(-1, d), g32(v).view(-1, d), B, h, L, L, dk, spec, drop)
    assert torch.equal(o, o2)          # same (seed, step, site) -> same mask
    ops.advance_rng(dev())
    o3, _, _ = ops.attn_fwd(g32(q).view(-1, d), g32(k).view(-1, d), g32(v).view(-1, d), B, h, L, L, dk, spec, drop)
    assert not torch.equal(o, o3)      # step counter changes the mask


# ------------------------------------------------------------------------------------ embedding
def test_embed_fwd_bwd():
    from oracle import tvae_oracle as O
    from transvae_amd import ops
    B, L, d, V = 7, 127, 256, 26
    ids = torch.randint(0, V, (B, L), generator=torch.Generator().manual_seed(0))
    lut = rnd(V, d, seed=1)
    pe = O.positional_table(5000, d)
    out = ops.EmbedFn.apply(ids.to(dev()), g32(lut), pe.to(dev()), None)
    ref = lut[ids] * math.sqrt(d) + pe[:L].double()
    assert rel_l2(out.cpu(), ref) < TOL
    dout = rnd(B, L, d, seed=2)
    dlut = torch.zeros(V, d, device=dev())
    from transvae_amd.ops import _p, _stream, call
    call("tvae_embed_bwd", _p(ids.to(dev())), _p(g32(dout)), _p(dlut), B, L, d, V, None, 0, 0.0, _stream())
    ref_g = torch.zeros(V, d, dtype=torch.float64).index_add_(0, ids.view(-1), dout.view(-1, d)) * math.sqrt(d)
    assert rel_l2(dlut.cpu(), ref_g) < 5e-6


# ------------------------------------------------------------------------------------ conv / deconv
def _conv_ref(x, Ws, bs):
    """fp64 ConvBottleneck + the smallest gate margin (max-pool pair gap, ReLU input magnitude)."""
    y, margin = x.permute(0, 2, 1), float("inf")
    for W, b in zip(Ws, bs):
        c = F.conv1d(y, W, b)
        L = c.shape[-1] // 2 * 2
        u, v = c[..., 0:L:2], c[..., 1:L:2]
        margin = min(margin, float((u - v).abs().min()), float(torch.maximum(u, v).abs().min()))
        y = F.relu(F.max_pool1d(c, 2))
    return y.contiguous().view(x.shape[0], -1), margin


def _deconv_ref(h, Ws, bs):
    y, margin = h.view(h.shape[0], 64, 9), float("inf")
    for i, s in enumerate((4, 3, 1)):
        c = F.conv_transpose1d(y, Ws[i], bs[i], stride=s, padding=2)
        margin = min(margin, float(c.abs().min()))
        y = F.relu(c)
    return y.permute(0, 2, 1), margin


def _gate_safe(make, ref, need=1e-5, tries=200):
    """ReLU / MaxPool gates make the gradient discontinuous; pick seeded inputs whose gates are all
    further than `need` from switching, so fp32-vs-fp64 rounding cannot flip one."""
    best = None
    for seed in range(tries):
        args = make(seed)
        with torch.no_grad():
            _, margin = ref(*args)
        if best is None or margin > best[0]:
            best = (margin, args)
        if margin >= need:
            break
    return best[1]


@pytest.mark.parametrize("d", [128, 256])
def test_conv_bottleneck_fn(d):
    from transvae_amd import ops
    B, L = 2, 127
    chans = [d]
    for i in range(3):
        chans.append(64 if i == 2 else (chans[-1] - 64) // 2 + 64)
    ks = [9, 8, 8]

    def make(seed):
        return (rnd(B, L, d, seed=1000 + seed),
                [rnd(chans[i + 1], chans[i], ks[i], seed=10 + i, scale=0.05) for i in range(3)],
                [rnd(chans[i + 1], seed=20 + i, scale=0.1) for i in range(3)])

    x, Ws, bs = _gate_safe(make, _conv_ref)
    xr = x.clone().requires_grad_(True)
    Wr = [w.clone().requires_grad_(True) for w in Ws]
    br = [b.clone().requires_grad_(True) for b in bs]
    ref, _ = _conv_ref(xr, Wr, br)
    dout = rnd(B, 576, seed=30)
    ref.backward(dout)
    xs = g32(x).requires_grad_(True)
    params = []
    for i in range(3):
        params += [g32(Ws[i]).requires_grad_(True), g32(bs[i]).requires_grad_(True)]
    out = ops.ConvBottleneckFn.apply(xs, *params)
    assert out.shape == (B, 576)
    assert rel_l2(out.detach().cpu(), ref.detach()) < TOL
    out.backward(g32(dout))
    assert rel_l2(xs.grad.cpu(), xr.grad) < 5e-6
    for i in range(3):
        assert rel_l2(params[2 * i].grad.cpu(), Wr[i].grad) < 5e-6, i
        assert rel_l2(params[2 * i + 1].grad.cpu(), br[i].grad) < 5e-6, i


@pytest.mark.parametrize("d", [128, 256])
def test_deconv_bottleneck_fn(d):
    from transvae_amd import ops
    B = 2
    chans = [64]
    for i in range(3):
        chans.append(d if i == 2 else (d - chans[-1]) // 4 + chans[-1])

    def make(seed):
        return (rnd(B, 576, seed=1000 + seed).abs(),
                [rnd(chans[i], chans[i + 1], 11, seed=10 + i, scale=0.05) for i in range(3)],
                [rnd(chans[i + 1], seed=20 + i, scale=0.1) for i in range(3)])

    hl, Ws, bs = _gate_safe(make, _deconv_ref)
    hr = hl.clone().requires_grad_(True)
    Wr = [w.clone().requires_grad_(True) for w in Ws]
    br = [b.clone().requires_grad_(True) for b in bs]
    ref, _ = _deconv_ref(hr, Wr, br)
    assert ref.shape == (B, 127, d)
    dout = rnd(B, 127, d, seed=30)
    ref.backward(dout)
    hs = g32(hl).requires_grad_(True)
    params = []
    for i in range(3):
        params += [g32(Ws[i]).requires_grad_(True), g32(bs[i]).requires_grad_(True)]
    out = ops.DeconvBottleneckFn.apply(hs, *params)
    assert rel_l2(out.detach().cpu(), ref.detach()) < TOL
    out.backward(g32(dout))
    assert rel_l2(hs.grad.cpu(), hr.grad) < 5e-6
    for i in range(3):
        assert rel_l2(params[2 * i].grad.cpu(), Wr[i].grad) < 5e-6, i
        assert rel_l2(params[2 * i + 1].grad.cpu(), br[i].grad) < 5e-6, i


# ------------------------------------------------------------------------------------ loss / Adam
def test_loss_kernel_against_reference_golden():
    from helpers import load
    from transvae_amd import ops, synth
    g = load("units")
    toks = torch.from_numpy(g["tokens"].astype(np.int64)).to(dev())
    w = torch.tensor(synth.load_vocab("zinc")[2], dtype=torch.float32, device=dev())
    for tag, mu_key, prop in (("plain", "mu", False), ("prop", "mu", True), ("nan", "mu_nan", False)):
        t = {k: torch.from_numpy(g[k]).to(dev()) for k in ("logits", mu_key, "logvar", "pred_len", "pred_prop", "true_prop")}
        out5, gr = ops.loss_fwd_bwd(toks, t["logits"], w, t["pred_len"], t[mu_key], t["logvar"],
                                    t["pred_prop"] if prop else None, t["true_prop"], float(g["beta"]), 24)
        np.testing.assert_allclose(out5.cpu().numpy(), g["loss_" + tag], rtol=3e-6, atol=1e-7)
        assert rel_l2(gr["logits"].cpu().view(5, 126, 25), g["dlogits_" + tag]) < 5e-6
        assert rel_l2(gr["pred_len"].cpu(), g["dpred_len_" + tag]) < 5e-6
        if tag == "nan":
            assert float(gr["mu"].abs().max()) == 0.0 and float(gr["logvar"].abs().max()) == 0.0
        else:
            assert rel_l2(gr["mu"].cpu(), g["dmu_" + tag]) < 5e-6
            assert rel_l2(gr["logvar"].cpu(), g["dlogvar_" + tag]) < 5e-6


def test_fused_adam_matches_torch_adam():
    from transvae_amd import ops
    shapes = [(25,), (300, 17), (4096,), (5000,), (1,)]
    ps = [rnd(*s, seed=i).float() for i, s in enumerate(shapes)]
    ref_p = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.Adam(ref_p, lr=0.0, betas=(0.9, 0.98), eps=1e-9)
    mine = [p.clone().to(dev()).requires_grad_(True) for p in ps]
    fa = ops.FusedAdam(mine)
    for step in range(1, 4):
        lr = 1e-3 * step
        for i, (rp, mp) in enumerate(zip(ref_p, mine)):
            gstep = rnd(*shapes[i], seed=100 + 10 * step + i).float()
            rp.grad = gstep.clone()
            mp.grad = gstep.clone().to(dev()) if mp.grad is None else mp.grad.copy_(gstep.to(dev()))
        for gq in opt.param_groups:
            gq["lr"] = lr
        opt.step()
        fa.set_hyper(lr, step)
        fa.launch(zero_grad=True)
        torch.cuda.synchronize()
        for rp, mp, p0 in zip(ref_p, mine, ps):
            assert rel_l2((mp.detach().cpu() - p0), (rp.detach() - p0)) < 1e-5
            assert float(mp.grad.abs().max()) == 0.0


# ------------------------------------------------------------------------------------ fused sublayers
def test_sublayer_functions_match_oracle():
    from oracle import tvae_oracle as O
    from helpers import oracle_params
    from transvae_amd import ops
    P = oracle_params("4x256_zinc", torch.float64, requires_grad=True)
    B, L, d, h = 4, 127, 256, 4
    x = rnd(B, L, d, seed=1)
    lens = torch.tensor([5, 127, 60, 0])
    mask = (torch.arange(L)[None, :] < lens[:, None]).unsqueeze(1)
    pre = "encoder.layers.1"
    xr = x.clone().requires_grad_(True)
    ref = O.encoder_layer(P, pre, xr, mask, h)
    dout = rnd(B, L, d, seed=2)
    ref.backward(dout)
    names = ["sublayer.0.norm.a_2", "sublayer.0.norm.b_2"] + ["self_attn.linears.%d.%s" % (i, wb) for i in range(4) for wb in ("weight", "bias")]
    prm = [g32(P[pre + "." + n].detach()).requires_grad_(True) for n in names]
    xs = g32(x).requires_grad_(True)
    spec = ops.MaskSpec(key_mask=mask[:, 0].to(torch.uint8).to(dev()))
    y1 = ops.AttnSublayerFn.apply(xs, None, *prm, spec, h, None, None, 1e-6)
    names2 = ["sublayer.1.norm.a_2", "sublayer.1.norm.b_2", "feed_forward.w_1.weight", "feed_forward.w_1.bias",
              "feed_forward.w_2.weight", "feed_forward.w_2.bias"]
    prm2 = [g32(P[pre + "." + n].detach()).requires_grad_(True) for n in names2]
    y2 = ops.FFNSublayerFn.apply(y1, *prm2, None, None, 1e-6)
    assert rel_l2(y2.detach().cpu(), ref.detach()) < TOL
    y2.backward(g32(dout))
    assert rel_l2(xs.grad.cpu(), xr.grad) < 5e-6
    for n, t in zip(names + names2, prm + prm2):
        gref = P[pre + "." + n].grad
        if n.endswith("linears.1.bias"):          # analytically zero (softmax shift invariance)
            assert float(t.grad.abs().max()) < 1e-4 * float(dout.abs().max())
        else:
            assert rel_l2(t.grad.cpu(), gref) < 1e-5, n


# ---- fp32-accurate GEMM on the bf16 matrix cores (dtype f32x3) -------------------------------------
@pytest.mark.parametrize("presplit", [False, True])
@pytest.mark.parametrize("M,N,K", [(777, 256, 256), (1500, 1024, 256), (640, 256, 1024), (300, 260, 2304), (130, 132, 40),
                                   (500, 144, 272), (129, 208, 48), (200, 64, 16), (200, 64, 32), (257, 768, 256)])
def test_gemm_f32x3_is_as_accurate_as_fp32(M, N, K, presplit):
    """Every GEMM role (forward with bias/ReLU/residual, data gradient, weight gradient with the fused
    bias gradient) in 3xbf16-split arithmetic against an fp64 product: the error must be at the level
    of the exact-fp32 kernel's (both are printed), far from bf16's 2e-3."""
    from transvae_amd import ops
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device=dev())
    w = torch.randn(N, K, device=dev()) * 0.05
    if presplit:            # an nn.Parameter weight is split once (tvae_split3) and reaches the kernel as bf16 planes
        w = torch.nn.Parameter(w, requires_grad=False)
    b = torch.randn(N, device=dev())
    res = torch.randn(M, N, device=dev())
    dy = torch.randn(M, N, device=dev())
    ref_f = torch.relu(x.double() @ w.double().t() + b.double()) + res.double()
    ref_d = dy.double() @ w.double()
    ref_w = dy.double().t() @ x.double()
    ref_b = dy.double().sum(0)
    errs = {}
    prev = ops.compute_dtype()
    try:
        for mode in ("fp32", "f32x3"):
            ops.set_compute_dtype(mode)
            yf = ops.linear_fwd(x, w, b, relu=True, residual=res)
            yd = ops.linear_dgrad(dy, w)
            yw, yb = ops.linear_wgrad(dy, x)
            errs[mode] = [rel_l2(yf.cpu(), ref_f.cpu()), rel_l2(yd.cpu(), ref_d.cpu()), rel_l2(yw.cpu(), ref_w.cpu()),
                          rel_l2(yb.cpu(), ref_b.cpu())]
    finally:
        ops.set_compute_dtype(prev)
    print("rel-L2 vs fp64 (fwd, dgrad, wgrad, bias grad): fp32 %s  f32x3 %s" %
          (["%.1e" % e for e in errs["fp32"]], ["%.1e" % e for e in errs["f32x3"]]))
    for e3, e1 in zip(errs["f32x3"], errs["fp32"]):
        assert e3 < max(2.0 * e1, 3e-7), (errs)


@pytest.mark.parametrize("M,N,K", [(777, 256, 256), (500, 144, 272), (1030, 384, 64)])
def test_gemm_f32x3_streamed_kernel_epilogues_match_the_fp32_kernel(M, N, K):
    """The streamed f32x3 kernel (gemm3_dma.hip: pre-split nn.Parameter weight on the B side) has its own epilogue for its
    row-wise wave layout.  Every fused form the model uses -- bias + ReLU + dropout, bias + dropout + residual, the data
    gradient through a ReLU mask with alpha, the data gradient accumulated into its output -- against the exact-fp32 kernel
    with the same flags (the dropout mask is a hash of the element index: identical), interior and edge tiles."""
    from transvae_amd import ops
    torch.manual_seed(M + N)
    x = torch.randn(M, K, device=dev())
    w = torch.nn.Parameter(torch.randn(N, K, device=dev()) * 0.05, requires_grad=False)
    b = torch.randn(N, device=dev())
    res = torch.randn(M, N, device=dev())
    dy = torch.randn(M, N, device=dev())
    saved = torch.randn(M, K, device=dev())
    base = torch.randn(M, K, device=dev())
    drop = ops.Drop(0.1, 21, ops.rng_state(dev()))
    outs = {}
    prev = ops.compute_dtype()
    try:
        for mode in ("fp32", "f32x3"):
            ops.set_compute_dtype(mode)
            if mode == "f32x3":
                assert ops._split_weight(w) is not None              # the streamed kernel's precondition
            acc = base.clone()
            ops.linear_dgrad(dy, w, out=acc, accumulate=True)
            outs[mode] = [ops.linear_fwd(x, w, b, relu=True, drop=drop), ops.linear_fwd(x, w, b, residual=res, drop=drop),
                          ops.linear_fwd(x, w, None), ops.linear_dgrad(dy, w, relumask=saved, alpha=1.25), acc]
    finally:
        ops.set_compute_dtype(prev)
    for a, c in zip(outs["f32x3"], outs["fp32"]):
        assert torch.equal(a == 0, c == 0)                           # the same elements dropped / masked
        assert rel_l2(a.cpu(), c.cpu()) < 1e-6


# ---- streamed bf16 NT kernel (LDS-DMA, weight bf16 in memory) --------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(777, 256, 256), (1500, 1024, 256), (640, 256, 1024), (130, 136, 64), (12700, 768, 256),
                                   (50, 25 * 8, 32)])
def test_streamed_bf16_gemm_equals_the_staged_bf16_gemm(M, N, K):
    """Forward (bias / ReLU / residual / accumulate) and data gradient through gemm_dma.hip (nn.Parameter weight ->
    bf16 copies of W and W^T, TVAE_GEMM_B_BF16) against (a) the register-staged bf16 kernel on the same operands (a
    plain-tensor weight takes that path) and (b) an fp64 product of the bf16-rounded operands."""
    from transvae_amd import ops
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device=dev())
    w = torch.randn(N, K, device=dev()) * 0.05
    wp = torch.nn.Parameter(w.clone(), requires_grad=False)
    b = torch.randn(N, device=dev())
    res = torch.randn(M, N, device=dev())
    dy = torch.randn(M, N, device=dev())
    rb = lambda t: t.to(torch.bfloat16).double()
    ref_f = torch.relu(rb(x) @ rb(w).t() + b.double()) + res.double()
    ref_d = rb(dy) @ rb(w)
    prev = ops.compute_dtype()
    try:
        ops.set_compute_dtype("bf16")
        assert ops._bf16_weight(wp) is not None and ops._bf16_weight(w) is None
        new_f = ops.linear_fwd(x, wp, b, relu=True, residual=res)
        old_f = ops.linear_fwd(x, w, b, relu=True, residual=res)
        assert rel_l2(new_f.cpu(), ref_f.cpu()) < 2e-6 and rel_l2(new_f.cpu(), old_f.cpu()) < 1e-6
        acc_new = res.clone()
        ops.gemm(x, K, ops._bf16_weight(wp)[0], K, acc_new, N, M, N, K, bias=b, flags=ops.GEMM_ACCUM | ops.GEMM_B_BF16)
        assert rel_l2(acc_new.cpu(), (rb(x) @ rb(w).t() + b.double() + res.double()).cpu()) < 2e-6
        if N % 32 == 0:
            new_d = ops.linear_dgrad(dy, wp)
            old_d = ops.linear_dgrad(dy, w)
            assert rel_l2(new_d.cpu(), ref_d.cpu()) < 2e-6 and rel_l2(new_d.cpu(), old_d.cpu()) < 1e-6
            # a dropped-out output's gradient: the mask is applied by a pre-pass, the result is the staged kernel's
            ops.seed_rng(dev(), 11)
            drop = ops.Drop(0.1, 3, ops.rng_state(dev()))
            assert rel_l2(ops.linear_dgrad(dy, wp, a_drop=drop).cpu(), ops.linear_dgrad(dy, w, a_drop=drop).cpu()) < 1e-6
        # the weight moves (in place, as the optimizer does): the bf16 copies follow
        with torch.no_grad():
            wp.mul_(1.5)
        ops.weights_changed()
        assert rel_l2(ops.linear_fwd(x, wp).cpu(), (rb(x) @ rb(w * 1.5).t()).cpu()) < 2e-6
    finally:
        ops.set_compute_dtype(prev)


def test_streamed_fp32_gemm_switch(monkeypatch):
    "the exact-fp32 instantiation of the streamed kernel (TVAE_GEMM_DMA=1 in a fresh process) against fp64"
    import subprocess, sys
    code = (
        "import torch, sys; sys.path.insert(0, %r)\n"
        "from transvae_amd import ops\n"
        "torch.manual_seed(0)\n"
        "for M, N, K in ((777, 256, 256), (1500, 1024, 256), (640, 256, 1024), (130, 136, 48)):\n"
        "    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * 0.05; b = torch.randn(N, device='cuda')\n"
        "    res = torch.randn(M, N, device='cuda')\n"
        "    y = ops.linear_fwd(x, w, b, relu=True, residual=res)\n"
        "    ref = torch.relu(x.double() @ w.double().t() + b.double()) + res.double()\n"
        "    err = float((y.double() - ref).norm() / ref.norm())\n"
        "    assert err < 5e-7, (M, N, K, err)\n"
        "print('ok')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, TVAE_GEMM_DMA="1"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-1500:]
