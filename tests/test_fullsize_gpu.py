"""Parity at BASELINE.json's full sizes (trans4x-256, batch 500 = 5 x 100, thousands of latents), where the
reference itself cannot be run per test: size-independent properties of the path, plus one full chunk
against the float64 oracle.

  * the one-pass batch of 500 with per-chunk loss normalisation == the reference's literal loop over 5
    chunks of 100 (transvae/trans_models.py:202-236): same per-chunk loss rows (2e-6), same accumulated
    gradients up to gate flips;
  * one full chunk of 100 SMILES-shaped molecules: the five loss terms within 1e-5 of the oracle's float64
    run, gradients at the level a few marginal ReLU / MaxPool gates allow (see _report_and_check);
  * greedy decode is row-wise: decoding a slab of latents at once, in pieces, or in another order gives
    the same token rows bit for bit (what sharding the latents over ranks relies on).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import CONFIGS, oracle_params, rel_l2  # noqa: E402
from test_model_gpu import build  # noqa: E402


def _report_and_check(what, errs):
    """Per-tensor gradient errors on RANDOM inputs of full size.  Unlike the golden inputs these are not chosen
    gate-safe: among the ~10^7 ReLU / MaxPool gates of such a batch a few have margins
    below fp32 resolution and decide differently in any two implementations (the reference on another BLAS
    included).  One flipped MaxPool gate of the conv bottleneck moves every tensor upstream of it (the whole
    encoder) by 1e-4 .. 5e-3 -- measured here: median 5e-6 / 1e-5, worst 3e-4 / 5e-3 -- so the tight
    per-tensor gradient parity lives in the golden tests (gate-safe inputs, tests/test_model_gpu.py) and the
    bars here are what a few flips allow and any wrong kernel (errors of order 1) cannot pass."""
    v = np.array(sorted(errs.values()))
    worst = max(errs, key=errs.get)
    print("%s: gradient rel-L2 median %.2e, 80%% %.2e, worst %.2e (%s), %d/%d tensors within 1e-5"
          % (what, float(np.median(v)), float(v[int(0.8 * len(v))]), errs[worst], worst, int((v < 1e-5).sum()), len(v)))
    assert float(np.median(v)) < 5e-5
    assert float(v[int(0.8 * len(v))]) < 2e-3
    assert errs[worst] < 3e-2, (worst, errs[worst])


def _batch(n, seed):
    from transvae_amd import synth
    toks = synth.synth_tokens(n, seed=seed).float()
    return torch.cat([toks, torch.zeros(n, 1)], 1)


def test_batch500_one_pass_equals_the_chunk_loop_4x256():
    data = _batch(500, seed=100)
    eps = torch.randn(500, 128, generator=torch.Generator().manual_seed(11)).cuda()
    grads, tables = [], []
    for fuse in (True, False):
        vae = build("4x256_zinc", BATCH_SIZE=500, BATCH_CHUNKS=5, FUSE_CHUNKS=fuse)
        vae.model.eval()                                     # dropout off: the two runs must see the same arithmetic
        vae.optimizer.optimizer.fuse_zero_grad = False       # keep the gradients to look at them
        tables.append(vae.train_step(data, 0.02, eps=eps).cpu().numpy())
        grads.append({k: p.grad.detach().cpu().numpy().copy() for k, p in vae.model.named_parameters()})
        del vae
    assert tables[0].shape[0] == 5
    np.testing.assert_allclose(tables[0], tables[1], rtol=2e-6, atol=1e-7)
    errs = {}
    for k in grads[0]:
        if k.endswith("attn.linears.1.bias"):       # analytically zero (softmax shift invariance): noise only
            assert np.linalg.norm(grads[0][k]) < 1e-4 and np.linalg.norm(grads[1][k]) < 1e-4
        else:
            errs[k] = rel_l2(grads[0][k], grads[1][k])
    _report_and_check("one pass vs chunk loop, batch 500", errs)


def test_full_chunk_of_100_against_the_float64_oracle_4x256():
    from oracle import tvae_oracle as O
    from transvae_amd import synth
    from test_model_gpu import run_chunk
    toks = synth.synth_tokens(100, seed=101)
    props = torch.zeros(100)
    eps = torch.randn(100, 128, generator=torch.Generator().manual_seed(12))
    vae = build("4x256_zinc")
    vae.model.eval()
    out, _ = run_chunk(vae, toks, props, eps, 0.02)
    out[0].backward()
    got = np.array([float(o) for o in out])
    P = oracle_params("4x256_zinc", torch.float64, requires_grad=True)
    w = torch.tensor(synth.load_vocab(CONFIGS["4x256_zinc"]["vocab"])[2], dtype=torch.float64)
    losses, _ = O.chunk_loss(P, toks, props.double(), w, 0.02, eps=eps.double())
    losses[0].backward()
    want = np.array([float(x) for x in losses])
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-7)
    named = dict(vae.model.named_parameters())
    errs = {}
    for k, p in named.items():
        ref = P[k].grad.numpy()
        if np.linalg.norm(ref) < 1e-10:
            continue
        errs[k] = rel_l2(p.grad.cpu(), ref)
    _report_and_check("full chunk of 100 vs the fp64 oracle", errs)
    # the same yardstick the golden test uses: how far does a plain fp32 run of the reference's own
    # arithmetic (the oracle in float32 on the host) land from the float64 truth on THESE inputs?
    P32 = oracle_params("4x256_zinc", torch.float32, requires_grad=True)
    l32, _ = O.chunk_loss(P32, toks, props, w.float(), 0.02, eps=eps)
    l32[0].backward()
    ref = {k: rel_l2(P32[k].grad.double().numpy(), P[k].grad.numpy()) for k in errs}
    rv, mv = np.array(list(ref.values())), np.array(list(errs.values()))
    print("    the oracle's own fp32 run vs fp64 on the same chunk: median %.2e, 80%% %.2e, worst %.2e"
          % (float(np.median(rv)), float(np.sort(rv)[int(0.8 * len(rv))]), float(rv.max())))
    for k, e in sorted(errs.items(), key=lambda kv: -kv[1])[:16]:
        print("    %-55s mine %.2e   oracle fp32 %.2e" % (k, e, ref[k]))
    # Which gates flip differs between any two fp32 implementations, and ONE flipped gate behind
    # decoder.linear (576 ReLUs per molecule, 100 molecules) moves that weight's gradient by ~4e-3 and
    # z_var's with it -- seen here -- so only the bulk is compared with the fp32 yardstick; the cap on the
    # worst tensor is _report_and_check's.
    assert float(np.median(mv)) < max(2e-5, 5.0 * float(np.median(rv)))


def test_decode_is_row_wise_at_slab_size_4x256():
    vae = build("4x256_zinc")
    vae.model.eval()
    n = 3000                                                  # above the graph-replay limit: the plain slab path
    z = torch.randn(n, 128, generator=torch.Generator().manual_seed(13)).cuda()
    whole = vae.greedy_decode(z)
    assert whole.shape == (n, 125) and whole.dtype == torch.int64
    assert int(whole.min()) >= 1 and int(whole.max()) <= 25
    pieces = torch.cat([vae.greedy_decode(z[i:i + 1000]) for i in range(0, n, 1000)], 0)
    assert torch.equal(whole, pieces)
    again = torch.cat([vae.greedy_decode(z[i:i + 1000]) for i in range(0, n, 1000)], 0)   # second call: graph replay
    assert torch.equal(whole, again)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(14)).cuda()
    assert torch.equal(vae.greedy_decode(z[perm]), whole[perm])
