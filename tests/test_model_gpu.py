"""Whole-path parity on the MI355X: the TransVAE classes (HIP kernels through the C ABI) against
(a) golden vectors captured from the real reference and (b) the float64 oracle on the same
seeded inputs.  Tolerance: 1e-5 relative on loss and per-tensor gradient L2 (north star), tokens
bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import CONFIGS, load, oracle_params, rel_l2  # noqa: E402


def build(cfg_name, **extra):
    from transvae_amd import synth
    from transvae_amd.trans_models import TransVAE
    cfg = CONFIGS[cfg_name]
    cd, od, w = synth.load_vocab(cfg["vocab"])
    params = {"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w}
    params.update(extra)
    vae = TransVAE(params=params, name="t_" + cfg_name, property_predictor=cfg["pp"], **synth.ARCHS[cfg["arch"]])
    sd = synth.synth_state_dict(vae.model.state_dict(), 0)
    vae.model.load_state_dict(sd)
    return vae


def run_chunk(vae, toks, props, eps, beta):
    from transvae_amd.data import make_src_mask, make_std_mask
    from transvae_amd.loss import trans_vae_loss
    dev = torch.device("cuda:0")
    src = toks.to(dev).long()
    tgt = src[:, :-1]
    src_mask, tgt_mask = make_src_mask(src, vae.pad_idx), make_std_mask(tgt, vae.pad_idx)
    x_out, mu, logvar, pred_len, prop = vae.model(src, tgt, src_mask, tgt_mask, eps.to(dev))
    true_len = src_mask.sum(dim=-1)
    out = trans_vae_loss(src, x_out, mu, logvar, true_len, pred_len, props.to(dev), prop, vae.params["CHAR_WEIGHTS"], beta)
    return out, (x_out, mu, logvar, pred_len, prop)


@pytest.fixture
def gemm_mode(request):
    "run a test under one of the fp32-accurate GEMM arithmetics (exact fp32 MFMA | 3xbf16 split) and restore the default"
    from transvae_amd import ops
    prev = ops.compute_dtype()
    ops.set_compute_dtype(request.param)
    yield request.param
    ops.set_compute_dtype(prev)


@pytest.mark.parametrize("gemm_mode", ["fp32", "f32x3"], indirect=True)
@pytest.mark.parametrize("cfg_name", ["1x128_zinc", "4x256_zinc", "4x256_pubchem_pp"])
def test_train_chunk_matches_reference_golden_and_oracle(cfg_name, gemm_mode):
    from oracle import tvae_oracle as O
    from transvae_amd import synth
    g = load("step_" + cfg_name)
    vae = build(cfg_name)
    vae.model.eval()                       # dropout off, as in the golden capture
    toks = torch.from_numpy(g["tokens"].astype(np.int64))
    props, eps = torch.from_numpy(g["props"]), torch.from_numpy(g["eps"])
    out, (x_out, mu, logvar, pred_len, prop) = run_chunk(vae, toks, props, eps, float(g["beta"]))
    out[0].backward()
    got = np.array([float(o) for o in out])
    # forward: loss components within 1e-5 relative of the reference (fp64 truth and its own fp32 run)
    np.testing.assert_allclose(got, g["loss_f64"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(got, g["loss_f32"], rtol=1e-5, atol=1e-7)
    assert rel_l2(mu.detach().cpu(), g["mu"]) < 1e-5
    assert rel_l2(logvar.detach().cpu(), g["logvar"]) < 1e-5
    assert rel_l2(pred_len.detach().cpu(), g["pred_len"]) < 1e-5
    assert rel_l2(x_out.detach().cpu()[[0, 2, 5]], g["logits_rows"]) < 1e-5
    if "pred_prop" in g.files:
        assert rel_l2(prop.detach().cpu(), g["pred_prop"]) < 1e-5
    # backward.  Bar: per-tensor relative L2 vs the fp64 truth
    #   * at least 0.8x as many tensors within 1e-5 (north star) as the reference's own fp32 run
    #     manages on these inputs, median below max(5e-6, 2x the reference's median);
    #   * every tensor within max(2e-5, 10x the reference's OWN fp32 rounding error on that tensor
    #     (its fp32 run vs its fp64 run, stored in the golden)).  The tensors above 1e-5 are the
    #     source-attention query/key projections and the LayerNorm in front of them: with the
    #     near-uniform attention of these weights dS = P (dP - sum P dP) cancels 2-3 digits in ANY
    #     fp32 implementation -- the reference's own fp32 error there is 0.4-1.1e-5.
    #   * key-projection biases have an exactly-zero true gradient (softmax shift invariance):
    #     absolute bound.
    named = dict(vae.model.named_parameters())
    keys = [str(k) for k in g["grad_keys"]]
    assert set(keys) == set(named)
    ref_err = {k: float(e) for k, e in zip(keys, g["grad_ref_f32_relerr"])}
    # floor 2e-5 (round 2: 4e-5).  The golden inputs are now chosen safe for EVERY gate population: the bottleneck's ReLU /
    # MaxPool gates and the 910,000 hidden ReLUs per molecule of the feed-forward blocks (tools/make_goldens.py
    # safe_step_inputs; the margins, relative to their thresholds, are stored in the golden).  Round 2 screened the
    # bottleneck only, and ONE feed-forward gate deciding differently from float64 (|pre| < 2e-7) put 2-3e-5 on the
    # tensors on the input side of its block -- in either arithmetic, and in the reference's own fp32 run too (its worst
    # tensor went 4.5e-4 -> 9e-6 with the new inputs).
    floor = 2e-5
    tol = {k: max(floor, 10.0 * ref_err[k]) for k in keys}
    # (1) reference summaries: gradient norm of every tensor
    for i, k in enumerate(keys):
        gr = named[k].grad.detach().cpu().double().numpy().ravel()
        nrm = g["grad_norm_f64"][i]
        if nrm < 1e-10:
            assert np.linalg.norm(gr) < 1e-6, k
        else:
            assert abs(np.linalg.norm(gr) - nrm) < tol[k] * nrm, (k, np.linalg.norm(gr), nrm, tol[k])
    for name in g.files:
        if name.startswith("grad_full/"):
            k = name[len("grad_full/"):]
            assert rel_l2(named[k].grad.cpu(), g[name]) < tol[k], k
    # (2) per-tensor relative L2 against the float64 oracle (itself pinned to the reference)
    P = oracle_params(cfg_name, torch.float64, requires_grad=True)
    w = torch.tensor(synth.load_vocab(CONFIGS[cfg_name]["vocab"])[2], dtype=torch.float64)
    losses, _ = O.chunk_loss(P, toks, props.double(), w, float(g["beta"]), eps=eps.double())
    losses[0].backward()
    errs = {}
    for k in keys:
        ref = P[k].grad.numpy()
        if np.linalg.norm(ref) < 1e-10:
            continue
        errs[k] = rel_l2(named[k].grad.cpu(), ref)
    worst = max(errs, key=errs.get)
    n_tight = sum(e < 1e-5 for e in errs.values())
    print("grad rel-L2 vs fp64 oracle [%s, %s]: median %.2e, worst %.2e (%s; reference's own fp32 error there %.2e), "
          "%d/%d tensors within 1e-5" % (cfg_name, gemm_mode, float(np.median(list(errs.values()))), errs[worst], worst,
                                         ref_err[worst], n_tight, len(errs)))
    for k, e in sorted(errs.items(), key=lambda kv: -kv[1])[:14]:
        print("    %-55s mine %.2e   reference fp32 %.2e" % (k, e, ref_err[k]))
    from helpers import record
    ref_vals_all = np.array([ref_err[k] for k in errs])
    record("golden_step_%s" % gemm_mode,
           ["# %s, mode %s: loss terms rel err vs reference fp64 %s" % (cfg_name, gemm_mode, np.array2string(
               np.abs(got - g["loss_f64"]) / np.maximum(np.abs(g["loss_f64"]), 1e-12), precision=1)),
            "gradient rel-L2 vs fp64: mine median %.2e worst %.2e, %d/%d within 1e-5 | reference's own fp32 run median %.2e worst "
            "%.2e, %d/%d within 1e-5" % (float(np.median(list(errs.values()))), errs[worst], n_tight, len(errs),
                                         float(np.median(ref_vals_all)), float(ref_vals_all.max()),
                                         int((ref_vals_all < 1e-5).sum()), len(errs)),
            "%-58s %10s %10s %10s" % ("tensor", "mine", "ref fp32", "bar")] +
           ["%-58s %10.2e %10.2e %10.2e" % (k, e, ref_err[k], tol[k]) for k, e in sorted(errs.items(), key=lambda kv: -kv[1])])
    for k, e in errs.items():
        assert e < tol[k], (k, e, tol[k])
    # aggregate bars, relative to how well the reference's own fp32 run does on the same inputs
    ref_vals = np.array([ref_err[k] for k in errs])
    ref_tight = int((ref_vals < 1e-5).sum())
    print("    reference fp32 on the same inputs: median %.2e, %d/%d within 1e-5" % (np.median(ref_vals), ref_tight, len(errs)))
    assert n_tight >= 0.8 * ref_tight, (n_tight, ref_tight, len(errs))
    assert float(np.median(list(errs.values()))) < max(5e-6, 2.0 * float(np.median(ref_vals)))


@pytest.mark.parametrize("gemm_mode", ["bf16"], indirect=True)
@pytest.mark.parametrize("cfg_name", ["1x128_zinc", "4x256_zinc", "4x256_pubchem_pp"])
def test_train_chunk_bf16_mode_within_1e_2(cfg_name, gemm_mode):
    """bf16 mode (GEMM operands rounded to bf16, fp32 accumulation; LayerNorm, softmax, loss, attention in
    fp32) against the reference's fp64 run on the golden inputs.  What is held to the north star's 1e-2:
    every loss term (measured: <= 1.3e-3), the encoder outputs and the logits, and the gradient NORM of the
    median tensor (measured 1e-3; worst tensor 2.8e-2).  What is not: the element-wise direction of a
    gradient tensor -- rel-L2 2e-2 at the median, 0.1-0.18 for the worst (embedding rows, z_var bias):
    forty-odd GEMMs with 2^-9 operand rounding in a row, no reference bf16 path to compare with
    (the reference has no autocast), so the bars below are the measured level plus margin and DESIGN.md
    section 2 states it."""
    g = load("step_" + cfg_name)
    vae = build(cfg_name)
    vae.model.eval()
    toks = torch.from_numpy(g["tokens"].astype(np.int64))
    props, eps = torch.from_numpy(g["props"]), torch.from_numpy(g["eps"])
    out, (x_out, mu, logvar, pred_len, prop) = run_chunk(vae, toks, props, eps, float(g["beta"]))
    out[0].backward()
    got = np.array([float(o.detach()) for o in out])
    np.testing.assert_allclose(got, g["loss_f64"], rtol=1e-2, atol=1e-6)
    assert rel_l2(mu.detach().cpu(), g["mu"]) < 1e-2
    assert rel_l2(logvar.detach().cpu(), g["logvar"]) < 1e-2
    assert rel_l2(pred_len.detach().cpu(), g["pred_len"]) < 1e-2
    assert rel_l2(x_out.detach().cpu()[[0, 2, 5]], g["logits_rows"]) < 1e-2
    named = dict(vae.model.named_parameters())
    keys = [str(k) for k in g["grad_keys"]]
    # gradient norms of every tensor against the reference's fp64 norms, full gradients where the golden holds them
    nerr = {}
    for i, k in enumerate(keys):
        nrm = g["grad_norm_f64"][i]
        if nrm < 1e-10:
            continue
        nerr[k] = abs(float(np.linalg.norm(named[k].grad.detach().cpu().double().numpy())) - nrm) / nrm
    ferr = {}
    for name in g.files:
        if name.startswith("grad_full/"):
            k = name[len("grad_full/"):]
            if np.linalg.norm(g[name]) > 1e-10:
                ferr[k] = rel_l2(named[k].grad.cpu(), g[name])
    worst_n, worst_f = max(nerr, key=nerr.get), max(ferr, key=ferr.get)
    print("bf16 [%s]: losses rel err %s; gradient-norm error median %.2e worst %.2e (%s); full-gradient rel-L2 median %.2e worst %.2e (%s)"
          % (cfg_name, np.array2string(np.abs(got - g["loss_f64"]) / np.maximum(np.abs(g["loss_f64"]), 1e-12), precision=1),
             float(np.median(list(nerr.values()))), nerr[worst_n], worst_n,
             float(np.median(list(ferr.values()))), ferr[worst_f], worst_f))
    from helpers import record
    record("golden_step_bf16", ["# %s, mode bf16: loss terms rel err vs reference fp64 %s" % (cfg_name, np.array2string(
        np.abs(got - g["loss_f64"]) / np.maximum(np.abs(g["loss_f64"]), 1e-12), precision=1)),
        "gradient-norm error median %.2e worst %.2e (%s); full-gradient rel-L2 (the %d tensors the golden holds in full) median %.2e worst %.2e (%s)"
        % (float(np.median(list(nerr.values()))), nerr[worst_n], worst_n, len(ferr), float(np.median(list(ferr.values()))), ferr[worst_f], worst_f),
        "%-58s %10s %10s" % ("tensor", "norm err", "full rel-L2")] +
        ["%-58s %10.2e %10s" % (k, e, ("%.2e" % ferr[k]) if k in ferr else "-") for k, e in sorted(nerr.items(), key=lambda kv: -kv[1])])
    assert float(np.median(list(nerr.values()))) < 1e-2
    assert nerr[worst_n] < 5e-2
    assert float(np.median(list(ferr.values()))) < 3e-2
    assert ferr[worst_f] < 0.25


def test_premasked_residual_gradients_equal_masks_regenerated_in_the_gemms(monkeypatch):
    """Training mode, dropout on.  The gradient at a sublayer's output is written by the next LayerNorm backward, which
    also writes it with the sublayer's residual-dropout mask applied (tvae_layernorm_bwd_masked); the sublayer's two
    gradient products then read that copy instead of regenerating the mask while staging.  Same seed, both ways: the
    same losses and the same gradients (weight gradients are summed with atomics: 2e-6)."""
    from transvae_amd import ops, synth
    g = load("step_4x256_zinc")
    toks = torch.from_numpy(g["tokens"].astype(np.int64))
    props, eps = torch.from_numpy(g["props"]), torch.from_numpy(g["eps"])
    grads, losses = {}, {}
    vae = build("4x256_zinc")                   # ONE model: dropout site ids are handed out at construction
    vae.model.train()
    for flag in ("0", "1"):
        monkeypatch.setenv("TVAE_PREMASK", flag)
        vae.model.zero_grad(set_to_none=True)
        ops.seed_rng(torch.device("cuda:0"), 31)
        ops.begin_step()
        before = dict(ops.premask_stats)
        out, _ = run_chunk(vae, toks, props, eps, 0.02)
        out[0].backward()
        used = ops.premask_stats["used"] - before["used"]
        assert used == (0 if flag == "0" else 17), used        # every one of the 17 sublayers of the model took the masked copy
        losses[flag] = [float(o.detach()) for o in out]
        grads[flag] = {k: p.grad.detach().clone() for k, p in vae.model.named_parameters()}
    np.testing.assert_allclose(losses["1"], losses["0"], rtol=1e-6)     # the loss kernel sums its blocks with atomics: order, not arithmetic
    for k in grads["0"]:
        if float(grads["0"][k].norm()) < 1e-7:       # key-projection biases: the true gradient is exactly zero, both are rounding noise
            assert float(grads["1"][k].norm()) < 1e-7, k
            continue
        assert rel_l2(grads["1"][k].cpu(), grads["0"][k].cpu()) < 2e-6, k


def test_weight_gradients_on_the_second_stream_equal_the_single_stream_ones(monkeypatch):
    """Inside direct_grads the weight-gradient products of the Linear layers run on a second stream beside the backward
    pass (ops._side_begin); leaving the context joins it.  Same seed with TVAE_WGRAD_STREAM=0 and =1: the same gradients
    (split-K sums use atomics: 2e-6), read right after the context without any host synchronisation in between; and the
    second stream was really used (17 sublayers -> at least 30 products)."""
    from transvae_amd import ops
    g = load("step_4x256_zinc")
    toks = torch.from_numpy(g["tokens"].astype(np.int64))
    props, eps = torch.from_numpy(g["props"]), torch.from_numpy(g["eps"])
    vae = build("4x256_zinc")
    vae.model.train()
    grads, launched = {}, {}
    for flag in ("0", "1", "1"):                   # the second "1" pass starts from gradient buffers that exist already
        monkeypatch.setenv("TVAE_WGRAD_STREAM", flag)
        vae.model.zero_grad(set_to_none=(flag == "0"))
        ops.seed_rng(torch.device("cuda:0"), 31)
        ops.begin_step()
        n_side = []
        with ops.direct_grads(ops.GradSink()):
            out, _ = run_chunk(vae, toks, props, eps, 0.02)
            out[0].backward()
            n_side.append(sum(e[1] for e in ops._side.values()))
        launched[flag] = n_side[0]
        assert sum(e[1] for e in ops._side.values()) == 0          # joined on the way out
        grads[flag] = {k: p.grad.detach().clone() for k, p in vae.model.named_parameters()}
        assert launched[flag] == (0 if flag == "0" else launched[flag]) and (flag == "0" or launched[flag] >= 30), launched
        if flag == "1":
            for k in grads["0"]:
                if float(grads["0"][k].norm()) < 1e-7:
                    assert float(grads["1"][k].norm()) < 1e-7, k
                    continue
                assert rel_l2(grads["1"][k].cpu(), grads["0"][k].cpu()) < 2e-6, k


def test_accumulation_noam_adam_matches_reference():
    g = load("accum_1x128_zinc")
    vae = build("1x128_zinc", WARMUP_STEPS=int(g["warmup"]), BATCH_CHUNKS=2, BATCH_SIZE=12)
    vae.model.eval()
    named = dict(vae.model.named_parameters())
    p0 = {str(k): named[str(k)].detach().cpu().numpy().copy() for k in g["keys"]}
    idx = 0
    for step in range(1, 4):
        toks = torch.from_numpy(np.concatenate([g["tokens"][idx], g["tokens"][idx + 1]]).astype(np.float32))
        data = torch.cat([toks, torch.zeros(12, 1)], 1)              # (12, 128): tokens | property
        eps = torch.from_numpy(np.concatenate([g["eps"][idx], g["eps"][idx + 1]])).cuda()
        table = vae.train_step(data, float(g["beta"]), eps=eps).cpu().numpy()
        np.testing.assert_allclose(table[:, 0], g["losses"][idx:idx + 2], rtol=2e-5)
        assert abs(vae.optimizer.state_dict["rate"] - g["rates"][step - 1]) < 1e-15
        idx += 2
        if step in (1, 3):
            for k in g["keys"]:
                ref = g["p_step%d/%s" % (step, k)]
                got = named[str(k)].detach().cpu().numpy()
                assert rel_l2(got - p0[str(k)], ref - p0[str(k)]) < 2e-3, (step, k)
    for p in vae.model.parameters():          # gradients were reset inside the fused Adam launch
        assert float(p.grad.abs().max()) == 0.0


@pytest.mark.parametrize("gemm_mode", ["fp32", "f32x3"], indirect=True)
def test_accumulation_noam_adam_4x256_matches_reference(gemm_mode):
    """The HEADLINE model's optimizer step against the reference's own run of it (tools/make_goldens.py
    golden_accum_4x256: 5 chunks of 6 gate-safe molecules accumulated -> NoamOpt.step, 3 steps, in float32 and float64):
    chunk losses at 1e-5, Noam rates exact, and the parameter UPDATES (p_step - p_0) of ten tensors per step against the
    float64 run.  Bar per (step, tensor): 1.5 x the reference's OWN fp32-vs-fp64 error there + 5e-6.  That error (6e-6 ..
    5e-4, stored in the golden) is mostly the fp32 STORAGE of the parameter: an update of lr = 6e-5 .. 2e-4 added to a
    value of order 1 keeps 3-4 digits, in any implementation -- measured, this build's numbers equal the reference's own
    to 2-3 digits on all thirty (step, tensor) pairs (profiles/r3_parity_accum_4x256_*.txt).  The inputs of every step
    are chosen gate-safe ON THAT STEP'S WEIGHTS: with inputs screened on the initial weights only, one feed-forward gate
    deciding differently at step 3 showed up as 1e-3 on three tensors (Adam turns a 2e-5 gradient difference into a
    relative update difference of that size where successive gradients nearly cancel)."""
    from helpers import record
    g = load("accum_4x256_zinc")
    chunks, csize = int(g["chunks"]), int(g["csize"])
    vae = build("4x256_zinc", WARMUP_STEPS=int(g["warmup"]), BATCH_CHUNKS=chunks, BATCH_SIZE=chunks * csize)
    vae.model.eval()
    named = dict(vae.model.named_parameters())
    keys = [str(k) for k in g["keys"]]
    for k in keys:
        np.testing.assert_array_equal(named[k].detach().cpu().numpy(), g["p0/" + k].astype(np.float32))
    lines = ["# accumulate %d chunks of %d -> NoamOpt.step, trans4x-256, mode %s: update rel-L2 vs the reference's float64 run"
             % (chunks, csize, gemm_mode), "%-4s %-55s %10s %10s %10s" % ("step", "tensor", "mine", "ref fp32", "bar")]
    bad = []
    for step in range(1, 4):
        toks = torch.from_numpy(g["tokens"][step - 1].astype(np.float32))
        data = torch.cat([toks, torch.zeros(chunks * csize, 1)], 1)
        eps = torch.from_numpy(g["eps"][step - 1]).cuda()
        table = vae.train_step(data, float(g["beta"]), eps=eps).cpu().numpy()
        sl = slice((step - 1) * chunks, step * chunks)
        np.testing.assert_allclose(table[:, 0], g["losses_f64"][sl], rtol=1e-5)
        np.testing.assert_allclose(table[:, 0], g["losses_f32"][sl], rtol=1e-5)
        assert abs(vae.optimizer.state_dict["rate"] - g["rates"][step - 1]) < 1e-15
        for i, k in enumerate(keys):
            p0 = g["p0/" + k]
            err = rel_l2(named[k].detach().cpu().double().numpy() - p0, g["p_f64_step%d/%s" % (step, k)] - p0)
            ref = float(g["update_ref_f32_relerr"][step - 1, i])
            bar = 1.5 * ref + 5e-6
            lines.append("%-4d %-55s %10.2e %10.2e %10.2e" % (step, k, err, ref, bar))
            if err >= bar:
                bad.append((step, k, err, bar))
    record("accum_4x256_%s" % gemm_mode, lines)
    assert not bad, bad


@pytest.mark.parametrize("gemm_mode", ["fp32", "f32x3"], indirect=True)
@pytest.mark.parametrize("cfg_name", ["1x128_zinc", "4x256_zinc"])
def test_greedy_decode_tokens_bit_exact(cfg_name, gemm_mode):
    g = load("decode_" + cfg_name)
    vae = build(cfg_name)
    z = torch.from_numpy(g["z"])
    lens = vae.model.encoder.predict_mask_length(z.cuda())
    assert np.array_equal(lens.cpu().numpy()[:, 0], g["mask_len"].astype(np.int64))
    ids = vae.greedy_decode(z)
    n = g["z"].shape[0]                   # 32 latents (trans1x-128), 256 (trans4x-256: 32,000 arg-max decisions, top-2 margins stored)
    assert ids.dtype == torch.int64 and tuple(ids.shape) == (n, 125)
    bad = (ids.cpu().numpy() != g["ids"].astype(np.int64))
    from helpers import record
    record("decode_golden_%s" % gemm_mode, "%s, mode %s: %d x 125 tokens against the reference's, %d differ; smallest top-2 logit margin in "
           "the golden %.2e, 1st percentile %.2e" % (cfg_name, gemm_mode, n, int(bad.sum()), float(g["margins"].min()),
                                                    float(np.percentile(g["margins"], 1))))
    assert not bad.any(), "mismatch at %s, margins there %s" % (np.argwhere(bad)[:5], g["margins"][bad][:5])
    ids_c = vae.greedy_decode(z[:8], condition=[str(c) for c in g["cond"]])
    assert np.array_equal(ids_c.cpu().numpy(), g["ids_cond"].astype(np.int64))
    lens_e = torch.from_numpy(g["mask_lens_explicit"].astype(np.int64))
    sm = (torch.arange(127)[None, :] < lens_e[:, None]).float().unsqueeze(1)
    ids_m = vae.greedy_decode(z[8:16], src_mask=sm)
    assert np.array_equal(ids_m.cpu().numpy(), g["ids_mask"].astype(np.int64))
    # strings: same post-processing as the reference's decode_mols
    from oracle import tvae_oracle as O
    from transvae_amd.tvae_util import decode_mols
    assert decode_mols(ids, vae.params["ORG_DICT"]) == O.decode_mols(torch.from_numpy(g["ids"].astype(np.int64)), vae.params["ORG_DICT"])


@pytest.mark.parametrize("gemm_mode", ["fp32", "f32x3"], indirect=True)
def test_graph_replayed_decode_is_the_same_decode(gemm_mode):
    """From the second call with one shape on, the 125 steps run as a replayed HIP graph: tokens must stay the
    reference's bit for bit, for the latents it was captured with, for new latents and masks, with condition
    tokens, and after the weights change (the capture is thrown away)."""
    from transvae_amd import decode
    g = load("decode_1x128_zinc")
    vae = build("1x128_zinc")
    z = torch.from_numpy(g["z"])
    want = g["ids"].astype(np.int64)
    for call in range(3):                                     # eager, capture + replay, replay
        assert np.array_equal(vae.greedy_decode(z).cpu().numpy(), want), call
    graphs = vae.model._tvae_decode_graphs["graphs"]
    assert len(graphs) == 1
    z2 = torch.flip(z, [0])                                   # other latents through the captured graph
    assert np.array_equal(vae.greedy_decode(z2).cpu().numpy(), want[::-1])
    lens_e = torch.from_numpy(np.resize(g["mask_lens_explicit"], 32).astype(np.int64))
    sm = (torch.arange(127)[None, :] < lens_e[:, None]).float().unsqueeze(1)
    got_m = [vae.greedy_decode(z, src_mask=sm).cpu().numpy() for _ in range(2)]      # replayed: same shape
    cond = [str(c) for c in g["cond"]]
    got_c = [vae.greedy_decode(z[:8], condition=cond).cpu().numpy() for _ in range(3)]
    assert all(np.array_equal(a, g["ids_cond"].astype(np.int64)) for a in got_c)
    limit, decode.GRAPH_MAX_N = decode.GRAPH_MAX_N, 0         # plain launches for comparison
    try:
        assert all(np.array_equal(a, vae.greedy_decode(z, src_mask=sm).cpu().numpy()) for a in got_m)
        with torch.no_grad():                                 # move the weights: in-place, as an optimizer would
            vae.model.generator.proj.bias.add_(torch.linspace(-3, 3, 25, device="cuda"))
        moved = vae.greedy_decode(z).cpu().numpy()
    finally:
        decode.GRAPH_MAX_N = limit
    assert not np.array_equal(moved, want)
    for call in range(3):
        assert np.array_equal(vae.greedy_decode(z).cpu().numpy(), moved), call
    # the cache is bounded: new batch sizes push the least recently used graph out, results stay the reference's
    for nb in (5, 6, 7, 9, 10, 11):
        for call in range(2):
            vae.greedy_decode(z[:nb])
    assert len(graphs) <= decode.GRAPH_CACHE_MAX
    assert np.array_equal(vae.greedy_decode(z).cpu().numpy(), moved)
    decode.clear_decode_graphs(vae.model)
    assert "_tvae_decode_graphs" not in vae.model.__dict__


def test_reference_style_decode_loop_matches_cached_decoder():
    """model.decode(mem, src_mask, decoded, subsequent_mask(i)) -- the reference's own loop body --
    must agree with the KV-cached path token for token (a few steps, small batch)."""
    from transvae_amd.tvae_util import subsequent_mask
    vae = build("1x128_zinc")
    vae.model.eval()
    g = load("decode_1x128_zinc")
    z = torch.from_numpy(g["z"][:4]).cuda()
    lens = vae.model.encoder.predict_mask_length(z)
    src_mask = (torch.arange(127, device="cuda")[None, :] < lens).float().unsqueeze(1)
    decoded = torch.zeros(4, 1, dtype=torch.long, device="cuda")
    with torch.no_grad():
        for i in range(12):
            out = vae.model.generator(vae.model.decode(z, src_mask, decoded, subsequent_mask(decoded.size(1)).long()))
            nxt = torch.max(out[:, i, :], dim=1)[1] + 1
            decoded = torch.cat([decoded, nxt.unsqueeze(1)], 1)
    assert np.array_equal(decoded[:, 1:].cpu().numpy(), g["ids"][:4, :12].astype(np.int64))


@pytest.mark.parametrize("cfg_name", ["1x128_zinc", "4x256_pubchem_pp"])
def test_attention_weight_dump_matches_reference(cfg_name):
    "forward_w_attn (scripts/attention.py path): dk = 32, and dk = 64 on BASELINE config 5's model (PubChem vocab + property head)"
    from transvae_amd.data import make_src_mask, make_std_mask
    g = load("attn_" + cfg_name)
    vae = build(cfg_name)
    vae.model.eval()
    src = torch.from_numpy(g["tokens"].astype(np.int64)).cuda()
    tgt = src[:, :-1]
    src_mask, tgt_mask = make_src_mask(src, vae.pad_idx), make_std_mask(tgt, vae.pad_idx)
    with torch.no_grad():
        mem, mu, logvar, pred_len, enc_w = vae.model.encoder.forward_w_attn(vae.model.src_embed(src), src_mask,
                                                                            torch.from_numpy(g["eps"]).cuda())
        x, fin_w, src_w = vae.model.decoder.forward_w_attn(vae.model.tgt_embed(tgt), mem, src_mask, tgt_mask)
    heads = g["heads"].tolist()
    for i in range(3):
        assert tuple(enc_w[i].shape) == (2, 4, 127, 127) and tuple(src_w[i].shape) == (2, 4, 126, 127)
        np.testing.assert_allclose(enc_w[i][:, heads].numpy(), g["enc"][i].astype(np.float32), atol=1e-3)
        np.testing.assert_allclose(src_w[i][:, heads].numpy(), g["src"][i].astype(np.float32), atol=1e-3)
    np.testing.assert_allclose(fin_w[0][:, heads].numpy(), g["fin"].astype(np.float32), atol=1e-3)
    assert rel_l2(x[:, :4].cpu(), g["x_rows"]) < 1e-5


def test_training_mode_dropout_runs_and_learns(tmp_path):
    """Dropout on (the reference's real training mode): losses finite, parameters move, the loss
    goes down over a few steps on a fixed batch, checkpoint round-trips."""
    from transvae_amd import synth, ops
    from transvae_amd.trans_models import TransVAE
    vae = build("1x128_zinc", WARMUP_STEPS=50, BATCH_SIZE=40, BATCH_CHUNKS=2)
    ops.seed_rng(torch.device("cuda:0"), 7)
    vae.model.train()
    toks = synth.synth_tokens(40, seed=3).float()
    data = torch.cat([toks, torch.zeros(40, 1)], 1)
    first = None
    for step in range(12):
        t = vae.train_step(data, 0.01).cpu().numpy()
        assert np.isfinite(t).all()
        first = t[:, 0].mean() if first is None else first
    assert t[:, 0].mean() < first
    vae.current_state["model_state_dict"] = vae.model.state_dict()
    vae.optimizer.sync_state_dict()
    vae.current_state["optimizer_state_dict"] = vae.optimizer.state_dict
    vae.current_state["epoch"] = 3
    vae.save(vae.current_state, "unit", path=str(tmp_path))
    path = os.path.join(str(tmp_path), "unit_%s.ckpt" % vae.name)
    assert os.path.exists(path)
    vae2 = TransVAE(load_fn=path)
    assert vae2.n_epochs == 3 and vae2.params["d_model"] == 128 and vae2.name == vae.name
    for (k1, v1), (k2, v2) in zip(vae.model.state_dict().items(), vae2.model.state_dict().items()):
        assert k1 == k2 and torch.equal(v1.cpu(), v2.cpu())
    assert vae2.optimizer.state_dict["step"] == 12


def test_generic_module_paths_agree_with_fused_paths():
    """MultiHeadedAttention / SublayerConnection / PositionwiseFeedForward used one by one (the
    reference's call pattern) give the same numbers as the fused sublayer Functions."""
    vae = build("1x128_zinc")
    vae.model.eval()
    layer = vae.model.encoder.layers[0]
    x = torch.randn(3, 127, 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    mask = torch.ones(3, 1, 127, device="cuda")
    mask[1, :, 40:] = 0
    with torch.no_grad():
        fused = layer(x, mask)
        y = layer.sublayer[0](x, lambda t: layer.self_attn(t, t, t, mask))
        y = layer.sublayer[1](y, layer.feed_forward)
    assert rel_l2(y.cpu(), fused.cpu()) < 1e-6


def test_fused_chunks_equal_literal_chunk_loop():
    """One pass over the batch with per-chunk loss normalisation == the reference's chunk loop:
    same per-chunk losses, same accumulated gradients (dropout off, same eps)."""
    from transvae_amd import synth
    grads, tables = [], []
    toks = synth.synth_tokens(24, seed=9).float()
    data = torch.cat([toks, torch.zeros(24, 1)], 1)
    eps = torch.randn(24, 128, generator=torch.Generator().manual_seed(4)).cuda()
    for fuse in (True, False):
        vae = build("1x128_zinc", BATCH_SIZE=24, BATCH_CHUNKS=3, FUSE_CHUNKS=fuse)
        vae.model.eval()
        vae.optimizer.optimizer.fuse_zero_grad = False       # keep the gradients to look at them
        tables.append(vae.train_step(data, 0.03, eps=eps).cpu().numpy())
        grads.append({k: p.grad.detach().cpu().numpy().copy() for k, p in vae.model.named_parameters()})
    np.testing.assert_allclose(tables[0], tables[1], rtol=2e-6, atol=1e-7)
    for k in grads[0]:
        if k.endswith("attn.linears.1.bias"):       # analytically zero (softmax shift invariance): noise only
            assert np.linalg.norm(grads[0][k]) < 1e-5 and np.linalg.norm(grads[1][k]) < 1e-5
        else:
            assert rel_l2(grads[0][k], grads[1][k]) < 2e-5, k


def test_packed_qkv_projections_equal_separate_ones(monkeypatch):
    """VAEShell lays q, k, v out as row blocks of one matrix and runs one GEMM for the three (exact-fp32 mode).  Against
    the same model with separately allocated projections (TVAE_PACK_QKV=0): identical forward losses (each output element
    is the same chain of products), gradients equal to rounding; parameter names and shapes untouched."""
    from transvae_amd import synth, ops
    toks = synth.synth_tokens(16, seed=5).float()
    data = torch.cat([toks, torch.zeros(16, 1)], 1)
    eps = torch.randn(16, 128, generator=torch.Generator().manual_seed(3)).cuda()
    grads, tables, names = [], [], []
    for pack in ("1", "0"):
        monkeypatch.setenv("TVAE_PACK_QKV", pack)
        vae = build("1x128_zinc", BATCH_SIZE=16, BATCH_CHUNKS=2)
        lin = vae.model.encoder.layers[0].self_attn.linears
        adjacent = ops._adjacent(lin[0].weight, lin[1].weight, lin[2].weight) and ops._adjacent(lin[0].weight.grad, lin[1].weight.grad, lin[2].weight.grad) \
            if lin[0].weight.grad is not None else False
        assert adjacent == (pack == "1")
        vae.model.eval()
        vae.optimizer.optimizer.fuse_zero_grad = False
        tables.append(vae.train_step(data, 0.03, eps=eps).cpu().numpy())
        grads.append({k: p.grad.detach().cpu().numpy().copy() for k, p in vae.model.named_parameters()})
        names.append([(k, tuple(v.shape)) for k, v in vae.model.state_dict().items()])
    assert names[0] == names[1]
    assert np.array_equal(tables[0], tables[1])
    for k in grads[0]:
        if k.endswith("attn.linears.1.bias"):
            assert np.linalg.norm(grads[0][k]) < 1e-5 and np.linalg.norm(grads[1][k]) < 1e-5
        else:
            assert rel_l2(grads[0][k], grads[1][k]) < 1e-5, k


def test_reference_written_checkpoint_decodes_to_the_reference_tokens():
    """tests/golden/ref_tiny.ckpt (written by the reference's VAEShell.save, read here by the safe loader) must put the
    model in the state the reference decoded from: mask lengths and greedy tokens bit for bit (ref_ckpt.npz)."""
    from transvae_amd.trans_models import TransVAE
    g = load("ref_ckpt")
    vae = TransVAE(load_fn=os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_tiny.ckpt"))
    z = torch.from_numpy(g["z"])
    lens = vae.model.encoder.predict_mask_length(z.cuda())
    assert np.array_equal(lens.cpu().numpy()[:, 0], g["mask_len"].astype(np.int64))
    assert np.array_equal(vae.greedy_decode(z).cpu().numpy(), g["ids"].astype(np.int64))


def test_adamopt_over_the_fused_kernel_matches_the_reference():
    """AdamOpt(params, 3e-4, Adam) -- the reference's fixed-rate wrapper (opt.py:41-52) -- over the one-launch Adam:
    three steps against what the reference's AdamOpt over torch.optim.Adam did with the same gradients (sampling.npz)."""
    from transvae_amd.opt import Adam, AdamOpt
    g = load("sampling")
    n = int(g["adamopt_n"])
    ps = [torch.nn.Parameter(torch.from_numpy(g["adamopt_p0/%d" % i]).cuda()) for i in range(n)]
    opt = AdamOpt(ps, float(g["adamopt_lr"]), Adam)
    assert opt.optimizer.param_groups[0]["lr"] == 3e-4 and opt.optimizer.param_groups[0]["betas"] == (0.9, 0.999)
    for step in range(1, 4):
        for i, p in enumerate(ps):
            p.grad = torch.from_numpy(g["adamopt_g%d/%d" % (step, i)]).cuda()
        opt.step()
        for i, p in enumerate(ps):
            want, p0 = g["adamopt_p%d/%d" % (step, i)], g["adamopt_p0/%d" % i]
            assert rel_l2(p.detach().cpu().numpy() - p0, want - p0) < 2e-6, (step, i)
    assert opt.state_dict["state"][0]["step"] == 3 and opt.state_dict["param_groups"][0]["lr"] == 3e-4


def test_steps_queued_without_a_sync_equal_steps_synced_one_by_one():
    """The host queues train steps ahead of the GPU (train() reads losses two steps late, bench.py never syncs):
    every step must still get ITS learning rate and bias corrections.  Eight steps back to back with no
    synchronisation against the same eight steps with a device sync after each: identical weights."""
    from transvae_amd import synth
    toks = synth.synth_tokens(24, seed=5).float()
    data = torch.cat([toks, torch.zeros(24, 1)], 1).cuda()
    eps = torch.randn(24, 128, generator=torch.Generator().manual_seed(4)).cuda()
    finals = []
    for sync in (False, True):
        vae = build("1x128_zinc", BATCH_SIZE=24, BATCH_CHUNKS=2, WARMUP_STEPS=400)    # linear ramp: the rate of step t is t x the rate of step 1
        vae.model.eval()
        torch.cuda.synchronize()
        for _ in range(8):
            vae.train_step(data, 0.02, eps=eps)
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        finals.append({k: p.detach().cpu().numpy().copy() for k, p in vae.model.named_parameters()})
    for k in finals[0]:
        # split-K atomics add in arrival order, so two runs of the same steps differ in the last bits; a step taken with a
        # neighbouring step's rate would move a weight by 12-100 % of its update
        if k.endswith("attn.linears.1.bias"):      # analytically zero gradient: Adam normalises pure rounding noise
            continue
        p0 = synth.synth_tensor(k, finals[1][k].shape, 0)
        assert rel_l2(finals[0][k] - p0, finals[1][k] - p0) < 0.02, k
