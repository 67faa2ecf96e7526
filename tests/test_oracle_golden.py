"""Pins oracle/tvae_oracle.py against golden vectors captured from the real reference
(tools/make_goldens.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import tvae_oracle as O
from transvae_amd import synth
from helpers import CONFIGS, load, oracle_params, probe_vec, rel_l2


def _chunk(cfg_name, dtype):
    g = load("step_" + cfg_name)
    P = oracle_params(cfg_name, dtype, requires_grad=True)
    toks = torch.from_numpy(g["tokens"].astype(np.int64))
    w = torch.tensor(synth.load_vocab(CONFIGS[cfg_name]["vocab"])[2], dtype=torch.float).to(dtype)
    losses, aux = O.chunk_loss(P, toks, torch.from_numpy(g["props"]).to(dtype), w, float(g["beta"]),
                               eps=torch.from_numpy(g["eps"]).to(dtype))
    losses[0].backward()
    return g, P, losses, aux


@pytest.mark.parametrize("cfg_name", ["1x128_zinc", "4x256_zinc", "4x256_pubchem_pp"])
def test_step_forward_and_grads_f64(cfg_name):
    g, P, losses, aux = _chunk(cfg_name, torch.float64)
    got = np.array([float(x) for x in losses])
    np.testing.assert_allclose(got, g["loss_f64"], rtol=1e-11, atol=1e-13)
    keys = [str(k) for k in g["grad_keys"]]
    assert set(keys) == {k for k in P if not k.endswith(".pe")}
    assert float(g["gate_margins"].min()) > 2e-6      # golden inputs were chosen gate-safe
    for i, k in enumerate(keys):
        gr = P[k].grad.numpy().ravel()
        nrm = g["grad_norm_f64"][i]
        assert abs(np.linalg.norm(gr) - nrm) <= 1e-9 * nrm + 1e-14, k
        assert abs(gr @ probe_vec(k, gr.size) - g["grad_dot_f64"][i]) <= 1e-8 * nrm * np.sqrt(gr.size) + 1e-14, k
    for name in g.files:
        if name.startswith("grad_full/"):
            k = name[len("grad_full/"):]
            ref = g[name]
            if np.linalg.norm(ref) < 1e-12:      # analytically-zero gradients (key-projection biases)
                assert np.abs(P[k].grad.numpy()).max() < 1e-12
            else:
                assert rel_l2(P[k].grad.numpy(), ref) < 1e-9, k


@pytest.mark.parametrize("cfg_name", ["1x128_zinc", "4x256_zinc", "4x256_pubchem_pp"])
def test_step_f32_matches_reference_f32(cfg_name):
    g, P, losses, aux = _chunk(cfg_name, torch.float32)
    got = np.array([float(x) for x in losses])
    np.testing.assert_allclose(got, g["loss_f32"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(aux["mu"].detach().numpy(), g["mu"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(aux["logvar"].detach().numpy(), g["logvar"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(aux["pred_len"].detach().numpy(), g["pred_len"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(aux["logits"].detach().numpy()[[0, 2, 5]], g["logits_rows"], rtol=1e-4, atol=5e-6)
    if "pred_prop" in g.files:
        np.testing.assert_allclose(aux["prop"].detach().numpy(), g["pred_prop"], rtol=1e-4, atol=2e-6)


def test_accumulate_and_noam_adam():
    g = load("accum_1x128_zinc")
    P = oracle_params("1x128_zinc", torch.float32, requires_grad=True)
    names = [k for k in P if not k.endswith(".pe")]
    w = torch.tensor(synth.load_vocab("zinc")[2], dtype=torch.float)
    m = [torch.zeros_like(P[k]) for k in names]
    v = [torch.zeros_like(P[k]) for k in names]
    idx = 0
    p0 = {str(k): P[str(k)].detach().numpy().copy() for k in g["keys"]}
    for step in range(1, 4):
        for k in names:
            P[k].grad = None
        for c in range(2):
            toks = torch.from_numpy(g["tokens"][idx].astype(np.int64))
            losses, _ = O.chunk_loss(P, toks, torch.zeros(6), w, float(g["beta"]),
                                     eps=torch.from_numpy(g["eps"][idx]))
            assert abs(float(losses[0]) - g["losses"][idx]) < 3e-6 * abs(g["losses"][idx])
            losses[0].backward()         # accumulates, no 1/chunks scaling (trans_models.py:236)
            idx += 1
        lr = O.noam_rate(step, 128, 1.0, int(g["warmup"]))
        assert abs(lr - g["rates"][step - 1]) < 1e-15
        with torch.no_grad():
            O.adam_step([P[k] for k in names], [P[k].grad for k in names], m, v, step, lr)
        if step in (1, 3):
            for k in g["keys"]:
                ref = g["p_step%d/%s" % (step, k)]
                # compare the UPDATE (p - p0): the parameters themselves barely move
                assert rel_l2(P[str(k)].detach().numpy() - p0[str(k)], ref - p0[str(k)]) < 2e-3, (step, k)


@pytest.mark.parametrize("cfg_name", ["1x128_zinc", "4x256_zinc"])
def test_greedy_decode_tokens(cfg_name):
    g = load("decode_" + cfg_name)
    P = oracle_params(cfg_name)
    z = torch.from_numpy(g["z"])
    n = 6 if cfg_name.startswith("4x") else 12          # keep the CPU suite short
    assert np.array_equal(O.predict_mask_length(P, z).numpy(), g["mask_len"].astype(np.int64))
    ids, gaps = O.greedy_decode(P, z[:n], margins=True)
    assert np.array_equal(ids.numpy(), g["ids"][:n].astype(np.int64))
    np.testing.assert_allclose(gaps.numpy(), g["margins"][:n], atol=2e-4)
    char_dict = synth.load_vocab("zinc")[0]
    cond = [char_dict[str(c)] for c in g["cond"]]
    ids_c = O.greedy_decode(P, z[:3], condition_ids=cond)
    assert np.array_equal(ids_c.numpy(), g["ids_cond"][:3].astype(np.int64))
    if cfg_name.startswith("1x"):
        sm = O.length_mask(torch.from_numpy(g["mask_lens_explicit"].astype(np.int64)))
        ids_m = O.greedy_decode(P, z[8:16], src_mask=sm)
        assert np.array_equal(ids_m.numpy(), g["ids_mask"].astype(np.int64))


def test_attention_maps_quirk():
    g = load("attn_1x128_zinc")
    P = oracle_params("1x128_zinc")
    toks = torch.from_numpy(g["tokens"].astype(np.int64))
    tgt, src_mask, tgt_mask = O.make_masks(toks)
    enc, fin, src = O.attention_maps(P, toks, src_mask, tgt, tgt_mask, eps=torch.from_numpy(g["eps"]))
    heads = g["heads"].tolist()
    for i in range(3):
        np.testing.assert_allclose(enc[i][:, heads].numpy(), g["enc"][i].astype(np.float32), atol=1e-3)
        np.testing.assert_allclose(src[i][:, heads].numpy(), g["src"][i].astype(np.float32), atol=1e-3)
    np.testing.assert_allclose(fin[0][:, heads].numpy(), g["fin"].astype(np.float32), atol=1e-3)


def test_loss_units_and_schedules():
    g = load("units")
    toks = torch.from_numpy(g["tokens"].astype(np.int64))
    w = torch.tensor(synth.load_vocab("zinc")[2], dtype=torch.float)
    true_len = (toks != 24).sum(-1)
    for tag, mu_key, prop in (("plain", "mu", False), ("prop", "mu", True), ("nan", "mu_nan", False)):
        ins = [torch.from_numpy(g[k]).clone().requires_grad_(True) for k in ("logits", mu_key, "logvar", "pred_len")]
        out = O.trans_vae_loss(toks, ins[0], ins[1], ins[2], true_len, ins[3], torch.from_numpy(g["true_prop"]),
                               torch.from_numpy(g["pred_prop"]) if prop else None, w, float(g["beta"]))
        out[0].backward()
        np.testing.assert_allclose(np.array([float(o) for o in out]), g["loss_" + tag], rtol=1e-6)
        np.testing.assert_allclose(ins[0].grad.numpy(), g["dlogits_" + tag], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(ins[3].grad.numpy(), g["dpred_len_" + tag], rtol=1e-5, atol=1e-9)
        dmu = np.zeros_like(g["mu"]) if ins[1].grad is None else ins[1].grad.numpy()
        np.testing.assert_allclose(dmu, g["dmu_" + tag], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose([O.kl_anneal(e, 1e-8, 0.05, 100, 0) for e in range(0, 120, 7)], g["anneal_0_100"], rtol=1e-14)
    np.testing.assert_allclose([O.kl_anneal(e, 0.001, 0.05, 60, 10) for e in range(0, 80, 5)], g["anneal_10_60"], rtol=1e-14)
    np.testing.assert_allclose([O.noam_rate(int(s), 256) for s in g["noam_steps"]], g["noam_256"], rtol=1e-14)
    np.testing.assert_allclose([O.noam_rate(int(s), 128) for s in g["noam_steps"]], g["noam_128"], rtol=1e-14)
    org = synth.load_vocab("zinc")[1]
    assert O.decode_mols(torch.from_numpy(g["decode_ids"].astype(np.int64)), org) == [str(s) for s in g["decode_strs"]]
    np.testing.assert_allclose(O.calc_entropy(g["entropy_sample"]), g["entropy"], rtol=1e-12)
