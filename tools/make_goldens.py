#!/usr/bin/env python3
"""Capture golden vectors from the REAL reference (container only; never runs on the GPU box).

Imports /root/reference/transvae with rdkit/seaborn stubbed (both absent here and untouched
by the hot path), loads the synthetic checkpoints of transvae_amd.synth into
the reference's own nn.Modules, runs its forward / loss / backward / NoamOpt step /
greedy_decode / forward_w_attn and writes small .npz fixtures into tests/golden/.

    python tools/make_goldens.py            # regenerates tests/golden/*.npz

The fixtures are data (inputs + expected outputs).  No reference source is copied.
"""
import os
import sys
import types
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    for name in ["rdkit", "rdkit.Chem", "rdkit.Chem.AllChem", "rdkit.Chem.Descriptors",
                 "rdkit.rdBase", "seaborn"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["rdkit.rdBase"].DisableLog = lambda *_: None
    sys.modules["rdkit.Chem.AllChem"].GetMorganFingerprintAsBitVect = None
    sys.modules["rdkit"].rdBase = sys.modules["rdkit.rdBase"]
    sys.modules["rdkit"].Chem = sys.modules["rdkit.Chem"]
    sys.modules["rdkit.Chem"].Descriptors = sys.modules["rdkit.Chem.Descriptors"]
    sys.modules["rdkit.Chem"].AllChem = sys.modules["rdkit.Chem.AllChem"]
    sys.path.insert(0, REF)
    import transvae.trans_models as tm
    import transvae.loss as loss
    import transvae.data as data
    import transvae.tvae_util as util
    import transvae.opt as opt
    return tm, loss, data, util, opt


def probe_vec(key, n):
    """Seeded N(0,1) probe: the golden stores <grad, probe> per tensor."""
    rng = np.random.default_rng(zlib.crc32(("probe:" + key).encode()))
    return rng.standard_normal(n)


FULL_GRAD_KEYS = [
    "encoder.layers.0.sublayer.0.norm.a_2", "encoder.layers.0.sublayer.0.norm.b_2",
    "decoder.layers.2.sublayer.2.norm.a_2", "decoder.layers.2.sublayer.2.norm.b_2",
    "decoder.norm.a_2", "encoder.norm.b_2", "generator.proj.weight", "generator.proj.bias",
    "encoder.z_means.bias", "encoder.z_var.bias",
    "encoder.conv_bottleneck.conv_layers.0.0.bias", "encoder.conv_bottleneck.conv_layers.2.0.bias",
    "decoder.deconv_bottleneck.deconv_layers.2.0.bias", "decoder.deconv_bottleneck.deconv_layers.0.0.bias",
    "src_embed.0.lut.weight", "tgt_embed.0.lut.weight",
    "encoder.layers.1.self_attn.linears.0.bias", "decoder.layers.1.src_attn.linears.2.bias",
    "encoder.predict_len2.bias", "decoder.linear.bias",
]

CONFIGS = {
    "1x128_zinc": dict(arch="trans1x-128", vocab="zinc", pp=False),
    "4x256_zinc": dict(arch="trans4x-256", vocab="zinc", pp=False),
    "4x256_pubchem_pp": dict(arch="trans4x-256", vocab="pubchem", pp=True),
}


def build(tm, cfg, seed=0, **extra):
    from transvae_amd import synth
    char_dict, org_dict, w = synth.load_vocab(cfg["vocab"])
    params = {"CHAR_DICT": char_dict, "ORG_DICT": org_dict, "CHAR_WEIGHTS": w}
    params.update(extra)
    vae = tm.TransVAE(params=params, name="golden", property_predictor=cfg["pp"],
                      **synth.ARCHS[cfg["arch"]])
    sd = synth.synth_state_dict(vae.model.state_dict(), seed)
    vae.model.load_state_dict(sd)
    return vae


def step_inputs(cfg_name, B=12, trial=0):
    from transvae_amd import synth
    lens = [1, 2, 126, 125, 35, 40, 22, 60, 90, 33, 47, 10][:B]
    toks = synth.synth_tokens(B, seed=(zlib.crc32(cfg_name.encode()) + 7919 * trial) & 0xFFFF, lengths=lens)
    rng = np.random.default_rng(7 + trial)
    eps = torch.from_numpy(rng.standard_normal((B, 128)).astype(np.float32))
    props = torch.from_numpy(rng.uniform(0, 3, B).astype(np.float32))
    return toks, props, eps


def run_chunk(tm, loss_mod, data_mod, vae, toks, props, eps, beta, dtype):
    """One chunk exactly as VAEShell.train does it (trans_models.py:211-236), eval mode."""
    model = vae.model
    model.to(dtype)
    model.eval()
    model.zero_grad()
    src = toks.long()
    tgt = toks[:, :-1].long()
    src_mask = (src != vae.pad_idx).unsqueeze(-2)
    tgt_mask = data_mod.make_std_mask(tgt, vae.pad_idx)
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: eps.to(t.dtype)
    try:
        x_out, mu, logvar, pred_len, pred_prop = model(src, tgt, src_mask, tgt_mask)
    finally:
        torch.randn_like = orig
    true_len = src_mask.sum(dim=-1)
    w = vae.params["CHAR_WEIGHTS"].to(dtype)
    out = loss_mod.trans_vae_loss(src, x_out, mu, logvar, true_len, pred_len,
                                  props.to(dtype), pred_prop, w, beta)
    out[0].backward()
    return out, x_out, mu, logvar, pred_len, pred_prop


def gate_margins(vae, toks, eps):
    """Per-molecule distance of every gate from its switching point, measured on the REFERENCE in float64 with forward
    hooks: max-pool pairs |u - v| and ReLU inputs of the conv stack, the ReLU inputs of decoder.linear and of the three
    deconv layers, and the hidden ReLU inputs of the seven feed-forward blocks (w_1 outputs: 910,000 gates per
    trans4x-256 molecule).  ReLU / MaxPool make the gradient discontinuous there: two correct fp32 implementations can
    legitimately pick different sides when a margin is below their rounding error, so golden inputs are chosen with
    safe margins.  Margins are RELATIVE to the layer's rms pre-activation over the batch (an fp32 product's error
    scales with the size of its terms), the convention of oracle.gate_margins and tests/helpers.GATE_THRESHOLDS.
    Returns {name: (B,) tensor}."""
    model = vae.model
    model.double()
    model.eval()
    m = {}
    hooks = []

    def conv_hook(i):
        def f(mod, inp, out):
            L = out.shape[-1] // 2 * 2
            u, v = out[..., 0:L:2], out[..., 1:L:2]
            rms = out.pow(2).mean().sqrt()
            m["pool%d" % i] = (u - v).abs().flatten(1).min(1)[0] / rms
            m["crelu%d" % i] = torch.maximum(u, v).abs().flatten(1).min(1)[0] / rms
        return f

    def relu_hook(name):
        def f(mod, inp, out):
            m[name] = out.abs().flatten(1).min(1)[0] / out.pow(2).mean().sqrt()
        return f

    for i, seq in enumerate(model.encoder.conv_bottleneck.conv_layers):
        hooks.append(seq[0].register_forward_hook(conv_hook(i)))
    hooks.append(model.decoder.linear.register_forward_hook(relu_hook("dlin")))
    for i, seq in enumerate(model.decoder.deconv_bottleneck.deconv_layers):
        hooks.append(seq[0].register_forward_hook(relu_hook("deconv%d" % i)))
    for mod_name, mod in model.named_modules():
        if mod_name.endswith("feed_forward.w_1"):
            hooks.append(mod.register_forward_hook(relu_hook("ffn:" + mod_name[:-len(".w_1")])))
    src = toks.long()
    tgt = toks[:, :-1].long()
    src_mask = (src != vae.pad_idx).unsqueeze(-2)
    tgt_mask = (tgt != vae.pad_idx).unsqueeze(-2) & torch.tril(torch.ones(1, 126, 126, dtype=torch.bool))
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: eps.to(t.dtype)
    try:
        with torch.no_grad():
            for lo in range(0, toks.shape[0], 96):
                part = {}
                sl = slice(lo, lo + 96)
                cur_eps = eps[sl]
                torch.randn_like = lambda t, *a, **k: cur_eps.to(t.dtype)
                before = dict(m)
                model(src[sl], tgt[sl], src_mask[sl], tgt_mask[sl])
                for k in list(m):
                    part[k] = m[k]
                    m[k] = part[k] if lo == 0 else torch.cat([before[k], part[k]])
    finally:
        torch.randn_like = orig
        for h in hooks:
            h.remove()
    model.float()
    return m


ENC_GATES = ("pool0", "crelu0", "pool1", "crelu1", "pool2", "crelu2")
DEC_GATES = ("dlin", "deconv0", "deconv1", "deconv2")


def _gate_score(m, names):
    """(n,) smallest margin / threshold over the named gates (>= 1: every such gate of the molecule is outside its
    unsafe band).  Thresholds: tests/helpers.GATE_THRESHOLDS for the bottleneck, FFN_GATE_THRESHOLD for 'ffn:' gates."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import FFN_GATE_THRESHOLD, GATE_THRESHOLDS
    rows = [m[k] / (FFN_GATE_THRESHOLD if k.startswith("ffn:") else GATE_THRESHOLDS[k]) for k in names]
    return torch.stack(rows).min(0)[0]


def safe_step_inputs(vae, name, n_cand=48, lens=(1, 2, 126, 125, 35, 40, 22, 60, 90, 33, 47, 10), seed_shift=0):
    """Choose every molecule (tokens, then its eps row) among n_cand seeded candidates so that its own gate margins are
    as large as possible RELATIVE TO THEIR THRESHOLDS; gates of one molecule do not depend on the others.  Tokens decide
    the encoder side (conv bottleneck gates + the three encoder feed-forward blocks), eps -- with the tokens fixed --
    the decoder side (decoder.linear, deconv gates, the feed-forward blocks of final_encodes and the decoder layers).
    Round 3: the feed-forward gates are screened too (round 2 screened the bottleneck only and had to keep a 4e-5
    gradient floor: one marginal FFN ReLU deciding differently puts 2-3e-5 on the tensors upstream of it)."""
    from transvae_amd import synth
    B = len(lens)
    base = (zlib.crc32(name.encode()) + seed_shift) & 0xFFFF
    cand = torch.cat([synth.synth_tokens(n_cand, seed=base + 101 * i, lengths=[L] * n_cand) for i, L in enumerate(lens)])
    m = gate_margins(vae, cand, torch.zeros(B * n_cand, 128))
    enc_names = list(ENC_GATES) + [k for k in m if k.startswith("ffn:encoder.")]
    pick = _gate_score(m, enc_names).view(B, n_cand).argmax(1)
    toks = torch.stack([cand[i * n_cand + int(pick[i])] for i in range(B)])
    rng = np.random.default_rng(base + 5)
    eps_c = torch.from_numpy(rng.standard_normal((B, n_cand, 128)).astype(np.float32))
    m2 = gate_margins(vae, toks.repeat_interleave(n_cand, 0), eps_c.view(B * n_cand, 128))
    dec_names = list(DEC_GATES) + [k for k in m2 if k.startswith("ffn:decoder.")]
    pick2 = _gate_score(m2, dec_names).view(B, n_cand).argmax(1)
    eps = torch.stack([eps_c[i, int(pick2[i])] for i in range(B)])
    props = torch.from_numpy(rng.uniform(0, 3, B).astype(np.float32))
    final = gate_margins(vae, toks, eps)
    score = _gate_score(final, list(final))
    margins = {k: float(v.min()) for k, v in final.items()}
    print("  inputs for %s: smallest margin / threshold per molecule %s" % (name, ["%.1f" % float(v) for v in score]))
    print("  margins %s" % {k: "%.1e" % v for k, v in margins.items()})
    assert float(score.min()) >= 1.0, "a molecule of %s kept a gate inside its unsafe band: raise n_cand" % name
    return toks, props, eps, 0, margins


def golden_step(tm, loss_mod, data_mod, name, cfg):
    vae = build(tm, cfg)
    toks, props, eps, trial, margins = safe_step_inputs(vae, name)
    beta = 0.02
    rec = dict(tokens=toks.numpy().astype(np.int8), props=props.numpy(), eps=eps.numpy(),
               beta=np.float64(beta), trial=np.int64(trial),
               gate_margin_names=np.array(sorted(margins)), gate_margins=np.array([margins[k] for k in sorted(margins)]))
    g32 = {}
    for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        out, x_out, mu, logvar, pred_len, pred_prop = run_chunk(
            tm, loss_mod, data_mod, vae, toks, props, eps, beta, dtype)
        rec["loss_" + tag] = np.array([float(o) for o in out], dtype=np.float64)
        if tag == "f32":
            rec["mu"] = mu.detach().numpy()
            rec["logvar"] = logvar.detach().numpy()
            rec["pred_len"] = pred_len.detach().numpy()
            rec["logits_rows"] = x_out.detach()[[0, 2, 5]].numpy()
            if pred_prop is not None:
                rec["pred_prop"] = pred_prop.detach().numpy()
        keys, norms, sums, dots, relerr = [], [], [], [], []
        for k, p in vae.model.named_parameters():
            g = p.grad.detach().double().numpy().ravel()
            keys.append(k)
            norms.append(np.sqrt((g * g).sum()))
            sums.append(g.sum())
            dots.append(float(g @ probe_vec(k, g.size)))
            if tag == "f32":
                g32[k] = g.copy()
            else:   # the reference's OWN fp32 rounding error on this tensor (rel. L2 vs its fp64 run)
                relerr.append(np.linalg.norm(g32[k] - g) / max(np.linalg.norm(g), 1e-300))
        rec["grad_keys"] = np.array(keys)
        rec["grad_norm_" + tag] = np.array(norms)
        rec["grad_sum_" + tag] = np.array(sums)
        rec["grad_dot_" + tag] = np.array(dots)
        if tag == "f64":
            rec["grad_ref_f32_relerr"] = np.array(relerr)
            named = dict(vae.model.named_parameters())
            for k in FULL_GRAD_KEYS:
                rec["grad_full/" + k] = named[k].grad.detach().numpy().astype(np.float64)
    vae.model.float()
    np.savez_compressed(os.path.join(OUT, "step_%s.npz" % name), **rec)
    nz = rec["grad_norm_f64"] > 1e-10
    print("step", name, rec["loss_f32"], rec["loss_f64"], "ref fp32-vs-fp64 grad rel-L2: median %.2e max %.2e"
          % (np.median(rec["grad_ref_f32_relerr"][nz]), rec["grad_ref_f32_relerr"][nz].max()))


ACCUM_KEYS = ["generator.proj.bias", "encoder.norm.a_2", "encoder.z_means.bias",
              "decoder.layers.0.self_attn.linears.3.bias", "src_embed.0.lut.weight",
              "encoder.conv_bottleneck.conv_layers.1.0.bias", "decoder.linear.bias",
              "decoder.layers.2.feed_forward.w_2.bias", "encoder.predict_len1.bias",
              "decoder.deconv_bottleneck.deconv_layers.1.0.bias"]


def golden_accum(tm, loss_mod, data_mod):
    """2 chunks accumulated, NoamOpt.step, 3 steps (trans1x-128)."""
    from transvae_amd import synth
    name, cfg = "1x128_zinc", CONFIGS["1x128_zinc"]
    vae = build(tm, cfg, WARMUP_STEPS=100)   # short warm-up so that updates are well above fp32 eps
    keys = ACCUM_KEYS
    rec = {"keys": np.array(keys)}
    rng = np.random.default_rng(11)
    all_toks, all_eps, rates, losses = [], [], [], []
    named = dict(vae.model.named_parameters())
    for step in range(3):
        vae.model.zero_grad()
        for c in range(2):
            toks = synth.synth_tokens(6, seed=100 + 10 * step + c)
            eps = torch.from_numpy(rng.standard_normal((6, 128)).astype(np.float32))
            props = torch.zeros(6)
            all_toks.append(toks.numpy().astype(np.int8))
            all_eps.append(eps.numpy())
            model = vae.model
            model.eval()
            src = toks.long()
            tgt = toks[:, :-1].long()
            src_mask = (src != vae.pad_idx).unsqueeze(-2)
            tgt_mask = data_mod.make_std_mask(tgt, vae.pad_idx)
            orig = torch.randn_like
            torch.randn_like = lambda t, *a, **k: eps.to(t.dtype)
            try:
                x_out, mu, logvar, pred_len, pred_prop = model(src, tgt, src_mask, tgt_mask)
            finally:
                torch.randn_like = orig
            out = loss_mod.trans_vae_loss(src, x_out, mu, logvar, src_mask.sum(dim=-1), pred_len,
                                          props, pred_prop, vae.params["CHAR_WEIGHTS"], 0.01)
            losses.append(float(out[0]))
            out[0].backward()
        vae.optimizer.step()
        rates.append(vae.optimizer.state_dict["rate"])
        for k in keys:
            rec["p_step%d/%s" % (step + 1, k)] = named[k].detach().numpy().copy()
    rec.update(tokens=np.stack(all_toks), eps=np.stack(all_eps), rates=np.array(rates),
               losses=np.array(losses), beta=np.float64(0.01), warmup=np.int64(100))
    np.savez_compressed(os.path.join(OUT, "accum_1x128_zinc.npz"), **rec)
    print("accum rates", rates)


def golden_accum_4x256(tm, loss_mod, data_mod, chunks=5, csize=6, steps=3):
    """The headline model's optimizer step as VAEShell.train runs it (trans_models.py:202-238, opt.py:21-36): `chunks`
    chunks of `csize` molecules, gradients ACCUMULATED over the chunks (no 1/chunks), then NoamOpt.step; `steps` steps.
    Run twice on the reference -- float32 (what it really does) and float64 (the truth) -- so that the test can hold the
    build's parameter updates to the reference's OWN fp32-vs-fp64 error instead of a guessed bar.  Inputs are chosen
    gate-safe for step 1 (the weights move by ~1e-7 relative per step, margins stay what they are to three digits)."""
    name, cfg = "4x256_zinc", CONFIGS["4x256_zinc"]
    warmup = 100
    lens = (35, 28, 41, 33, 37, 126, 30, 44, 36, 22, 39, 31, 34, 2, 38, 29, 42, 35, 33, 40, 27, 36, 60, 32, 45, 34, 31, 38, 90, 30)
    assert len(lens) == chunks * csize
    from transvae.opt import NoamOpt
    rec = {"keys": np.array(ACCUM_KEYS), "chunks": np.int64(chunks), "csize": np.int64(csize), "warmup": np.int64(warmup),
           "beta": np.float64(0.01)}
    inputs, step_margins = [], []

    def run(tag, dtype):
        vae = build(tm, cfg, WARMUP_STEPS=warmup)
        vae.model.to(dtype)
        adam = torch.optim.Adam(vae.model.parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9)     # as build_model (:703-705)
        vae.optimizer = NoamOpt(256, vae.params["LR_SCALE"], warmup, adam)
        named = dict(vae.model.named_parameters())
        for k in ACCUM_KEYS:
            rec["p0/%s" % k] = named[k].detach().double().numpy().copy()
        losses, rates = [], []
        for step in range(steps):
            if tag == "f32":
                # The inputs of step t are chosen gate-safe on the weights AS THEY ARE at step t: with the short warm-up the
                # weights move by lr = 6e-5 .. 2e-4 per element and step, which re-draws every pre-activation's distance
                # from zero (margins measured on the initial weights say nothing about steps 2 and 3).  gate_margins runs
                # the model in float64 and puts it back to float32: exact for float32 weights.
                toks, _, eps, _, margins = safe_step_inputs(vae, "accum4x256_step%d" % step, n_cand=48, lens=lens, seed_shift=977 * step)
                inputs.append((toks, eps))
                step_margins.append(min(margins.values()))
            toks_all, eps_all = inputs[step]
            vae.model.zero_grad()
            for c in range(chunks):
                sl = slice(c * csize, (c + 1) * csize)
                toks, eps = toks_all[sl], eps_all[sl]
                model = vae.model
                model.eval()
                src, tgt = toks.long(), toks[:, :-1].long()
                src_mask = (src != vae.pad_idx).unsqueeze(-2)
                tgt_mask = data_mod.make_std_mask(tgt, vae.pad_idx)
                orig = torch.randn_like
                torch.randn_like = lambda t, *a, **k: eps.to(t.dtype)
                try:
                    x_out, mu, logvar, pred_len, pred_prop = model(src, tgt, src_mask, tgt_mask)
                finally:
                    torch.randn_like = orig
                out = loss_mod.trans_vae_loss(src, x_out, mu, logvar, src_mask.sum(dim=-1), pred_len,
                                              torch.zeros(csize, dtype=dtype), pred_prop,
                                              vae.params["CHAR_WEIGHTS"].to(dtype), 0.01)
                losses.append(float(out[0].detach()))
                out[0].backward()
            vae.optimizer.step()
            rates.append(vae.optimizer.state_dict["rate"])
            for k in ACCUM_KEYS:
                rec["p_%s_step%d/%s" % (tag, step + 1, k)] = named[k].detach().double().numpy().copy()
        rec["losses_" + tag] = np.array(losses)
        rec["rates"] = np.array(rates)

    run("f32", torch.float32)       # what the reference really does; picks the inputs step by step
    run("f64", torch.float64)       # the truth, on the same inputs
    rec["tokens"] = np.stack([t.numpy().astype(np.int8) for t, _ in inputs])
    rec["eps"] = np.stack([e.numpy() for _, e in inputs])
    rec["smallest_relative_gate_margin_per_step"] = np.array(step_margins)
    # the reference's own fp32 error on every recorded UPDATE (p_step - p0), relative L2 against its float64 run
    errs = []
    for step in range(1, steps + 1):
        for k in ACCUM_KEYS:
            d32 = rec["p_f32_step%d/%s" % (step, k)] - rec["p0/%s" % k]
            d64 = rec["p_f64_step%d/%s" % (step, k)] - rec["p0/%s" % k]
            errs.append(np.linalg.norm(d32 - d64) / np.linalg.norm(d64))
    rec["update_ref_f32_relerr"] = np.array(errs).reshape(steps, len(ACCUM_KEYS))
    np.savez_compressed(os.path.join(OUT, "accum_4x256_zinc.npz"), **rec)
    print("accum 4x256: rates", rec["rates"], "reference fp32-vs-fp64 update rel-L2: median %.2e max %.2e" % (np.median(errs), np.max(errs)))


def golden_decode(tm, name, cfg, n=32):
    vae = build(tm, cfg)
    vae.use_gpu = False
    # shift the length head so predicted mask lengths are spread out (incl. a 0 case):
    # the bias of predict_len2 is part of the synthetic checkpoint, so the build sees it too.
    z = torch.from_numpy(np.random.default_rng(3).standard_normal((n, 128)).astype(np.float32))
    with torch.no_grad():
        mask_len = vae.model.encoder.predict_mask_length(z).squeeze(-1)
        ids = vae.greedy_decode(z)
        ids_cond = vae.greedy_decode(z[:8], condition=["C", "c"])
        lens = torch.tensor([0, 1, 5, 17, 40, 77, 126, 127])
        sm = (torch.arange(127)[None, :] < lens[:, None]).float().unsqueeze(1)
        ids_mask = vae.greedy_decode(z[8:16], src_mask=sm)
    # per-step top-2 logit margin, from a straight-line replay of the same loop
    gaps = np.zeros((n, 125), dtype=np.float32)
    with torch.no_grad():
        src_mask = torch.zeros(n, 1, 127)
        for i in range(n):
            src_mask[i, :, :int(mask_len[i])] = 1
        dec = torch.zeros(n, 1, dtype=torch.long)
        vae.model.eval()
        for i in range(125):
            out = vae.model.generator(vae.model.decode(z, src_mask, dec, tm.subsequent_mask(dec.size(1)).long()))
            t2 = torch.topk(out[:, i, :], 2, dim=-1)[0]
            gaps[:, i] = (t2[:, 0] - t2[:, 1]).numpy()
            nxt = torch.max(out[:, i, :], dim=1)[1] + 1
            assert torch.equal(nxt, ids[:, i]), "replay diverged"
            dec = torch.cat([dec, nxt.unsqueeze(1)], 1)
    np.savez_compressed(os.path.join(OUT, "decode_%s.npz" % name), z=z.numpy(),
                        mask_len=mask_len.numpy().astype(np.int16), ids=ids.numpy().astype(np.int8),
                        ids_cond=ids_cond.numpy().astype(np.int8), cond=np.array(["C", "c"]),
                        mask_lens_explicit=lens.numpy().astype(np.int16),
                        ids_mask=ids_mask.numpy().astype(np.int8), margins=gaps)
    print("decode", name, "mask_len", mask_len.tolist(), "min margin", gaps.min(),
          "distinct tokens", len(np.unique(ids.numpy())))


def golden_attn(tm, data_mod, name="1x128_zinc"):
    "forward_w_attn dumps; 1x128_zinc pins dk = 32, 4x256_pubchem_pp is BASELINE config 5's model (dk = 64, property head)"
    cfg = CONFIGS[name]
    vae = build(tm, cfg)
    from transvae_amd import synth
    toks = synth.synth_tokens(2, seed=5, lengths=[30, 126])
    eps = torch.from_numpy(np.random.default_rng(9).standard_normal((2, 128)).astype(np.float32))
    src, tgt = toks.long(), toks[:, :-1].long()
    src_mask = (src != vae.pad_idx).unsqueeze(-2)
    tgt_mask = data_mod.make_std_mask(tgt, vae.pad_idx)
    vae.model.eval()
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: eps.to(t.dtype)
    try:
        with torch.no_grad():
            mem, mu, logvar, pred_len, enc_w = vae.model.encoder.forward_w_attn(vae.model.src_embed(src), src_mask)
            x, fin_w, src_w = vae.model.decoder.forward_w_attn(vae.model.tgt_embed(tgt), mem, src_mask, tgt_mask)
    finally:
        torch.randn_like = orig
    heads = [0, 3]
    np.savez_compressed(os.path.join(OUT, "attn_%s.npz" % name), tokens=toks.numpy().astype(np.int8),
                        eps=eps.numpy(), heads=np.array(heads),
                        enc=np.stack([w[:, heads].numpy() for w in enc_w]).astype(np.float16),
                        fin=fin_w[0][:, heads].numpy().astype(np.float16),
                        src=np.stack([w[:, heads].numpy() for w in src_w]).astype(np.float16),
                        x_rows=x[:, :4].numpy())
    print("attn", name, [tuple(w.shape) for w in enc_w], tuple(fin_w[0].shape), [tuple(w.shape) for w in src_w])


TINY = dict(N=1, d_model=64, d_ff=64, d_latent=32, h=2)      # smallest architecture the reference's conv stack allows


def golden_ckpt(tm, loss_mod, data_mod):
    """Checkpoint interop, both directions (reference trans_models.py:69-124, opt.py:17-19, 38-39).

    (1) The REFERENCE builds a tiny model (N=1, d_model 64, 2 heads: the .ckpt is 7 MB, most of it the two (1,5000,64)
        positional buffers the format carries), takes one accumulate + NoamOpt.step, fills current_state the way its
        train() does (np.float64 best_loss included -- the value that makes torch >= 2.6 refuse the file by default) and
        writes tests/golden/ref_tiny.ckpt with ITS OWN VAEShell.save.  ref_ckpt.npz holds what the reference decodes from
        that state, its Noam step/rate and checksums of every tensor.
    (2) The BUILD (transvae_amd, CPU side: no kernel runs) writes a checkpoint with its VAEShell.save; the reference's
        VAEShell.load reads it (torch.load patched to weights_only=False: the file was written two lines above).  Recorded:
        that it loaded, the state_dict keys it saw, and that every tensor arrived bit for bit."""
    import tempfile
    from transvae_amd import synth
    from transvae_amd.trans_models import TransVAE as MyTransVAE
    cd, od, w = synth.load_vocab("zinc")
    vae = tm.TransVAE(params={"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w, "WARMUP_STEPS": 100}, name="tiny", **TINY)
    vae.model.load_state_dict(synth.synth_state_dict(vae.model.state_dict(), 3))
    vae.use_gpu = False
    toks = synth.synth_tokens(6, seed=77, lengths=[3, 9, 14, 20, 25, 28])    # length classes < d_latent = 32
    eps = torch.from_numpy(np.random.default_rng(5).standard_normal((6, TINY["d_latent"])).astype(np.float32))
    run_chunk(tm, loss_mod, data_mod, vae, toks, torch.zeros(6), eps, 0.01, torch.float32)
    vae.optimizer.step()
    vae.n_epochs = 1
    vae.best_loss = np.float64(3.25)
    vae.current_state.update(epoch=vae.n_epochs, model_state_dict=vae.model.state_dict(),
                             optimizer_state_dict=vae.optimizer.state_dict, best_loss=vae.best_loss)
    vae.save(vae.current_state, "ref", path=OUT)                      # -> tests/golden/ref_tiny.ckpt
    z = torch.from_numpy(np.random.default_rng(8).standard_normal((8, TINY["d_latent"])).astype(np.float32))
    with torch.no_grad():
        ids = vae.greedy_decode(z)
        mask_len = vae.model.encoder.predict_mask_length(z).squeeze(-1)
    sd = vae.model.state_dict()
    rec = dict(z=z.numpy(), ids=ids.numpy().astype(np.int8), mask_len=mask_len.numpy().astype(np.int16),
               noam_step=np.int64(vae.optimizer.state_dict["step"]), noam_rate=np.float64(vae.optimizer.state_dict["rate"]),
               keys=np.array(list(sd)), sums=np.array([float(v.double().sum()) for v in sd.values()]),
               best_loss=np.float64(vae.best_loss), epoch=np.int64(1),
               opt_state_entries=np.int64(len(vae.optimizer.state_dict["state"])))
    # ---- the other way round
    mine = MyTransVAE(params={"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w}, name="mine", **TINY)
    mine.model.load_state_dict(synth.synth_state_dict(mine.model.state_dict(), 4))
    mine.n_epochs = 7
    mine.optimizer.sync_state_dict()
    mine.current_state.update(epoch=7, model_state_dict=mine.model.state_dict(),
                              optimizer_state_dict=mine.optimizer.state_dict, best_loss=1.5)
    with tempfile.TemporaryDirectory() as tmp:
        mine.save(mine.current_state, "build", path=tmp)
        path = os.path.join(tmp, "build_mine.ckpt")
        orig = torch.load
        torch.load = lambda *a, **k: orig(*a, **dict(k, weights_only=False))
        try:
            back = tm.TransVAE(load_fn=path)
        finally:
            torch.load = orig
    bsd, msd = back.model.state_dict(), mine.model.state_dict()
    rec["reverse_loaded"] = np.bool_(True)
    rec["reverse_keys"] = np.array(list(bsd))
    rec["reverse_keys_equal"] = np.bool_(list(bsd) == list(msd))
    rec["reverse_tensors_equal"] = np.bool_(all(torch.equal(bsd[k], msd[k].cpu()) for k in msd))
    rec["reverse_epoch"] = np.int64(back.n_epochs)
    rec["reverse_arch"] = np.array([back.params[k] for k in ("N", "d_model", "d_ff", "d_latent", "h")])
    np.savez_compressed(os.path.join(OUT, "ref_ckpt.npz"), **rec)
    print("ckpt", os.path.getsize(os.path.join(OUT, "ref_tiny.ckpt")), "bytes; reverse:", bool(rec["reverse_keys_equal"]),
          bool(rec["reverse_tensors_equal"]), "mask_len", mask_len.tolist())


def golden_sampling(tm, opt_mod):
    """Seeded host-side draws of VAEShell.sample_from_memory (trans_models.py:355-381) in its three modes, and three
    steps of AdamOpt (opt.py:41-52) over torch.optim.Adam on small tensors with given gradients."""
    vae = build(tm, CONFIGS["1x128_zinc"])
    rec = {}
    dims = [3, 17, 90]
    pool = [1, 5, 9, 33, 64, 100, 127]
    torch.manual_seed(123)
    np.random.seed(5)
    rec["rand"] = vae.sample_from_memory(7, mode="rand").numpy()
    rec["top_dims"] = vae.sample_from_memory(7, mode="top_dims", sample_dims=dims).numpy()
    rec["k_dims"] = vae.sample_from_memory(7, mode="k_dims", sample_dims=pool, k=3).numpy()
    rec["k_dims_again"] = vae.sample_from_memory(5, mode="k_dims", sample_dims=pool, k=5).numpy()
    rec.update(dims=np.array(dims), pool=np.array(pool), torch_seed=np.int64(123), numpy_seed=np.int64(5))
    # AdamOpt: lr 3e-4, torch defaults (betas 0.9/0.999, eps 1e-8)
    rng = np.random.default_rng(31)
    shapes = [(25,), (40, 17), (4100,), (1,)]
    ps = [torch.nn.Parameter(torch.from_numpy(rng.standard_normal(sh).astype(np.float32))) for sh in shapes]
    for i, p0 in enumerate(ps):
        rec["adamopt_p0/%d" % i] = p0.detach().numpy().copy()
    ao = opt_mod.AdamOpt(ps, 3e-4, torch.optim.Adam)
    for step in range(1, 4):
        for i, pr in enumerate(ps):
            g = torch.from_numpy(rng.standard_normal(shapes[i]).astype(np.float32))
            rec["adamopt_g%d/%d" % (step, i)] = g.numpy()
            pr.grad = g
        ao.step()
        for i, pr in enumerate(ps):
            rec["adamopt_p%d/%d" % (step, i)] = pr.detach().numpy().copy()
    rec["adamopt_lr"] = np.float64(3e-4)
    rec["adamopt_n"] = np.int64(len(shapes))
    np.savez_compressed(os.path.join(OUT, "sampling.npz"), **rec)
    print("sampling", rec["rand"][0, :3], np.nonzero(rec["k_dims"][0])[0], np.nonzero(rec["k_dims_again"][0])[0])



def golden_units(tm, loss_mod, util, opt_mod):
    rng = np.random.default_rng(21)
    rec = {}
    # loss on random tensors incl. an all-pad row and a NaN-KLD case (loss.py:36-37)
    from transvae_amd import synth
    toks = synth.synth_tokens(5, seed=1, lengths=[0, 3, 126, 50, 12])
    logits = torch.from_numpy(rng.standard_normal((5, 126, 25)).astype(np.float32) * 3)
    mu = torch.from_numpy(rng.standard_normal((5, 128)).astype(np.float32))
    logvar = torch.from_numpy(rng.standard_normal((5, 128)).astype(np.float32))
    pred_len = torch.from_numpy(rng.standard_normal((5, 128)).astype(np.float32))
    true_len = (toks != 24).sum(-1)
    pp = torch.from_numpy(rng.uniform(0, 1, (5, 1)).astype(np.float32))
    tp = torch.from_numpy(rng.uniform(0, 1, 5).astype(np.float32))
    w = torch.tensor(synth.load_vocab("zinc")[2], dtype=torch.float)
    mu_nan = mu.clone()
    mu_nan[1, 7] = float("nan")
    rec["mu_nan"] = mu_nan.numpy()
    for tag, m_in, prop in (("plain", mu, None), ("prop", mu, pp), ("nan", mu_nan, None)):
        lv = logvar
        ins = [t.clone().requires_grad_(True) for t in (logits, m_in, lv, pred_len)]
        out = loss_mod.trans_vae_loss(toks, ins[0], ins[1], ins[2], true_len, ins[3], tp, prop, w, 0.3)
        out[0].backward()
        rec["loss_" + tag] = np.array([float(o) for o in out])
        rec["dlogits_" + tag] = ins[0].grad.numpy()
        rec["dmu_" + tag] = np.zeros_like(mu.numpy()) if ins[1].grad is None else ins[1].grad.numpy()
        rec["dlogvar_" + tag] = np.zeros_like(mu.numpy()) if ins[2].grad is None else ins[2].grad.numpy()
        rec["dpred_len_" + tag] = ins[3].grad.numpy()
    rec.update(tokens=toks.numpy().astype(np.int8), logits=logits.numpy(), mu=mu.numpy(), logvar=logvar.numpy(),
               pred_len=pred_len.numpy(), pred_prop=pp.numpy(), true_prop=tp.numpy(), beta=np.float64(0.3))
    # schedules
    ann = util.KLAnnealer(1e-8, 0.05, 100, 0)
    rec["anneal_0_100"] = np.array([ann(e) for e in range(0, 120, 7)])
    ann2 = util.KLAnnealer(0.001, 0.05, 60, 10)
    rec["anneal_10_60"] = np.array([ann2(e) for e in range(0, 80, 5)])
    no = opt_mod.NoamOpt(256, 1, 10000, torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=0))
    steps = [1, 2, 3, 100, 9999, 10000, 10001, 50000]
    rec["noam_steps"] = np.array(steps)
    rec["noam_256"] = np.array([no.rate(s) for s in steps])
    no2 = opt_mod.NoamOpt(128, 1, 10000, torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=0))
    rec["noam_128"] = np.array([no2.rate(s) for s in steps])
    # decode_mols
    _, org_dict, _ = synth.load_vocab("zinc")
    ids = torch.from_numpy(rng.integers(1, 26, (6, 125)))
    ids[0, 10] = 25
    ids[1, :] = 24
    ids[2, 0] = 25
    rec["decode_ids"] = ids.numpy().astype(np.int8)
    rec["decode_strs"] = np.array(util.decode_mols(ids, org_dict))
    # tokenizer + encode_smiles on real ZINC rows (reference example data; first 40)
    import pandas as pd
    smiles = pd.read_csv(os.path.join(REF, "notebooks/example_data/zinc_train_example.txt")).to_numpy()[:40, 1]
    char_dict = synth.load_vocab("zinc")[0]
    enc = np.stack([[0] + util.encode_smiles(util.tokenizer(s), 126, char_dict) for s in smiles])
    rec["smiles"] = np.array(list(smiles))
    rec["smiles_ids"] = enc.astype(np.int8)
    # entropy of a fixed sample (tvae_util.py:198-204)
    sample = rng.standard_normal((500, 6)) * np.array([0.01, 0.5, 1, 2, 3, 0.1])
    rec["entropy_sample"] = sample
    rec["entropy"] = util.calc_entropy(sample)
    np.savez_compressed(os.path.join(OUT, "units.npz"), **rec)
    print("units", rec["loss_plain"], rec["loss_nan"], rec["noam_256"][:3])


TRICKY_SMILES = [
    # strings the reference's tokenizer / encoder accept or refuse in ways a hand-written scanner can get wrong
    "", "C", "Cl", "Br", "B", "BrCl", "CCl", "ClC", "Clc1ccccc1Br", "C%12CC%12", "C%1", "%", "C%1a",
    "[nH]", "[", "[]", "[]]", "[C", "c1cc[nH]c1", "C[N+](C)(C)C", "[O-]C(=O)C", "[C@@H](N)O", "[nH", "]",
    "CX", "c1ccccc1 ", " C", "C\\C=C/C", "C\\", "H", "[H]", "C.C", "N#N", "C~C", "C:C", "C@C", "C?", "C>C",
    "C*", "C$C", "C_C", "_", "C[nH]]", "C[[nH]]", "Cc1nc2c([nH]1)c(=O)n(C)c(=O)n2C", "\u00e9", "C\u00e9", "[\u00e9]", "C\nC",
    "0123456789", "Si", "Se", "se", "as", "I", "F", "P", "p", "b", "(", ")", "=", "#", "-", "+", "/",
    "C" * 125, "C" * 126, "C" * 127, "Cl" * 126, "Cl" * 127, "[nH]" * 126, "C" * 200 + "X", "X" + "C" * 200,
]


def golden_data(data_mod, util):
    """The data pipeline on the reference's own example file (1000 ZINC rows) through ITS vae_data_gen, plus
    what its tokenizer / encode_smiles / row assignment do with tricky strings, one at a time (ids or the
    exception class) and in whole-column calls (which exception wins)."""
    import pandas as pd
    from transvae_amd import synth
    rec = {}
    smiles = pd.read_csv(os.path.join(REF, "notebooks/example_data/zinc_train_example.txt")).to_numpy()[:, 1:2]
    props = np.random.default_rng(17).uniform(-2, 5, len(smiles))
    for vocab in ("zinc", "pubchem"):
        char_dict = synth.load_vocab(vocab)[0]
        if vocab == "zinc":
            rows = data_mod.vae_data_gen(smiles, props, char_dict).numpy()
            rec["zinc_smiles"] = np.array([str(s) for s in smiles[:, 0]])
            rec["zinc_props"] = props
            rec["zinc_ids"] = rows[:, :127].astype(np.uint8)
            rec["zinc_prop_column"] = rows[:, 127]
            assert np.array_equal(rows[:, :127], rec["zinc_ids"])
        outcome, ids = [], []
        for smi in TRICKY_SMILES:
            try:
                row = data_mod.vae_data_gen(np.array([[smi]], dtype=object), None, char_dict).numpy()
                outcome.append("ok")
                ids.append(row[0, :127].astype(np.uint8))
            except Exception as e:                      # noqa: BLE001 -- the class is the datum
                outcome.append(type(e).__name__)
                ids.append(np.zeros(127, dtype=np.uint8))
        rec["tricky_outcome_" + vocab] = np.array(outcome)
        rec["tricky_ids_" + vocab] = np.stack(ids)
    rec["tricky_smiles"] = np.array(TRICKY_SMILES)
    # whole-column precedence: every string is tokenized before the first is encoded
    char_dict = synth.load_vocab("zinc")[0]
    cols = {"unknown_then_unmatched": ["CC", "C[N+]C", "CCX", "CC"], "long_then_unknown": ["C" * 130, "C[N+]C"],
            "unknown_and_long_in_one": ["[N+]" + "C" * 130], "long_then_unmatched": ["C" * 130, "C C"]}
    for k, col in cols.items():
        try:
            data_mod.vae_data_gen(np.array(col, dtype=object)[:, None], None, char_dict)
            rec["column_" + k] = np.array("ok")
        except Exception as e:                          # noqa: BLE001
            rec["column_" + k] = np.array(type(e).__name__)
        rec["column_input_" + k] = np.array(col)
    np.savez_compressed(os.path.join(OUT, "data.npz"), **rec)
    print("data", rec["zinc_ids"].shape, dict(zip(*np.unique(rec["tricky_outcome_zinc"], return_counts=True))),
          {k: str(rec["column_" + k]) for k in cols})


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    tm, loss_mod, data_mod, util, opt_mod = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "data":
        return golden_data(data_mod, util)
    if len(sys.argv) > 1 and sys.argv[1] == "accum":
        return golden_accum(tm, loss_mod, data_mod)
    if len(sys.argv) > 1 and sys.argv[1] == "accum4x256":
        return golden_accum_4x256(tm, loss_mod, data_mod)
    if len(sys.argv) > 1 and sys.argv[1] == "decode4x256":
        return golden_decode(tm, "4x256_zinc", CONFIGS["4x256_zinc"], n=256)
    if len(sys.argv) > 1 and sys.argv[1] == "ckpt":
        return golden_ckpt(tm, loss_mod, data_mod)
    if len(sys.argv) > 1 and sys.argv[1] == "sampling":
        return golden_sampling(tm, opt_mod)
    if len(sys.argv) > 1 and sys.argv[1] == "attn":
        golden_attn(tm, data_mod)
        return golden_attn(tm, data_mod, "4x256_pubchem_pp")
    if len(sys.argv) > 1 and sys.argv[1] == "step":
        for name, cfg in CONFIGS.items():
            golden_step(tm, loss_mod, data_mod, name, cfg)
        return
    for name, cfg in CONFIGS.items():
        golden_step(tm, loss_mod, data_mod, name, cfg)
    golden_accum(tm, loss_mod, data_mod)
    golden_accum_4x256(tm, loss_mod, data_mod)
    golden_decode(tm, "1x128_zinc", CONFIGS["1x128_zinc"])
    golden_decode(tm, "4x256_zinc", CONFIGS["4x256_zinc"], n=256)
    golden_attn(tm, data_mod)
    golden_attn(tm, data_mod, "4x256_pubchem_pp")
    golden_ckpt(tm, loss_mod, data_mod)
    golden_sampling(tm, opt_mod)
    golden_units(tm, loss_mod, util, opt_mod)
    golden_data(data_mod, util)


if __name__ == "__main__":
    main()
