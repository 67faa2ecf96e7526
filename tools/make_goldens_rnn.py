#!/usr/bin/env python3
"""Golden vectors of the RNN / RNNAttn models from the REAL reference (container only; never runs on the GPU box).

Imports /root/reference/transvae (as tools/make_goldens.py does), loads the synthetic checkpoints of transvae_amd.synth into
the reference's own RNNAttn / RNN modules (torch.nn.GRU inside), runs forward / vae_loss / backward / AdamOpt.step /
greedy_decode and writes small .npz fixtures into tests/golden/:

    python tools/make_goldens_rnn.py        # -> tests/golden/rnn_step_<cfg>.npz, rnn_decode_<cfg>.npz, rnn_adam_<cfg>.npz

The fixtures are data (inputs + expected outputs).  No reference source is copied.
"""
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_goldens import OUT, import_reference, probe_vec  # noqa: E402

CONFIGS = {
    "rnnattn256": dict(cls="RNNAttn", vocab="zinc", kw=dict(N=3, d_model=256, d_latent=128)),
    "rnn128_pp": dict(cls="RNN", vocab="pubchem", kw=dict(N=3, d_model=128, d_latent=128, property_predictor=True)),
    "rnnattn128_notf": dict(cls="RNNAttn", vocab="zinc", kw=dict(N=2, d_model=128, d_latent=128, tf=False)),
}
FULL_GRAD_KEYS = ["encoder.gru.weight_hh_l0", "encoder.gru.bias_hh_l1", "decoder.gru.bias_ih_l0", "decoder.gru.weight_ih_l1",
                  "encoder.norm.a_2", "decoder.norm.b_2", "generator.proj.bias", "encoder.z_means.bias", "encoder.attn.bias",
                  "decoder.unbottleneck.bias", "decoder.linear.bias"]


def build(rm, name, seed=0, **extra):
    from transvae_amd import synth
    cfg = CONFIGS[name]
    char_dict, org_dict, w = synth.load_vocab(cfg["vocab"])
    params = {"CHAR_DICT": char_dict, "ORG_DICT": org_dict, "CHAR_WEIGHTS": w}
    params.update(extra)
    vae = getattr(rm, cfg["cls"])(params=params, name="golden", **cfg["kw"])
    vae.model.load_state_dict(synth.synth_state_dict(vae.model.state_dict(), seed))
    return vae


def inputs(name, B=12):
    from transvae_amd import synth
    lens = [1, 2, 126, 125, 35, 40, 22, 60, 90, 33, 47, 10][:B]
    toks = synth.synth_tokens(B, seed=zlib.crc32(name.encode()) & 0xFFFF, lengths=lens)
    rng = np.random.default_rng(17)
    eps = torch.from_numpy(rng.standard_normal((B, 128)).astype(np.float32))
    props = torch.from_numpy(rng.uniform(0, 3, B).astype(np.float32))
    return toks, props, eps


def _follow_dtype(model, dtype):
    for m in (model.encoder, model.decoder):     # initH builds float32 zeros whatever the model's dtype: follow the dtype
        m.initH = (lambda mod: lambda b: torch.zeros(mod.n_layers, b, mod.size, dtype=dtype))(m)


def gate_margins(vae, toks, eps):
    """Per-molecule distance of every ReLU / MaxPool gate from its switching point, measured on the reference in float64:
    max-pool pairs |u - v| and ReLU inputs of the conv stack, the input of relu(bmm(attn_weights, mem)), the ReLU inputs of
    decoder.linear / decoder.unbottleneck and of the deconv layers.  Two correct fp32 implementations may legitimately
    pick different sides of a gate whose margin is below their rounding error (the gradient is discontinuous there), so
    the golden inputs are chosen with safe margins.  Returns {name: (B,) tensor}."""
    model = vae.model
    model.double()
    model.eval()
    _follow_dtype(model, torch.float64)
    m, hooks = {}, []

    def conv_hook(i):
        def f(mod, inp, out):
            L = out.shape[-1] // 2 * 2
            u, v = out[..., 0:L:2], out[..., 1:L:2]
            m["pool%d" % i] = (u - v).abs().flatten(1).min(1)[0]
            m["crelu%d" % i] = torch.maximum(u, v).abs().flatten(1).min(1)[0]
        return f

    def relu_hook(name):
        def f(mod, inp, out):
            m[name] = out.abs().flatten(1).min(1)[0]
        return f

    enc, dec = model.encoder, model.decoder
    if hasattr(enc, "conv_bottleneck"):
        for i, seq in enumerate(enc.conv_bottleneck.conv_layers):
            hooks.append(seq[0].register_forward_hook(conv_hook(i)))
        hooks.append(dec.linear.register_forward_hook(relu_hook("dlin")))
        for i, seq in enumerate(dec.deconv_bottleneck.deconv_layers):
            hooks.append(seq[0].register_forward_hook(relu_hook("deconv%d" % i)))
    else:
        hooks.append(dec.unbottleneck.register_forward_hook(relu_hook("dlin")))
    orig_bmm, orig_rl = torch.bmm, torch.randn_like

    def bmm(a, b):
        out = orig_bmm(a, b)
        m["bmm"] = out.abs().flatten(1).min(1)[0]
        return out
    torch.bmm = bmm
    torch.randn_like = lambda t, *a, **k: eps.to(t.dtype)
    try:
        with torch.no_grad():
            model(toks.long(), toks[:, :-1].long(), None, None)
    finally:
        torch.bmm, torch.randn_like = orig_bmm, orig_rl
        for h in hooks:
            h.remove()
    model.float()
    _follow_dtype(model, torch.float32)
    return m


def safe_inputs(vae, name, n_cand=24):
    "every molecule (tokens, then its eps row) chosen among n_cand seeded candidates for the largest gate margins"
    from transvae_amd import synth
    lens = [1, 2, 126, 125, 35, 40, 22, 60, 90, 33, 47, 10]
    B = len(lens)
    base = zlib.crc32(name.encode()) & 0xFFFF
    cand = torch.cat([synth.synth_tokens(n_cand, seed=base + 101 * i, lengths=[L] * n_cand) for i, L in enumerate(lens)])
    m = gate_margins(vae, cand, torch.zeros(B * n_cand, 128))
    # max-pool pairs are left out of the choice: over a run of padding the GRU state converges, neighbouring positions
    # (and the conv windows under them) become equal to 1e-9, and which of two equal inputs takes the gradient changes nothing
    enc_keys = [k for k in m if k.startswith(("crelu", "bmm"))]
    if enc_keys:
        pick = torch.stack([m[k] for k in enc_keys]).min(0)[0].view(B, n_cand).argmax(1)
    else:
        pick = torch.zeros(B, dtype=torch.long)
    toks = torch.stack([cand[i * n_cand + int(pick[i])] for i in range(B)])
    rng = np.random.default_rng(base + 5)
    eps_c = torch.from_numpy(rng.standard_normal((B, n_cand, 128)).astype(np.float32))
    m2 = gate_margins(vae, toks.repeat_interleave(n_cand, 0), eps_c.view(B * n_cand, 128))
    dec_keys = [k for k in m2 if k.startswith(("dlin", "deconv"))]
    pick2 = torch.stack([m2[k] for k in dec_keys]).min(0)[0].view(B, n_cand).argmax(1)
    eps = torch.stack([eps_c[i, int(pick2[i])] for i in range(B)])
    props = torch.from_numpy(rng.uniform(0, 3, B).astype(np.float32))
    final = {k: float(v.min()) for k, v in gate_margins(vae, toks, eps).items()}
    print("  inputs for %s: margins %s" % (name, {k: "%.1e" % v for k, v in final.items()}))
    return toks, props, eps, final


def run_chunk(loss_mod, vae, toks, props, eps, beta, dtype):
    "one chunk as VAEShell.train does it for model_type != 'transformer' (trans_models.py:211-236), eval mode"
    model = vae.model
    model.to(dtype)
    model.eval()
    model.zero_grad()
    _follow_dtype(model, dtype)
    src, tgt = toks.long(), toks[:, :-1].long()
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: eps.to(t.dtype)
    try:
        x_out, mu, logvar, pred_prop = model(src, tgt, None, None)
    finally:
        torch.randn_like = orig
    out = loss_mod.vae_loss(src, x_out, mu, logvar, props.to(dtype), pred_prop, vae.params["CHAR_WEIGHTS"].to(dtype), beta)
    out[0].backward()
    return out, x_out, mu, logvar, pred_prop


def golden_step(rm, loss_mod, name):
    vae = build(rm, name)
    toks, props, eps, margins = safe_inputs(vae, name)
    beta = 0.02
    rec = dict(tokens=toks.numpy().astype(np.int8), props=props.numpy(), eps=eps.numpy(), beta=np.float64(beta),
               gate_margin_names=np.array(sorted(margins)), gate_margins=np.array([margins[k] for k in sorted(margins)]))
    g32 = {}
    for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        out, x_out, mu, logvar, pred_prop = run_chunk(loss_mod, vae, toks, props, eps, beta, dtype)
        rec["loss_" + tag] = np.array([float(o) for o in out], dtype=np.float64)
        if tag == "f32":
            rec["mu"], rec["logvar"] = mu.detach().numpy(), logvar.detach().numpy()
            rec["logits_rows"] = x_out.detach()[[0, 2, 5]].numpy()
            if pred_prop is not None:
                rec["pred_prop"] = pred_prop.detach().numpy()
        keys, norms, sums, dots, relerr = [], [], [], [], []
        named = dict(vae.model.named_parameters())
        for k, p in named.items():
            if p.grad is None:                   # tgt_embed is never used (rnn_models.py:207)
                continue
            g = p.grad.detach().double().numpy().ravel()
            keys.append(k)
            norms.append(np.sqrt((g * g).sum()))
            sums.append(g.sum())
            dots.append(float(g @ probe_vec(k, g.size)))
            if tag == "f32":
                g32[k] = g.copy()
            else:
                relerr.append(np.linalg.norm(g32[k] - g) / max(np.linalg.norm(g), 1e-300))
        rec["grad_keys"] = np.array(keys)
        rec["grad_norm_" + tag], rec["grad_sum_" + tag], rec["grad_dot_" + tag] = np.array(norms), np.array(sums), np.array(dots)
        if tag == "f64":
            rec["grad_ref_f32_relerr"] = np.array(relerr)
            for k in FULL_GRAD_KEYS:
                if k in named and named[k].grad is not None:
                    rec["grad_full/" + k] = named[k].grad.detach().numpy().astype(np.float64)[:6]   # matrices: first 6 rows
    vae.model.float()
    np.savez_compressed(os.path.join(OUT, "rnn_step_%s.npz" % name), **rec)
    print("step", name, rec["loss_f32"], rec["loss_f64"], "ref fp32-vs-fp64 grad rel-L2: median %.2e max %.2e"
          % (np.median(rec["grad_ref_f32_relerr"]), rec["grad_ref_f32_relerr"].max()))


def golden_adam(rm, loss_mod, name):
    """3 optimizer steps of 2 accumulated chunks each (AdamOpt over torch.optim.Adam, rnn_models.py:105-106), eval mode,
    run by the reference in float32 (the golden) AND in float64: the distance between the two runs is how well-conditioned
    each recorded parameter element is (Adam's m / sqrt(v) amplifies gradient noise where a gradient changes sign between
    steps), and the test's bar follows it."""
    from transvae_amd import synth
    rec = {"lr": np.float64(1e-4)}
    for tag, dtype in (("p", torch.float32), ("p64", torch.float64)):
        vae = build(rm, name, ADAM_LR=1e-4)
        vae.model.to(dtype)
        _follow_dtype(vae.model, dtype)
        # the optimizer was built over the float32 parameters: rebuild it over the converted ones, as the shell does
        vae.optimizer = type(vae.optimizer)([p for p in vae.model.parameters() if p.requires_grad], 1e-4, torch.optim.Adam)
        keys = [k for k in ("generator.proj.bias", "encoder.norm.a_2", "encoder.gru.bias_hh_l0", "decoder.gru.weight_hh_l1",
                            "encoder.z_means.bias", "src_embed.lut.weight", "encoder.attn.bias", "decoder.unbottleneck.bias")
                if k in dict(vae.model.named_parameters())]
        rec["keys"] = np.array(keys)
        rng = np.random.default_rng(23)
        named = dict(vae.model.named_parameters())
        all_toks, all_eps, losses = [], [], []
        w = vae.params["CHAR_WEIGHTS"].to(dtype)
        for step in range(3):
            vae.model.zero_grad()
            for c in range(2):
                toks = synth.synth_tokens(6, seed=300 + 10 * step + c)
                eps = torch.from_numpy(rng.standard_normal((6, 128)).astype(np.float32))
                all_toks.append(toks.numpy().astype(np.int8))
                all_eps.append(eps.numpy())
                vae.model.eval()
                src, tgt = toks.long(), toks[:, :-1].long()
                orig = torch.randn_like
                torch.randn_like = lambda t, *a, **k: eps.to(t.dtype)
                try:
                    x_out, mu, logvar, pred_prop = vae.model(src, tgt, None, None)
                finally:
                    torch.randn_like = orig
                out = loss_mod.vae_loss(src, x_out, mu, logvar, torch.zeros(6, dtype=dtype), pred_prop, w, 0.01)
                losses.append(float(out[0]))
                out[0].backward()
            vae.optimizer.step()
            for k in keys:
                rec["%s_step%d/%s" % (tag, step + 1, k)] = named[k].detach().numpy()[:6].copy()      # matrices: first 6 rows
        if tag == "p":
            rec.update(tokens=np.stack(all_toks), eps=np.stack(all_eps), losses=np.array(losses), beta=np.float64(0.01))
        print("adam", name, tag, losses)
    np.savez_compressed(os.path.join(OUT, "rnn_adam_%s.npz" % name), **rec)
    for k in rec["keys"]:
        dd = np.abs(rec["p_step3/%s" % k].astype(np.float64) - rec["p64_step3/%s" % k])
        print("   %-32s reference fp32 vs fp64 after 3 steps: median %.2e max %.2e (lr 1e-4)" % (k, np.median(dd), dd.max()))


def golden_attn(rm, name="rnnattn256", n=6):
    "attention weights of RNNAttnEncoder (rnn_models.py:246, return_attn=True), as scripts/attention.py dumps them"
    from transvae_amd import synth
    vae = build(rm, name)
    vae.model.eval()
    toks = synth.synth_tokens(n, seed=77, lengths=[1, 126, 35, 60, 22, 90][:n])
    with torch.no_grad():
        _, mu, _, wts = vae.model.encoder(vae.model.src_embed(toks.long()), return_attn=True)
    np.savez_compressed(os.path.join(OUT, "rnn_attn_%s.npz" % name), tokens=toks.numpy().astype(np.int8),
                        attn=wts.numpy().astype(np.float32), mu=mu.numpy())
    print("attn", name, tuple(wts.shape), float(wts.sum(-1).mean()))


def golden_decode(rm, name, n=16):
    vae = build(rm, name)
    vae.use_gpu = False
    z = torch.from_numpy(np.random.default_rng(5).standard_normal((n, 128)).astype(np.float32))
    with torch.no_grad():
        ids = vae.greedy_decode(z)
        ids_cond = vae.greedy_decode(z[:6], condition=["C", "c"])
    np.savez_compressed(os.path.join(OUT, "rnn_decode_%s.npz" % name), z=z.numpy(), ids=ids.numpy().astype(np.int8),
                        ids_cond=ids_cond.numpy().astype(np.int8), cond=np.array(["C", "c"]))
    print("decode", name, "distinct tokens", len(np.unique(ids.numpy())), ids[0, :12].tolist())


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    import_reference()
    import transvae.rnn_models as rm
    import transvae.loss as loss_mod
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for name in CONFIGS:
        if only in ("", "step"):
            golden_step(rm, loss_mod, name)
        if only in ("", "decode"):
            golden_decode(rm, name)
    if only in ("", "attn"):
        golden_attn(rm)
    if only not in ("", "adam"):
        return
    golden_adam(rm, loss_mod, "rnnattn256")
    golden_adam(rm, loss_mod, "rnn128_pp")


if __name__ == "__main__":
    main()
