"""Shared helpers for the parity tests (CPU side: builds oracle parameter dicts)."""
import os
import zlib

import numpy as np
import torch

from transvae_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CONFIGS = {
    "1x128_zinc": dict(arch="trans1x-128", vocab="zinc", pp=False),
    "4x256_zinc": dict(arch="trans4x-256", vocab="zinc", pp=False),
    "4x256_pubchem_pp": dict(arch="trans4x-256", vocab="pubchem", pp=True),
}


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def state_shapes(d_model, d_ff, pp=False, N=3, d_latent=128, vocab=26, d_pp=256, depth_pp=2):
    """Key -> shape table of the reference's state_dict, built from first
    principles so the oracle does not need the product's modules."""
    S = {}

    def lin(k, o, i):
        S[k + ".weight"] = (o, i)
        S[k + ".bias"] = (o,)

    def attn(k):
        for j in range(4):
            lin("%s.linears.%d" % (k, j), d_model, d_model)

    def enc_layer(k):
        attn(k + ".self_attn")
        lin(k + ".feed_forward.w_1", d_ff, d_model)
        lin(k + ".feed_forward.w_2", d_model, d_ff)
        for j in range(2):
            S["%s.sublayer.%d.norm.a_2" % (k, j)] = (d_model,)
            S["%s.sublayer.%d.norm.b_2" % (k, j)] = (d_model,)

    for i in range(N):
        enc_layer("encoder.layers.%d" % i)
    c = d_model
    for i, ksz in enumerate((9, 8, 8)):
        o = 64 if i == 2 else (c - 64) // 2 + 64
        S["encoder.conv_bottleneck.conv_layers.%d.0.weight" % i] = (o, c, ksz)
        S["encoder.conv_bottleneck.conv_layers.%d.0.bias" % i] = (o,)
        c = o
    lin("encoder.z_means", d_latent, 576)
    lin("encoder.z_var", d_latent, 576)
    S["encoder.norm.a_2"] = (d_model,)
    S["encoder.norm.b_2"] = (d_model,)
    lin("encoder.predict_len1", 2 * d_latent, d_latent)
    lin("encoder.predict_len2", d_latent, 2 * d_latent)
    enc_layer("decoder.final_encodes.0")
    for i in range(N):
        k = "decoder.layers.%d" % i
        attn(k + ".self_attn")
        attn(k + ".src_attn")
        lin(k + ".feed_forward.w_1", d_ff, d_model)
        lin(k + ".feed_forward.w_2", d_model, d_ff)
        for j in range(3):
            S["%s.sublayer.%d.norm.a_2" % (k, j)] = (d_model,)
            S["%s.sublayer.%d.norm.b_2" % (k, j)] = (d_model,)
    S["decoder.norm.a_2"] = (d_model,)
    S["decoder.norm.b_2"] = (d_model,)
    lin("decoder.linear", 576, d_latent)
    c = 64
    for i in range(3):
        o = d_model if i == 2 else (d_model - c) // 4 + c
        S["decoder.deconv_bottleneck.deconv_layers.%d.0.weight" % i] = (c, o, 11)
        S["decoder.deconv_bottleneck.deconv_layers.%d.0.bias" % i] = (o,)
        c = o
    S["src_embed.0.lut.weight"] = (vocab, d_model)
    S["tgt_embed.0.lut.weight"] = (vocab, d_model)
    lin("generator.proj", vocab - 1, d_model)
    if pp:
        for i in range(depth_pp):
            lin("property_predictor.prediction_layers.%d" % i,
                1 if i == depth_pp - 1 else d_pp, d_latent if i == 0 else d_pp)
    return S


def oracle_params(cfg_name, dtype=torch.float32, seed=0, requires_grad=False):
    """Synthetic checkpoint as an oracle parameter dict (same bits as the golden run)."""
    from oracle import tvae_oracle as O
    cfg = CONFIGS[cfg_name]
    arch = synth.ARCHS[cfg["arch"]]
    shapes = state_shapes(arch["d_model"], arch["d_ff"], cfg["pp"])
    P = {k: torch.from_numpy(synth.synth_tensor(k, s, seed)).to(dtype) for k, s in shapes.items()}
    if requires_grad:
        for v in P.values():
            v.requires_grad_(True)
    pe = O.positional_table(5000, arch["d_model"]).unsqueeze(0)
    P["src_embed.1.pe"] = pe.to(dtype)
    P["tgt_embed.1.pe"] = pe.to(dtype)
    return P


def probe_vec(key, n):
    rng = np.random.default_rng(zlib.crc32(("probe:" + key).encode()))
    return rng.standard_normal(n)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    nb = np.linalg.norm(b)
    if nb < 1e-12:          # analytically-zero reference: fall back to an absolute bound
        return float(np.linalg.norm(a - b)) / 10.0
    return float(np.linalg.norm(a - b) / nb)


PARITY_DIR = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out", "parity")


def record(name, lines):
    """Append measured parity numbers to gpurun_out/parity/<name>.txt (the directory gpurun brings back from the GPU
    box; the round's tables are copied from there into profiles/).  pytest -q drops what tests print: this keeps it."""
    try:
        os.makedirs(PARITY_DIR, exist_ok=True)
        with open(os.path.join(PARITY_DIR, name + ".txt"), "a") as f:
            for ln in ([lines] if isinstance(lines, str) else lines):
                f.write(ln.rstrip("\n") + "\n")
    except OSError:
        pass


# RELATIVE margins (oracle.gate_margins: distance from the switching point / the layer's rms pre-activation, float64)
# below which a bottleneck gate of a molecule counts as "unsafe".  An fp32 implementation's error on such a pre-activation
# is 3e-7 .. 1e-6 of that rms (K = 700 .. 2300 products), so 4e-6 leaves a factor of four; a deconv2 unit is one of
# 32,512 output units per molecule and moves gradients 4x less (the weight tools/make_goldens.py gives it)
GATE_THRESHOLDS = dict(pool0=4e-6, crelu0=4e-6, pool1=4e-6, crelu1=4e-6, pool2=4e-6, crelu2=4e-6,
                       dlin=4e-6, deconv0=4e-6, deconv1=4e-6, deconv2=1e-6)
# hidden ReLU of the seven feed-forward blocks (910,000 gates per trans4x-256 molecule): measured on the MI355X
# (tools/debug/ffn_big.py) the fp32 GEMM decides 5 of 65 M such gates differently from float64, all with |pre| < 2e-7;
# ONE such flip moves the gradients on the input side of that block by 1e-4
FFN_GATE_THRESHOLD = 1e-6


def gate_safe_rows(P64, toks, eps, thresholds=None, piece=96):
    """(bool (n,) of molecules whose every bottleneck gate clears its threshold, {gate: unsafe molecules})"""
    from oracle import tvae_oracle as O
    th = thresholds or GATE_THRESHOLDS
    parts = []
    for lo in range(0, toks.shape[0], piece):
        parts.append(O.gate_margins(P64, toks[lo:lo + piece], eps[lo:lo + piece].double()))
    m = {k: torch.cat([p[k] for p in parts]) for k in parts[0]}
    safe = torch.ones(toks.shape[0], dtype=torch.bool)
    unsafe = {}
    for k, t in th.items():
        bad = m[k] < t
        unsafe[k] = int(bad.sum())
        safe &= ~bad
    ffn_bad = torch.zeros_like(safe)
    for k in m:
        if k.startswith("ffn:"):
            ffn_bad |= m[k] < FFN_GATE_THRESHOLD
    unsafe["ffn (7 blocks)"] = int(ffn_bad.sum())
    safe &= ~ffn_bad
    return safe, unsafe


# ---- RNN / RNNAttn models ---------------------------------------------------------------------------------------------
RNN_CONFIGS = {      # tools/make_goldens_rnn.py CONFIGS
    "rnnattn256": dict(cls="RNNAttn", vocab="zinc", kw=dict(N=3, d_model=256, d_latent=128)),
    "rnn128_pp": dict(cls="RNN", vocab="pubchem", kw=dict(N=3, d_model=128, d_latent=128, property_predictor=True)),
    "rnnattn128_notf": dict(cls="RNNAttn", vocab="zinc", kw=dict(N=2, d_model=128, d_latent=128, tf=False)),
}


def rnn_state_shapes(cls, d_model, N=3, d_latent=128, tf=True, property_predictor=False, vocab=26, d_pp=256, depth_pp=2):
    """Key -> shape table of the reference's RNN / RNNAttn state_dict (transvae/rnn_models.py:81-95, 149-162, 225-231,
    281-285, 320-323, 362-365), from first principles so that the oracle does not need the product's modules."""
    S = {}

    def lin(k, o, i):
        S[k + ".weight"] = (o, i)
        S[k + ".bias"] = (o,)

    def gru(k, inp):
        for l in range(N):
            S["%s.weight_ih_l%d" % (k, l)] = (3 * d_model, inp if l == 0 else d_model)
            S["%s.weight_hh_l%d" % (k, l)] = (3 * d_model, d_model)
            S["%s.bias_ih_l%d" % (k, l)] = (3 * d_model,)
            S["%s.bias_hh_l%d" % (k, l)] = (3 * d_model,)

    gru("encoder.gru", d_model)
    if cls == "RNNAttn":
        lin("encoder.attn", 127, 2 * d_model)
        c = d_model
        for i, ksz in enumerate((9, 8, 8)):
            o = 64 if i == 2 else (c - 64) // 2 + 64
            S["encoder.conv_bottleneck.conv_layers.%d.0.weight" % i] = (o, c, ksz)
            S["encoder.conv_bottleneck.conv_layers.%d.0.bias" % i] = (o,)
            c = o
        lin("encoder.z_means", d_latent, 576)
        lin("encoder.z_var", d_latent, 576)
    else:
        lin("encoder.z_means", d_latent, d_model)
        lin("encoder.z_var", d_latent, d_model)
    S["encoder.norm.a_2"] = (d_model,)
    S["encoder.norm.b_2"] = (d_model,)
    if cls == "RNNAttn":
        lin("decoder.linear", 576, d_latent)
        c = 64
        for i in range(3):
            o = d_model if i == 2 else (d_model - c) // 4 + c
            S["decoder.deconv_bottleneck.deconv_layers.%d.0.weight" % i] = (c, o, 11)
            S["decoder.deconv_bottleneck.deconv_layers.%d.0.bias" % i] = (o,)
            c = o
    gru("decoder.gru", 2 * d_model if tf else d_model)
    if cls == "RNN":
        lin("decoder.unbottleneck", d_model, d_latent)
    S["decoder.norm.a_2"] = (d_model,)
    S["decoder.norm.b_2"] = (d_model,)
    S["src_embed.lut.weight"] = (vocab, d_model)
    S["tgt_embed.lut.weight"] = (vocab, d_model)
    lin("generator.proj", vocab - 1, d_model)
    if property_predictor:
        for i in range(depth_pp):
            lin("property_predictor.prediction_layers.%d" % i, 1 if i == depth_pp - 1 else d_pp, d_latent if i == 0 else d_pp)
    return S


def rnn_oracle_params(cfg_name, dtype=torch.float32, seed=0, requires_grad=False):
    cfg = RNN_CONFIGS[cfg_name]
    shapes = rnn_state_shapes(cfg["cls"], **cfg["kw"])
    P = {k: torch.from_numpy(synth.synth_tensor(k, s, seed)).to(dtype) for k, s in shapes.items()}
    if requires_grad:
        for v in P.values():
            v.requires_grad_(True)
    return P
