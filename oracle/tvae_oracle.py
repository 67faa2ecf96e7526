"""CPU ORACLE for the TransVAE hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, with plain torch CPU ops in functional style, the algorithm of the
reference's transformer-VAE training step and greedy decode.  It is the *checker*:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product (``transvae_amd``) never does; it fails loudly when the HIP
library is missing instead of falling back here.

Pinning: ``tests/test_oracle_golden.py`` checks every function below against golden
vectors produced by importing the real reference in the build container
(``tools/make_goldens.py`` -> ``tests/golden/*.npz``).  The reference itself ships no
tests or known-answer vectors for this path.

All ``file:line`` citations are relative to /root/reference.  The functions work on a flat
``state_dict``-style mapping ``P`` using the reference's parameter names and run in
whatever dtype ``P`` holds (float32 for parity with the reference, float64 for "truth").
"""
import math

import torch
import torch.nn.functional as F

PAD, START, END = 24, 0, 25       # data/char_dict_{zinc,pubchem}.pkl (both vocabularies)


# --------------------------------------------------------------------------- helpers
def kl_anneal(epoch, kl_low, kl_high, n_epochs, start_epoch):
    """transvae/tvae_util.py:73-92 (KLAnnealer)."""
    slope = (kl_high - kl_low) / (n_epochs - start_epoch)
    k = epoch - start_epoch if epoch >= start_epoch else 0
    return min(kl_low + k * slope, kl_high)


def noam_rate(step, model_size, factor=1.0, warmup=10000):
    """transvae/opt.py:32-36."""
    return factor * (model_size ** (-0.5) * min(step ** (-0.5), step * warmup ** (-1.5)))


def subsequent_mask(n):
    """transvae/tvae_util.py:32-37 -> (1, n, n) bool, True on and below the diagonal."""
    return torch.tril(torch.ones(1, n, n, dtype=torch.bool))


def make_masks(src, pad=PAD):
    """transvae/trans_models.py:211-215 + transvae/data.py:35-48."""
    tgt = src[:, :-1]
    src_mask = (src != pad).unsqueeze(-2)
    tgt_mask = (tgt != pad).unsqueeze(-2) & subsequent_mask(tgt.size(-1))
    return tgt, src_mask, tgt_mask


def positional_table(n, d, dtype=torch.float32):
    """transvae/trans_models.py:1052-1057 (built in float32 there, then a buffer)."""
    pe = torch.zeros(n, d)
    pos = torch.arange(0, n).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(dtype)


# --------------------------------------------------------------------------- layers
def layer_norm(x, a, b, eps=1e-6):
    """transvae/trans_models.py:1084-1087: unbiased std, eps added to std."""
    mu = x.mean(-1, keepdim=True)
    sd = x.std(-1, keepdim=True)
    return a * (x - mu) / (sd + eps) + b


def linear(P, name, x):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def mha(P, pre, q_in, k_in, v_in, mask, h, p_drop=0.0, want_p=False):
    """transvae/trans_models.py:921-941 + transvae/tvae_util.py:39-48.

    mask: (B,1,Lk) or (B,Lq,Lk), zero entries are filled with -1e9 before softmax.
    """
    B, d = q_in.size(0), q_in.size(-1)
    dk = d // h
    q = linear(P, pre + ".linears.0", q_in).view(B, -1, h, dk).transpose(1, 2)
    k = linear(P, pre + ".linears.1", k_in).view(B, -1, h, dk).transpose(1, 2)
    v = linear(P, pre + ".linears.2", v_in).view(B, -1, h, dk).transpose(1, 2)
    s = q @ k.transpose(-2, -1) / math.sqrt(dk)
    if mask is not None:
        s = s.masked_fill(mask.unsqueeze(1) == 0, -1e9)
    p = F.softmax(s, dim=-1)
    if want_p:
        return p
    if p_drop > 0:
        p = F.dropout(p, p_drop)
    o = (p @ v).transpose(1, 2).contiguous().view(B, -1, d)
    return linear(P, pre + ".linears.3", o)


_FFN_TAP = [None]        # gate_margins() listens here to the hidden pre-activations of every feed-forward block


def ffn(P, pre, x, p_drop=0.0):
    """transvae/trans_models.py:951-952."""
    hpre = linear(P, pre + ".w_1", x)
    if _FFN_TAP[0] is not None:
        _FFN_TAP[0](pre, hpre)
    hmid = F.relu(hpre)
    if p_drop > 0:
        hmid = F.dropout(hmid, p_drop)
    return linear(P, pre + ".w_2", hmid)


def _drop(x, p):
    return F.dropout(x, p) if p > 0 else x


def encoder_layer(P, pre, x, mask, h, p_drop=0.0, attn_drop=0.0):
    """transvae/trans_models.py:825-832 with the pre-norm residual of :1099-1101."""
    n0 = layer_norm(x, P[pre + ".sublayer.0.norm.a_2"], P[pre + ".sublayer.0.norm.b_2"])
    x = x + _drop(mha(P, pre + ".self_attn", n0, n0, n0, mask, h, attn_drop), p_drop)
    n1 = layer_norm(x, P[pre + ".sublayer.1.norm.a_2"], P[pre + ".sublayer.1.norm.b_2"])
    return x + _drop(ffn(P, pre + ".feed_forward", n1, p_drop), p_drop)


def decoder_layer(P, pre, x, mem, src_mask, tgt_mask, h, p_drop=0.0, attn_drop=0.0):
    """transvae/trans_models.py:893-904."""
    n0 = layer_norm(x, P[pre + ".sublayer.0.norm.a_2"], P[pre + ".sublayer.0.norm.b_2"])
    x = x + _drop(mha(P, pre + ".self_attn", n0, n0, n0, tgt_mask, h, attn_drop), p_drop)
    n1 = layer_norm(x, P[pre + ".sublayer.1.norm.a_2"], P[pre + ".sublayer.1.norm.b_2"])
    x = x + _drop(mha(P, pre + ".src_attn", n1, mem, mem, src_mask, h, attn_drop), p_drop)
    n2 = layer_norm(x, P[pre + ".sublayer.2.norm.a_2"], P[pre + ".sublayer.2.norm.b_2"])
    return x + _drop(ffn(P, pre + ".feed_forward", n2, p_drop), p_drop)


def embed(P, pre, ids, p_drop=0.0):
    """transvae/trans_models.py:1042-1043 and :1060-1063."""
    w = P[pre + ".0.lut.weight"]
    d = w.size(1)
    x = w[ids] * math.sqrt(d) + P[pre + ".1.pe"][:, :ids.size(1)].to(w.dtype)
    return _drop(x, p_drop)


def conv_bottleneck(P, x):
    """transvae/trans_models.py:980-983 (Conv1d -> MaxPool1d(2) -> ReLU, three times)."""
    for i in range(3):
        pre = "encoder.conv_bottleneck.conv_layers.%d.0" % i
        x = F.relu(F.max_pool1d(F.conv1d(x, P[pre + ".weight"], P[pre + ".bias"]), 2))
    return x


def deconv_bottleneck(P, x):
    """transvae/trans_models.py:1006-1009 (k=11, padding 2, strides 4,3,1)."""
    for i, stride in enumerate((4, 3, 1)):
        pre = "decoder.deconv_bottleneck.deconv_layers.%d.0" % i
        x = F.relu(F.conv_transpose1d(x, P[pre + ".weight"], P[pre + ".bias"],
                                      stride=stride, padding=2))
    return x


def count_layers(P, prefix):
    n = 0
    while (prefix + ".%d.sublayer.0.norm.a_2" % n) in P:
        n += 1
    return n


# --------------------------------------------------------------------------- model
def encode(P, src, src_mask, h=4, eps=None, eps_scale=1.0, p_drop=0.0, attn_drop=0.0):
    """transvae/trans_models.py:731-732 and :777-794.  ``eps`` is the N(0,1) draw of
    ``reparameterize`` (:771-775) passed explicitly; None -> fresh torch.randn."""
    x = embed(P, "src_embed", src, p_drop)
    for i in range(count_layers(P, "encoder.layers")):
        x = encoder_layer(P, "encoder.layers.%d" % i, x, src_mask, h, p_drop, attn_drop)
    mem = layer_norm(x, P["encoder.norm.a_2"], P["encoder.norm.b_2"])
    mem = conv_bottleneck(P, mem.permute(0, 2, 1))
    mem = mem.contiguous().view(mem.size(0), -1)
    mu, logvar = linear(P, "encoder.z_means", mem), linear(P, "encoder.z_var", mem)
    if eps is None:
        eps = torch.randn_like(mu)
    z = mu + eps * eps_scale * torch.exp(0.5 * logvar)
    pred_len = linear(P, "encoder.predict_len2", linear(P, "encoder.predict_len1", mu))
    return z, mu, logvar, pred_len


def expand_memory(P, z, src_mask, h=4, p_drop=0.0, attn_drop=0.0):
    """First half of VAEDecoder.forward, transvae/trans_models.py:851-860."""
    m = F.relu(linear(P, "decoder.linear", z)).view(-1, 64, 9)
    m = deconv_bottleneck(P, m).permute(0, 2, 1)
    m = encoder_layer(P, "decoder.final_encodes.0", m, src_mask, h, p_drop, attn_drop)
    return layer_norm(m, P["decoder.norm.a_2"], P["decoder.norm.b_2"])


def decode(P, z, src_mask, tgt, tgt_mask, h=4, p_drop=0.0, attn_drop=0.0):
    """transvae/trans_models.py:734-735 and :849-864."""
    mem = expand_memory(P, z, src_mask, h, p_drop, attn_drop)
    x = embed(P, "tgt_embed", tgt, p_drop)
    for i in range(count_layers(P, "decoder.layers")):
        x = decoder_layer(P, "decoder.layers.%d" % i, x, mem, src_mask, tgt_mask, h,
                          p_drop, attn_drop)
    return layer_norm(x, P["decoder.norm.a_2"], P["decoder.norm.b_2"])


def property_head(P, mu):
    """transvae/trans_models.py:1028-1031: ReLU after every layer, the last included."""
    i, x = 0, mu
    while ("property_predictor.prediction_layers.%d.weight" % i) in P:
        x = F.relu(linear(P, "property_predictor.prediction_layers.%d" % i, x))
        i += 1
    return x if i else None


def forward(P, src, tgt, src_mask, tgt_mask, h=4, eps=None, eps_scale=1.0,
            p_drop=0.0, attn_drop=0.0):
    """EncoderDecoder.forward, transvae/trans_models.py:720-729."""
    z, mu, logvar, pred_len = encode(P, src, src_mask, h, eps, eps_scale, p_drop, attn_drop)
    x = decode(P, z, src_mask, tgt, tgt_mask, h, p_drop, attn_drop)
    logits = linear(P, "generator.proj", x)
    return logits, mu, logvar, pred_len, property_head(P, mu)


def trans_vae_loss(src, logits, mu, logvar, true_len, pred_len, true_prop, pred_prop,
                   weights, beta=1.0):
    """transvae/loss.py:23-38."""
    y = (src.long()[:, 1:] - 1).contiguous().view(-1)
    bce_mol = F.cross_entropy(logits.contiguous().view(-1, logits.size(2)), y,
                              reduction="mean", weight=weights)
    bce_mask = F.cross_entropy(pred_len, true_len.contiguous().view(-1), reduction="mean")
    kld = beta * -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())
    mse = F.mse_loss(pred_prop.squeeze(-1), true_prop) if pred_prop is not None \
        else torch.zeros((), dtype=logits.dtype)
    if torch.isnan(kld):
        kld = torch.zeros((), dtype=logits.dtype)
    return bce_mol + bce_mask + kld + mse, bce_mol, bce_mask, kld, mse


def chunk_loss(P, src, props, weights, beta, h=4, eps=None, eps_scale=1.0,
               p_drop=0.0, attn_drop=0.0):
    """One chunk of VAEShell.train, transvae/trans_models.py:211-224."""
    tgt, src_mask, tgt_mask = make_masks(src)
    logits, mu, logvar, pred_len, prop = forward(P, src, tgt, src_mask, tgt_mask, h, eps,
                                                 eps_scale, p_drop, attn_drop)
    true_len = src_mask.sum(dim=-1)
    losses = trans_vae_loss(src, logits, mu, logvar, true_len, pred_len, props, prop,
                            weights, beta)
    return losses, dict(logits=logits, mu=mu, logvar=logvar, pred_len=pred_len, prop=prop)


def gate_margins(P, src, eps, h=4, eps_scale=1.0):
    """Distance of every ReLU / MaxPool gate from its switching point, per molecule: MaxPool pairs |u - v| and the ReLU
    input max(u, v) of the three conv layers (transvae/trans_models.py:980-983), the ReLU input of decoder.linear
    (:852), of the three deconv layers (:1006-1009) and of the hidden layer of all seven feed-forward blocks (:952).
    ReLU / MaxPool make the GRADIENT discontinuous there: when a margin is below an implementation's rounding error
    on that pre-activation, two correct fp32 implementations (the reference on another BLAS included) may pick
    different sides, and the gradients upstream differ by far more than rounding.  Run in float64; the parity tests
    use it to tell arithmetic error from such coin flips.
    Margins are RELATIVE to the layer's rms pre-activation over the batch (an fp32 product's error scales with the
    size of its terms: 1.2 for conv0, 0.08 for deconv2 on the synthetic checkpoint).
    Returns {gate name: (B,) tensor of the molecule's smallest relative margin}."""
    m = {}

    def tap(pre, hpre):
        m["ffn:" + pre] = hpre.abs().flatten(1).min(1)[0] / hpre.pow(2).mean().sqrt()

    _FFN_TAP[0] = tap
    try:
        with torch.no_grad():
            tgt, src_mask, tgt_mask = make_masks(src)
            x = embed(P, "src_embed", src)
            for i in range(count_layers(P, "encoder.layers")):
                x = encoder_layer(P, "encoder.layers.%d" % i, x, src_mask, h)
            y = layer_norm(x, P["encoder.norm.a_2"], P["encoder.norm.b_2"]).permute(0, 2, 1)
            for i in range(3):
                pre = "encoder.conv_bottleneck.conv_layers.%d.0" % i
                c = F.conv1d(y, P[pre + ".weight"], P[pre + ".bias"])
                L = c.shape[-1] // 2 * 2
                u, v = c[..., 0:L:2], c[..., 1:L:2]
                rms = c.pow(2).mean().sqrt()
                m["pool%d" % i] = (u - v).abs().flatten(1).min(1)[0] / rms
                m["crelu%d" % i] = torch.maximum(u, v).abs().flatten(1).min(1)[0] / rms
                y = F.relu(F.max_pool1d(c, 2))
            flat = y.contiguous().view(y.size(0), -1)
            mu, logvar = linear(P, "encoder.z_means", flat), linear(P, "encoder.z_var", flat)
            z = mu + eps.to(mu.dtype) * eps_scale * torch.exp(0.5 * logvar)
            d = linear(P, "decoder.linear", z)
            m["dlin"] = d.abs().min(1)[0] / d.pow(2).mean().sqrt()
            t = F.relu(d).view(-1, 64, 9)
            for i, stride in enumerate((4, 3, 1)):
                pre = "decoder.deconv_bottleneck.deconv_layers.%d.0" % i
                c = F.conv_transpose1d(t, P[pre + ".weight"], P[pre + ".bias"], stride=stride, padding=2)
                m["deconv%d" % i] = c.abs().flatten(1).min(1)[0] / c.pow(2).mean().sqrt()
                t = F.relu(c)
            decode(P, z, src_mask, tgt, tgt_mask, h)          # the feed-forward blocks of final_encodes and the decoder layers
    finally:
        _FFN_TAP[0] = None
    return m


def adam_step(params, grads, m, v, step, lr, beta1=0.9, beta2=0.98, eps=1e-9):
    """torch.optim.Adam as configured at transvae/trans_models.py:703-705 (no weight
    decay, no amsgrad), in-place on the given lists.  ``step`` is 1-based."""
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    for p, g, mi, vi in zip(params, grads, m, v):
        mi.lerp_(g, 1 - beta1)
        vi.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (vi.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(mi, denom, value=-lr / bc1)


# --------------------------------------------------------------------------- sampling
def predict_mask_length(P, z):
    """transvae/trans_models.py:763-769 -> (n,) int64 (argmax, first index on ties)."""
    logits = linear(P, "encoder.predict_len2", linear(P, "encoder.predict_len1", z))
    return torch.topk(F.softmax(logits, dim=-1), 1)[1].squeeze(-1)


def length_mask(lens, width=127):
    """transvae/trans_models.py:405-408 -> (n,1,width) float, ones on the first len cols."""
    return (torch.arange(width)[None, :] < lens[:, None]).float().unsqueeze(1)


@torch.no_grad()
def greedy_decode(P, z, src_mask=None, condition_ids=(), h=4, max_len=125, margins=False):
    """transvae/trans_models.py:383-436 with the reference's full recompute per step.

    condition_ids: token ids forced after <start>.  Returns (n, 125) int64 ids in 1..25
    (and, if ``margins``, the per-step top-1 minus top-2 logit gap)."""
    n = z.size(0)
    if src_mask is None:
        src_mask = length_mask(predict_mask_length(P, z))
    decoded = torch.full((n, 1), START, dtype=torch.long)
    for t in condition_ids:
        decoded = torch.cat([decoded, torch.full((n, 1), int(t), dtype=torch.long)], 1)
    tgt = torch.full((n, max_len + 1), START, dtype=torch.long)
    tgt[:, :decoded.size(1)] = decoded
    gaps = []
    for i in range(len(condition_ids), max_len):
        out = decode(P, z, src_mask, decoded, subsequent_mask(decoded.size(1)).long(), h)
        logit = linear(P, "generator.proj", out)[:, i, :]
        prob = F.softmax(logit, dim=-1)
        nxt = torch.max(prob, dim=1)[1] + 1
        if margins:
            top2 = torch.topk(logit, 2, dim=-1)[0]
            gaps.append(top2[:, 0] - top2[:, 1])
        tgt[:, i + 1] = nxt
        decoded = torch.cat([decoded, nxt.unsqueeze(1)], 1)
    ids = tgt[:, 1:]
    return (ids, torch.stack(gaps, 1)) if margins else ids


def decode_mols(ids, org_dict):
    """transvae/tvae_util.py:152-166."""
    out = []
    for row in ids.tolist():
        s = ""
        for t in row:
            tok = org_dict[t - 1]
            if tok == "<end>":
                break
            s += tok
        out.append(s.strip("_"))
    return out


@torch.no_grad()
def attention_maps(P, src, src_mask, tgt, tgt_mask, h=4, eps=None):
    """forward_w_attn of encoder and decoder, transvae/trans_models.py:796-813, :866-880,
    incl. the quirk that the dumped weights come from the UN-normalised layer input
    (:827, :898).  Returns (enc_self[3], final_encode[1], dec_src[3]) lists of P tensors."""
    x = embed(P, "src_embed", src)
    enc = []
    for i in range(count_layers(P, "encoder.layers")):
        pre = "encoder.layers.%d" % i
        enc.append(mha(P, pre + ".self_attn", x, x, x, src_mask, h, want_p=True))
        x = encoder_layer(P, pre, x, src_mask, h)
    mem = layer_norm(x, P["encoder.norm.a_2"], P["encoder.norm.b_2"])
    mem = conv_bottleneck(P, mem.permute(0, 2, 1)).contiguous().view(src.size(0), -1)
    mu, logvar = linear(P, "encoder.z_means", mem), linear(P, "encoder.z_var", mem)
    z = mu + (torch.randn_like(mu) if eps is None else eps) * torch.exp(0.5 * logvar)
    m = F.relu(linear(P, "decoder.linear", z)).view(-1, 64, 9)
    m = deconv_bottleneck(P, m).permute(0, 2, 1)
    fin = [mha(P, "decoder.final_encodes.0.self_attn", m, m, m, src_mask, h, want_p=True)]
    m = encoder_layer(P, "decoder.final_encodes.0", m, src_mask, h)
    m = layer_norm(m, P["decoder.norm.a_2"], P["decoder.norm.b_2"])
    y = embed(P, "tgt_embed", tgt)
    src_maps = []
    for i in range(count_layers(P, "decoder.layers")):
        pre = "decoder.layers.%d" % i
        n0 = layer_norm(y, P[pre + ".sublayer.0.norm.a_2"], P[pre + ".sublayer.0.norm.b_2"])
        y1 = y + mha(P, pre + ".self_attn", n0, n0, n0, tgt_mask, h)
        src_maps.append(mha(P, pre + ".src_attn", y1, m, m, src_mask, h, want_p=True))
        y = decoder_layer(P, pre, y, m, src_mask, tgt_mask, h)
    return enc, fin, src_maps


def calc_entropy(sample):
    """transvae/tvae_util.py:198-204 (numpy histogram + scipy entropy)."""
    import numpy as np
    from scipy.stats import entropy
    return np.array([entropy(np.histogram(sample[:, i], bins=1000, range=(-5., 5.),
                                          density=True)[0]) for i in range(sample.shape[1])])


# ----------------------------------------------------------------------------- data pipeline
import re as _re

# transvae/tvae_util.py:99 -- the pattern is written there as an ordinary (non-raw) string, in which the
# four backslashes of the `\\\\` alternative reach the regex engine as `\\`, i.e. one literal backslash
_SMILES_PATTERN = _re.compile(
    r"(\[[^\]]+]|Br?|Cl?|N|O|S|P|F|I|b|c|n|o|s|p|\(|\)|\.|=|#|-|\+|\\|\/|_|:|~|@|\?|>|\*|\$|\%[0-9]{2}|[0-9])")


def tokenize(smile):
    "tokenizer(), transvae/tvae_util.py:97-103: regex findall, then the join-back assertion"
    tokens = _SMILES_PATTERN.findall(smile)
    assert smile == ''.join(tokens), ("{} could not be joined".format(smile))
    return tokens


def encode_tokens(tokens, max_len, char_dict):
    "encode_smiles(), transvae/tvae_util.py:114-122: '<end>' once, then '_' while the list is short of max_len"
    tokens = list(tokens)
    for i in range(max_len - len(tokens)):
        tokens.append('<end>' if i == 0 else '_')
    return [char_dict[c] for c in tokens]


def data_rows(smiles, props, char_dict):
    """vae_data_gen(), transvae/data.py:10-33: every string tokenized first, then per row
    [0] + 126 ids into columns 0..126 and the property into column 127 of a float32 (N, 128) array."""
    import numpy as np
    toks = [tokenize(s) for s in smiles]
    if props is None:
        props = np.zeros(len(toks))
    out = np.empty((len(toks), 128), dtype=np.float32)
    for j, t in enumerate(toks):
        row = [0] + encode_tokens(t, 126, char_dict)
        if len(row) != 127:
            raise RuntimeError("row %d has %d ids, the tensor row takes 127" % (j, len(row)))
        out[j, :127] = row
        out[j, 127] = props[j]
    return out
