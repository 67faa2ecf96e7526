"""CPU ORACLE for the RNN / RNNAttn models -- TEST INFRASTRUCTURE ONLY (see oracle/tvae_oracle.py for the rules: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product never does).

Restates with plain torch CPU ops the forward pass, loss and greedy decode of the reference's recurrent VAEs
(/root/reference/transvae/rnn_models.py:178-384, transvae/loss.py:8-21).  The GRU is written out cell by cell (torch.nn.GRU
semantics: gates r, z, n; n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1 - z) n + z h), so nothing here calls
torch.nn.GRU.  Functions work on a flat state_dict-style mapping ``P`` with the reference's parameter names and run in
whatever dtype ``P`` holds.

Pinning: tests/test_oracle_golden.py checks these functions against tests/golden/rnn_*.npz, produced by importing the real
reference in the build container (tools/make_goldens_rnn.py).
"""
import math

import torch
import torch.nn.functional as F

from .tvae_oracle import conv_bottleneck, deconv_bottleneck, layer_norm, linear, property_head


def gru_layers(P, pre):
    n = 0
    while (pre + ".weight_ih_l%d" % n) in P:
        n += 1
    return n


def gru_cell(P, pre, l, x, h):
    "one step of layer l: x (B, I), h (B, H) -> h' (B, H)   (torch.nn.GRU, as used at rnn_models.py:241/301/333/378)"
    gi = F.linear(x, P[pre + ".weight_ih_l%d" % l], P[pre + ".bias_ih_l%d" % l])
    gh = F.linear(h, P[pre + ".weight_hh_l%d" % l], P[pre + ".bias_hh_l%d" % l])
    H = h.shape[1]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def gru(P, pre, x, p_drop=0.0):
    "multi-layer GRU over x (B, T, I) from zero states -> (outputs of the top layer (B, T, H), final states [(B, H)] per layer)"
    B, T, _ = x.shape
    H = P[pre + ".weight_hh_l0"].shape[1]
    finals = []
    for l in range(gru_layers(P, pre)):
        h = torch.zeros(B, H, dtype=x.dtype)
        outs = []
        for t in range(T):
            h = gru_cell(P, pre, l, x[:, t], h)
            outs.append(h)
        x = torch.stack(outs, 1)
        finals.append(h)
        if p_drop > 0 and l + 1 < gru_layers(P, pre):
            x = F.dropout(x, p_drop)
    return x, finals


def embed(P, ids):
    "Embeddings: lut[id] * sqrt(d_model) (trans_models.py:1042-1043); both encode and decode use src_embed (rnn_models.py:204, 207)"
    w = P["src_embed.lut.weight"]
    return w[ids] * math.sqrt(w.shape[1])


def is_attn(P):
    return "encoder.attn.weight" in P


def encode(P, src, eps=None):
    "RNNAttnEncoder.forward (rnn_models.py:238-260) / RNNEncoder.forward (:330-340)"
    x = embed(P, src)
    x_out, finals = gru(P, "encoder.gru", x)
    if is_attn(P):
        mem = layer_norm(x_out, P["encoder.norm.a_2"], P["encoder.norm.b_2"])
        w = F.softmax(linear(P, "encoder.attn", torch.cat((x, mem), 2)), dim=2)
        mem = F.relu(torch.bmm(w, mem))
        mem = conv_bottleneck(P, mem.permute(0, 2, 1))
        mem = mem.contiguous().view(mem.size(0), -1)
    else:
        w = None
        mem = layer_norm(finals[-1], P["encoder.norm.a_2"], P["encoder.norm.b_2"])
    mu, logvar = linear(P, "encoder.z_means", mem), linear(P, "encoder.z_var", mem)
    if eps is None:
        eps = torch.randn_like(mu)
    return mu + eps * torch.exp(0.5 * logvar), mu, logvar, w


def expand_memory(P, z, length=126):
    "the memory rows the decoder GRU reads beside the embedded tokens (rnn_models.py:290-296 / :370-373)"
    if is_attn(P):
        m = F.relu(linear(P, "decoder.linear", z)).view(-1, 64, 9)
        m = deconv_bottleneck(P, m).permute(0, 2, 1)
        m = layer_norm(m, P["decoder.norm.a_2"], P["decoder.norm.b_2"])
        return m[:, :-1, :]
    m = F.relu(linear(P, "decoder.unbottleneck", z)).unsqueeze(1).repeat(1, length, 1)
    return layer_norm(m, P["decoder.norm.a_2"], P["decoder.norm.b_2"])


def decode(P, tgt, z, teacher_force=True):
    "RNNAttnDecoder.forward (rnn_models.py:287-304) / RNNDecoder.forward (:367-381), eval mode"
    mem = expand_memory(P, z, tgt.shape[1])
    if teacher_force:
        mem = torch.cat((embed(P, tgt), mem), dim=2)
    x, _ = gru(P, "decoder.gru", mem)
    return layer_norm(x, P["decoder.norm.a_2"], P["decoder.norm.b_2"])


def forward(P, src, eps=None, teacher_force=True):
    "RNNEncoderDecoder.forward (rnn_models.py:193-201) on src, tgt = src[:, :-1]"
    z, mu, logvar, _ = encode(P, src, eps)
    logits = linear(P, "generator.proj", decode(P, src[:, :-1], z, teacher_force))
    return logits, mu, logvar, property_head(P, mu)


def vae_loss(src, logits, mu, logvar, true_prop, pred_prop, weights, beta=1.0):
    "transvae/loss.py:8-21"
    y = (src.long()[:, 1:] - 1).contiguous().view(-1)
    bce = F.cross_entropy(logits.contiguous().view(-1, logits.size(2)), y, reduction="mean", weight=weights)
    kld = beta * -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())
    mse = F.mse_loss(pred_prop.squeeze(-1), true_prop) if pred_prop is not None else torch.zeros((), dtype=logits.dtype)
    if torch.isnan(kld):
        kld = torch.zeros((), dtype=logits.dtype)
    return bce + kld + mse, bce, kld, mse


def greedy_decode(P, z, condition_ids=(), max_len=125, start=0, teacher_force=True, margins=False):
    """VAEShell.greedy_decode for model_type != 'transformer' (trans_models.py:394-436).  The reference re-runs the decoder
    over the whole 126-token buffer per new token and reads position i; the GRU only looks backwards, so carrying the
    hidden states gives the same numbers step by step -- which is what this loop does (the golden tokens come from the
    reference's own loop)."""
    n = z.shape[0]
    rows = expand_memory(P, z, max_len + 1)
    tgt = torch.full((n, max_len + 1), start, dtype=torch.long)
    for j, c in enumerate(condition_ids):
        tgt[:, j + 1] = c
    L = gru_layers(P, "decoder.gru")
    H = P["decoder.gru.weight_hh_l0"].shape[1]
    h = [torch.zeros(n, H, dtype=z.dtype) for _ in range(L)]
    gaps = torch.zeros(n, max_len)
    for i in range(max_len):
        x = rows[:, i]
        if teacher_force:
            x = torch.cat((embed(P, tgt[:, i]), x), dim=1)
        for l in range(L):
            h[l] = gru_cell(P, "decoder.gru", l, x, h[l])
            x = h[l]
        if i >= len(condition_ids):
            out = linear(P, "generator.proj", layer_norm(x, P["decoder.norm.a_2"], P["decoder.norm.b_2"]))
            t2 = torch.topk(out, 2, dim=-1)[0]
            gaps[:, i] = (t2[:, 0] - t2[:, 1]).float()
            tgt[:, i + 1] = torch.max(out, dim=1)[1] + 1
    return (tgt[:, 1:], gaps) if margins else tgt[:, 1:]
