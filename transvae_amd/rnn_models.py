"""RNN and RNNAttn model classes with the reference's API and state_dict keys (reference transvae/rnn_models.py:19-384),
executing on the HIP kernels of libtvae_hip.so.

Same class names, constructor signatures and parameter names as the reference, so checkpoints are interchangeable: the
recurrent layers keep torch.nn.GRU's parameter names (``gru.weight_ih_l0`` ...).  What differs is the execution: a GRU
stack is one autograd Function (ops.GRUStackFn: one GEMM per layer for the input projections of all time steps, one fused
launch per time step for the recurrence), the attention of RNNAttnEncoder is a row-softmax kernel plus one batched GEMM,
the bottlenecks, LayerNorm, embeddings, loss and Adam are the kernels the transformer uses, and greedy decoding carries
the hidden states from step to step instead of re-running the decoder over the whole prefix for every token.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .opt import Adam, AdamOpt
from .trans_models import (ConvBottleneck, DeconvBottleneck, Embeddings, Generator, LayerNorm, PropertyPredictor, VAEShell,
                           _Sited)


class GRU(_Sited):
    """torch.nn.GRU(input_size, hidden_size, num_layers, dropout) for batch-major (B, T, I) inputs from a zero initial state.
    Parameters carry nn.GRU's names, shapes ((3H, I) / (3H, H) / (3H,), gates r, z, n) and default initialisation."""

    def __init__(self, input_size, hidden_size, num_layers=1, dropout=0.0):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers, self.dropout = input_size, hidden_size, num_layers, float(dropout)
        if hidden_size % 64:
            raise ops._lib.TvaeError("GRU: hidden size %d must be a multiple of 64 (tvae_gru_fwd)" % hidden_size)
        bound = 1.0 / math.sqrt(hidden_size)
        for l in range(num_layers):
            shapes = ((3 * hidden_size, input_size if l == 0 else hidden_size), (3 * hidden_size, hidden_size),
                      (3 * hidden_size,), (3 * hidden_size,))
            for name, shape in zip(("weight_ih", "weight_hh", "bias_ih", "bias_hh"), shapes):
                setattr(self, "%s_l%d" % (name, l), nn.Parameter(torch.empty(shape).uniform_(-bound, bound)))
        self._sites = [ops.next_site() for _ in range(max(4, num_layers))]

    def _new_sites(self):
        self._sites = [ops.next_site() for _ in range(max(4, self.num_layers))]

    def flat_params(self):
        return [getattr(self, "%s_l%d" % (name, l)) for l in range(self.num_layers)
                for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]

    def forward(self, x):
        "(B, T, I) -> (top-layer outputs (B, T, H), top-layer final state (B, H))"
        drops = [self._drop(self.dropout, l, x.device) for l in range(self.num_layers - 1)]
        return ops.GRUStackFn.apply(x, drops, *self.flat_params())

    def step(self, x, h):
        """One time step for greedy decoding: x (n, I), h list of (n, H) per layer -> new h list (eval mode, no dropout)."""
        n = x.shape[0]
        H = self.hidden_size
        out = []
        with torch.no_grad():
            for l in range(self.num_layers):
                w_ih, w_hh, b_ih, b_hh = self.flat_params()[4 * l:4 * l + 4]
                gi = ops.linear_fwd(x, w_ih, b_ih)
                hn = torch.empty((n, H), device=x.device, dtype=torch.float32)
                ops.call("tvae_gru_fwd", ops._p(gi), ops._p(w_hh), ops._p(b_hh), ops._p(h[l]), ops._p(hn), None, 1, n, H, 1,
                         ops._stream())
                out.append(hn)
                x = hn
        return out


class RNNEncoderDecoder(nn.Module):
    "encode -> decode -> generator (+ property head)  (reference rnn_models.py:178-210)"

    def __init__(self, encoder, decoder, src_embed, tgt_embed, generator, property_predictor, params):
        super().__init__()
        self.params = params
        self.encoder = encoder
        self.decoder = decoder
        self.src_embed = src_embed
        self.tgt_embed = tgt_embed
        self.generator = generator
        self.property_predictor = property_predictor

    def forward(self, src, tgt, src_mask=None, tgt_mask=None, eps=None):
        mem, mu, logvar = self.encode(src, eps)
        x, h = self.decode(tgt, mem)
        x = self.generator(x)
        prop = self.predict_property(mu) if self.property_predictor is not None else None
        return x, mu, logvar, prop

    def encode(self, src, eps=None):
        if self.training:
            ops.next_pass()              # fresh dropout masks for every training-mode forward
        return self.encoder(self.src_embed(src), eps=eps)

    def decode(self, tgt, mem):
        # reference quirk kept (rnn_models.py:207): the target tokens go through src_embed; tgt_embed is never used
        return self.decoder(self.src_embed(tgt), mem)

    def predict_property(self, mu):
        return self.property_predictor(mu)


def _reparameterize(mu, logvar, eps=None):
    if eps is None:
        eps = torch.randn_like(mu)
    return ops.ReparamFn.apply(mu, logvar, eps, 1)


class RNNAttnEncoder(nn.Module):
    "GRU -> norm -> attention over the sequence -> conv bottleneck -> (mu, logvar) -> z  (reference rnn_models.py:212-263)"

    def __init__(self, size, d_latent, N, dropout, src_length, bypass_attention, bypass_bottleneck, device=None):
        super().__init__()
        if bypass_bottleneck:
            raise NotImplementedError("bypass_bottleneck=True cannot train in the reference either: its stand-in "
                                      "logvar = 100 makes the KL term infinite (rnn_models.py:249-250, loss.py:14)")
        self.size, self.n_layers, self.max_length = size, N, src_length + 1
        self.bypass_attention, self.bypass_bottleneck, self.device = bypass_attention, bypass_bottleneck, device
        self.gru = GRU(size, size, num_layers=N, dropout=dropout)
        self.attn = nn.Linear(size * 2, self.max_length)
        self.conv_bottleneck = ConvBottleneck(size)
        self.z_means = nn.Linear(576, d_latent)
        self.z_var = nn.Linear(576, d_latent)
        self.dropout = nn.Dropout(p=dropout)
        self.norm = LayerNorm(size)

    def reparameterize(self, mu, logvar, eps=None):
        return _reparameterize(mu, logvar, eps)

    def forward(self, x, return_attn=False, eps=None):
        x_out, _ = self.gru(x)
        mem = self.norm(x_out)
        attn_weights = None
        if not self.bypass_attention:
            logits = ops.LinearFn.apply(torch.cat((x, mem), 2), self.attn.weight, self.attn.bias, False)
            attn_weights = ops.SoftmaxFn.apply(logits)
            mem = ops.BmmReluFn.apply(attn_weights, mem)
        flat = self.conv_bottleneck.forward_token_major(mem)
        mu = ops.LinearFn.apply(flat, self.z_means.weight, self.z_means.bias, False)
        logvar = ops.LinearFn.apply(flat, self.z_var.weight, self.z_var.bias, False)
        mem = self.reparameterize(mu, logvar, eps)
        if return_attn:
            return mem, mu, logvar, attn_weights.detach().cpu()
        return mem, mu, logvar

    def initH(self, batch_size):
        return torch.zeros(self.n_layers, batch_size, self.size, device=self.gru.weight_hh_l0.device)


class _DecoderBase(_Sited):
    def _embedded(self, tgt):
        return ops.ResidualDropoutFn.apply(tgt, None, self._drop(self.dropout.p, 0, tgt.device))

    def _run(self, embedded, mem):
        if self.teacher_force:
            mem = torch.cat((embedded, mem), dim=2)
        x, h = self.gru(mem.contiguous())
        return self.norm(x), h

    def initH(self, batch_size):
        return torch.zeros(self.n_layers, batch_size, self.size, device=self.gru.weight_hh_l0.device)


class RNNAttnDecoder(_DecoderBase):
    "linear + deconv memory expansion -> norm -> [embedded ; memory] -> GRU -> norm  (reference rnn_models.py:265-307)"

    def __init__(self, size, d_latent, N, dropout, tf, bypass_bottleneck, device=None):
        super().__init__()
        if bypass_bottleneck:
            raise NotImplementedError("bypass_bottleneck=True: see RNNAttnEncoder")
        self.size, self.n_layers, self.teacher_force = size, N, tf
        self.gru_size = size * 2 if tf else size
        self.bypass_bottleneck, self.device = bypass_bottleneck, device
        self.linear = nn.Linear(d_latent, 576)
        self.deconv_bottleneck = DeconvBottleneck(size)
        self.dropout = nn.Dropout(p=dropout)
        self.gru = GRU(self.gru_size, size, num_layers=N, dropout=dropout)
        self.norm = LayerNorm(size)

    def expand_memory(self, mem):
        "z (B, d_latent) -> normalised memory rows (B, 126, size) the GRU reads beside the embedded tokens"
        h = ops.LinearFn.apply(mem, self.linear.weight, self.linear.bias, True)
        m = self.norm(self.deconv_bottleneck.forward_token_major(h))
        return m[:, :-1, :]

    def forward(self, tgt, mem):
        return self._run(self._embedded(tgt), self.expand_memory(mem))


class RNNEncoder(nn.Module):
    "GRU -> norm of the final state -> (mu, logvar) -> z  (reference rnn_models.py:309-343)"

    def __init__(self, size, d_latent, N, dropout, bypass_bottleneck, device=None):
        super().__init__()
        if bypass_bottleneck:
            raise NotImplementedError("bypass_bottleneck=True is not supported (RNNDecoder of the reference then feeds a "
                                      "(B, size) memory to torch.cat with a (B, 126, size) tensor, rnn_models.py:370-375)")
        self.size, self.n_layers, self.bypass_bottleneck, self.device = size, N, bypass_bottleneck, device
        self.gru = GRU(size, size, num_layers=N, dropout=dropout)
        self.z_means = nn.Linear(size, d_latent)
        self.z_var = nn.Linear(size, d_latent)
        self.norm = LayerNorm(size)

    def reparameterize(self, mu, logvar, eps=None):
        return _reparameterize(mu, logvar, eps)

    def forward(self, x, eps=None):
        _, h_last = self.gru(x)
        mem = self.norm(h_last)
        mu = ops.LinearFn.apply(mem, self.z_means.weight, self.z_means.bias, False)
        logvar = ops.LinearFn.apply(mem, self.z_var.weight, self.z_var.bias, False)
        return self.reparameterize(mu, logvar, eps), mu, logvar

    def initH(self, batch_size):
        return torch.zeros(self.n_layers, batch_size, self.size, device=self.gru.weight_hh_l0.device)


class RNNDecoder(_DecoderBase):
    "relu(unbottleneck(z)) repeated over the sequence -> norm -> [embedded ; memory] -> GRU -> norm  (reference :345-384)"

    def __init__(self, size, d_latent, N, dropout, tgt_length, tf, bypass_bottleneck, device=None):
        super().__init__()
        if bypass_bottleneck:
            raise NotImplementedError("bypass_bottleneck=True: see RNNEncoder")
        self.size, self.n_layers, self.max_length, self.teacher_force = size, N, tgt_length + 1, tf
        self.gru_size = size * 2 if tf else size
        self.bypass_bottleneck, self.device = bypass_bottleneck, device
        self.gru = GRU(self.gru_size, size, num_layers=N, dropout=dropout)
        self.unbottleneck = nn.Linear(d_latent, size)
        self.dropout = nn.Dropout(dropout)
        self.norm = LayerNorm(size)

    def expand_memory(self, mem):
        "z (B, d_latent) -> (B, 126, size): every position reads the same normalised row (norm of a repeat = repeat of the norm)"
        m = self.norm(ops.LinearFn.apply(mem, self.unbottleneck.weight, self.unbottleneck.bias, True))
        return m.unsqueeze(1).expand(m.shape[0], self.max_length, m.shape[1])

    def forward(self, tgt, mem):
        return self._run(self._embedded(tgt), self.expand_memory(mem))


def _finish_model(model):
    for p in model.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)
    for m in model.modules():
        if hasattr(m, "_new_sites"):
            m._new_sites()
    return model


class _RNNShell(VAEShell):
    def _common(self, arch):
        if 'ADAM_LR' not in self.params.keys():
            self.params['ADAM_LR'] = 3e-4
        self.params['model_type'] = self.model_type
        self.params.update(arch)
        self.arch_params = list(arch.keys())

    def _finish_build(self):
        self.use_gpu = torch.cuda.is_available()
        if self.use_gpu:
            self.model.cuda()
            self.params['CHAR_WEIGHTS'] = self.params['CHAR_WEIGHTS'].cuda()
        self.optimizer = AdamOpt([p for p in self.model.parameters() if p.requires_grad], self.params['ADAM_LR'], Adam)
        self.optimizer.optimizer.fuse_zero_grad = True


class RNNAttn(_RNNShell):
    "RNN-based VAE with attention (reference transvae/rnn_models.py:19-106)"

    def __init__(self, params={}, name=None, N=3, d_model=128, d_latent=128, dropout=0.1, tf=True, bypass_attention=False,
                 bypass_bottleneck=False, property_predictor=False, d_pp=256, depth_pp=2, load_fn=None):
        super().__init__(params, name)
        self.model_type = 'rnn_attn'
        self._common(dict(N=N, d_model=d_model, d_latent=d_latent, dropout=dropout, teacher_force=tf,
                          bypass_attention=bypass_attention, bypass_bottleneck=bypass_bottleneck,
                          property_predictor=property_predictor, d_pp=d_pp, depth_pp=depth_pp))
        if load_fn is None:
            self.build_model()
        else:
            self.load(load_fn)

    def build_model(self):
        p = self.params
        encoder = RNNAttnEncoder(p['d_model'], p['d_latent'], p['N'], p['dropout'], self.src_len, p['bypass_attention'],
                                 p['bypass_bottleneck'])
        decoder = RNNAttnDecoder(p['d_model'], p['d_latent'], p['N'], p['dropout'], p['teacher_force'], p['bypass_bottleneck'])
        pp = PropertyPredictor(p['d_pp'], p['depth_pp'], p['d_latent']) if p['property_predictor'] else None
        self.model = _finish_model(RNNEncoderDecoder(encoder, decoder, Embeddings(p['d_model'], self.vocab_size),
                                                     Embeddings(p['d_model'], self.vocab_size),
                                                     Generator(p['d_model'], self.vocab_size), pp, self.params))
        self._finish_build()


class RNN(_RNNShell):
    "RNN-based VAE without attention (reference transvae/rnn_models.py:108-173)"

    def __init__(self, params={}, name=None, N=3, d_model=128, d_latent=128, dropout=0.1, tf=True, bypass_bottleneck=False,
                 property_predictor=False, d_pp=256, depth_pp=2, load_fn=None):
        super().__init__(params, name)
        self.model_type = 'rnn'
        self._common(dict(N=N, d_model=d_model, d_latent=d_latent, dropout=dropout, teacher_force=tf,
                          bypass_bottleneck=bypass_bottleneck, property_predictor=property_predictor, d_pp=d_pp,
                          depth_pp=depth_pp))
        if load_fn is None:
            self.build_model()
        else:
            self.load(load_fn)

    def build_model(self):
        p = self.params
        encoder = RNNEncoder(p['d_model'], p['d_latent'], p['N'], p['dropout'], p['bypass_bottleneck'])
        decoder = RNNDecoder(p['d_model'], p['d_latent'], p['N'], p['dropout'], 125, p['teacher_force'], p['bypass_bottleneck'])
        pp = PropertyPredictor(p['d_pp'], p['depth_pp'], p['d_latent']) if p['property_predictor'] else None
        self.model = _finish_model(RNNEncoderDecoder(encoder, decoder, Embeddings(p['d_model'], self.vocab_size),
                                                     Embeddings(p['d_model'], self.vocab_size),
                                                     Generator(p['d_model'], self.vocab_size), pp, self.params))
        self._finish_build()
