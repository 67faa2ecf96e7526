"""TransVAE model classes with the reference's API and state_dict keys, executing on the HIP
kernels of libtvae_hip.so (reference: transvae/trans_models.py:616-1101).

Same class names, constructor signatures, forward signatures and parameter names as the
reference, so checkpoints are interchangeable.  What differs is the execution: every
EncoderLayer / DecoderLayer runs as two or three fused autograd Functions (ops.AttnSublayerFn,
ops.FFNSublayerFn) and the bottlenecks as one Function each; dropout masks are counter-based
and regenerated in the backward kernels.  Modules used on their own (MultiHeadedAttention,
SublayerConnection, LayerNorm, ...) still work and still run on the kernels.
"""
import copy
import math

import torch
import torch.nn as nn

from . import ops
from .tvae_util import ListModule, clones


def _site_holder(module):
    """Give `module` fresh dropout site ids (also called on every deep copy by clones())."""
    module._sites = [ops.next_site() for _ in range(4)]


class _Sited(nn.Module):
    def __init__(self):
        super().__init__()
        _site_holder(self)

    def _new_sites(self):
        _site_holder(self)

    def _drop(self, p, which, device):
        return ops.Drop.make(p, self._sites[which], self.training, device)


# ------------------------------------------------------------------------------ leaf layers
class LayerNorm(nn.Module):
    "y = a_2 * (x - mean) / (std_unbiased + eps) + b_2   (reference :1076-1087)"

    def __init__(self, features, eps=1e-6):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(features))
        self.b_2 = nn.Parameter(torch.zeros(features))
        self.eps = eps

    def forward(self, x):
        return ops.LayerNormFn.apply(x, self.a_2, self.b_2, self.eps)


class SublayerConnection(_Sited):
    "x + dropout(sublayer(norm(x))): pre-norm residual (reference :1089-1101)"

    def __init__(self, size, dropout):
        super().__init__()
        self.norm = LayerNorm(size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, sublayer):
        y = sublayer(self.norm(x))
        return ops.ResidualDropoutFn.apply(y, x, self._drop(self.dropout.p, 0, x.device))


class MultiHeadedAttention(_Sited):
    "Four d_model x d_model projections around scaled-dot-product attention (reference :908-941)"

    def __init__(self, h, d_model, dropout=0.1):
        super().__init__()
        assert d_model % h == 0
        self.d_k = d_model // h
        self.h = h
        self.linears = clones(nn.Linear(d_model, d_model), 4)
        self.attn = None       # the reference caches P here with its graph; we only fill it on request
        self.dropout = nn.Dropout(p=dropout)

    def params(self):
        out = []
        for lin in self.linears:
            out += [lin.weight, lin.bias]
        return out

    def pack_qkv(self):
        """Lay the q, k, v projections (weights, biases and their gradients) out as row blocks of one matrix each.  The
        parameters keep their names, shapes and values; ops.AttnSublayerFn finds them adjacent and runs one GEMM for the
        three (two for source attention) in the forward, the data gradient and the weight gradient.  Call after the
        module is on its device and before the optimizer is built."""
        lins = self.linears[:3]
        d = lins[0].weight.shape[0]
        w = torch.cat([l.weight.data for l in lins], 0)
        b = torch.cat([l.bias.data for l in lins], 0)
        gw, gb = torch.zeros_like(w), torch.zeros_like(b)
        for i, l in enumerate(lins):
            l.weight = nn.Parameter(w[i * d:(i + 1) * d])
            l.bias = nn.Parameter(b[i * d:(i + 1) * d])
            l.weight.grad = gw[i * d:(i + 1) * d]
            l.bias.grad = gb[i * d:(i + 1) * d]

    def forward(self, query, key, value, mask=None, return_attn=False):
        B, Lq, _ = query.shape
        Lk = key.shape[1]
        spec = ops.mask_spec(mask, B, Lq, Lk, query.device)
        lin = self.linears
        q = ops.LinearFn.apply(query, lin[0].weight, lin[0].bias, False)
        k = ops.LinearFn.apply(key, lin[1].weight, lin[1].bias, False)
        if return_attn:
            with torch.no_grad():
                self.attn = ops.attention_weights(q, k, spec, self.h)
            return self.attn
        v = ops.LinearFn.apply(value, lin[2].weight, lin[2].bias, False)
        o = ops.AttnCoreFn.apply(q, k, v, spec, self.h, self._drop(self.dropout.p, 0, query.device))
        return ops.LinearFn.apply(o, lin[3].weight, lin[3].bias, False)


class PositionwiseFeedForward(_Sited):
    "w_2(dropout(relu(w_1(x))))  (reference :943-952)"

    def __init__(self, d_model, d_ff, dropout=0.1):
        super().__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        hmid = ops.LinearFn.apply(x, self.w_1.weight, self.w_1.bias, True)
        hmid = ops.ResidualDropoutFn.apply(hmid, None, self._drop(self.dropout.p, 0, x.device))
        return ops.LinearFn.apply(hmid, self.w_2.weight, self.w_2.bias, False)


def _attn_sublayer(x, mem, sub, attn, spec):
    dev = x.device
    return ops.AttnSublayerFn.apply(x, mem, sub.norm.a_2, sub.norm.b_2, *attn.params(), spec, attn.h,
                                    attn._drop(attn.dropout.p, 0, dev), sub._drop(sub.dropout.p, 0, dev),
                                    sub.norm.eps)


def _ffn_sublayer(x, sub, ff):
    dev = x.device
    return ops.FFNSublayerFn.apply(x, sub.norm.a_2, sub.norm.b_2, ff.w_1.weight, ff.w_1.bias, ff.w_2.weight,
                                   ff.w_2.bias, ff._drop(ff.dropout.p, 0, dev), sub._drop(sub.dropout.p, 0, dev),
                                   sub.norm.eps)


class EncoderLayer(nn.Module):
    "self-attention sublayer + feed-forward sublayer (reference :815-832)"

    def __init__(self, size, src_len, self_attn, feed_forward, dropout):
        super().__init__()
        self.size = size
        self.src_len = src_len
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.sublayer = clones(SublayerConnection(self.size, dropout), 2)

    def forward(self, x, mask, return_attn=False):
        B, L, _ = x.shape
        spec = ops.mask_spec(mask, B, L, L, x.device)
        if return_attn:
            # reference quirk (:827): the dumped weights use the UN-normalised layer input
            attn = self.self_attn(x, x, x, spec, return_attn=True)
        x = _attn_sublayer(x, None, self.sublayer[0], self.self_attn, spec)
        x = _ffn_sublayer(x, self.sublayer[1], self.feed_forward)
        return (x, attn) if return_attn else x


class DecoderLayer(nn.Module):
    "masked self-attention, source attention over the memory, feed-forward (reference :882-904)"

    def __init__(self, size, tgt_len, self_attn, src_attn, feed_forward, dropout):
        super().__init__()
        self.size = size
        self.tgt_len = tgt_len
        self.self_attn = self_attn
        self.src_attn = src_attn
        self.feed_forward = feed_forward
        self.sublayer = clones(SublayerConnection(self.size, dropout), 3)

    def forward(self, x, memory_key, memory_val, src_mask, tgt_mask, return_attn=False):
        if memory_key is not memory_val:
            raise ops._lib.TvaeError("DecoderLayer: memory_key and memory_val must be the same tensor "
                                     "(as in VAEDecoder.forward of the reference)")
        B, Lt, _ = x.shape
        Ls = memory_key.shape[1]
        tspec = ops.mask_spec(tgt_mask, B, Lt, Lt, x.device)
        sspec = ops.mask_spec(src_mask, B, Lt, Ls, x.device)
        x = _attn_sublayer(x, None, self.sublayer[0], self.self_attn, tspec)
        if return_attn:
            # reference quirk (:898): weights from the un-normalised x after the first sublayer
            src_attn = self.src_attn(x, memory_key, memory_val, sspec, return_attn=True)
        x = _attn_sublayer(x, memory_key, self.sublayer[1], self.src_attn, sspec)
        x = _ffn_sublayer(x, self.sublayer[2], self.feed_forward)
        return (x, src_attn) if return_attn else x


# ------------------------------------------------------------------------------ bottlenecks
class ConvBottleneck(nn.Module):
    "3 x relu(MaxPool1d(2)(Conv1d)): (B, size, 127) -> (B, 64, 9)  (reference :957-983)"

    def __init__(self, size):
        super().__init__()
        layers, in_d = [], size
        for i in range(3):
            out_d = 64 if i == 2 else int((in_d - 64) // 2 + 64)
            layers.append(nn.Sequential(nn.Conv1d(in_d, out_d, 9 if i == 0 else 8), nn.MaxPool1d(2)))
            in_d = out_d
        self.conv_layers = ListModule(*layers)

    def _params(self):
        out = []
        for seq in self.conv_layers:
            out += [seq[0].weight, seq[0].bias]
        return out

    def forward_token_major(self, x):
        "(B, L, size) token-major in -> (B, 576) in the reference's channel-major flattening"
        return ops.ConvBottleneckFn.apply(x, *self._params())

    def forward(self, x):
        "(B, size, L) as in the reference -> (B, 64, 9)"
        return self.forward_token_major(x.permute(0, 2, 1)).view(x.shape[0], 64, -1)


class DeconvBottleneck(nn.Module):
    "3 x relu(ConvTranspose1d(k=11, stride 4/3/1, padding 2)): (B, 64, 9) -> (B, size, 127)  (reference :985-1009)"

    def __init__(self, size):
        super().__init__()
        layers, in_d = [], 64
        for i in range(3):
            out_d = size if i == 2 else (size - in_d) // 4 + in_d
            stride = 1 if i == 2 else 4 - i
            layers.append(nn.Sequential(nn.ConvTranspose1d(in_d, out_d, 11, stride=stride, padding=2)))
            in_d = out_d
        self.deconv_layers = ListModule(*layers)

    def _params(self):
        out = []
        for seq in self.deconv_layers:
            out += [seq[0].weight, seq[0].bias]
        return out

    def forward_token_major(self, h):
        "(B, 576) -> (B, 127, size) token-major"
        return ops.DeconvBottleneckFn.apply(h, *self._params())

    def forward(self, x):
        "(B, 64, 9) -> (B, size, 127) as in the reference"
        return self.forward_token_major(x.reshape(x.shape[0], -1)).permute(0, 2, 1)


class PropertyPredictor(nn.Module):
    "depth_pp Linear layers, ReLU after every one including the last (reference :1013-1031)"

    def __init__(self, d_pp, depth_pp, d_latent):
        super().__init__()
        layers = []
        for i in range(depth_pp):
            if i == 0:
                layers.append(nn.Linear(d_latent, d_pp))
            elif i == depth_pp - 1:
                layers.append(nn.Linear(d_pp, 1))
            else:
                layers.append(nn.Linear(d_pp, d_pp))
        self.prediction_layers = ListModule(*layers)

    def forward(self, x):
        for layer in self.prediction_layers:
            x = ops.LinearFn.apply(x, layer.weight, layer.bias, True)
        return x


class Generator(nn.Module):
    "Linear(d_model, vocab-1): token logits (reference :740-747)"

    def __init__(self, d_model, vocab):
        super().__init__()
        self.proj = nn.Linear(d_model, vocab - 1)

    def forward(self, x):
        return ops.LinearFn.apply(x, self.proj.weight, self.proj.bias, False)


# ------------------------------------------------------------------------------ embeddings
class Embeddings(nn.Module):
    "lut[id] * sqrt(d_model)  (reference :1035-1043)"

    def __init__(self, d_model, vocab):
        super().__init__()
        self.lut = nn.Embedding(vocab, d_model)
        self.d_model = d_model

    def forward(self, x):
        zero_pe = torch.zeros((x.shape[1], self.d_model), device=self.lut.weight.device)
        return ops.EmbedFn.apply(x, self.lut.weight, zero_pe, None)


class PositionalEncoding(_Sited):
    "x + pe[:L] then dropout; sinusoidal table kept as the 'pe' buffer (reference :1045-1063)"

    def __init__(self, d_model, dropout, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0))

    def forward(self, x):
        return ops.PosEncFn.apply(x, self.pe[0], self._drop(self.dropout.p, 0, x.device))


class EmbedSequential(nn.Sequential):
    """nn.Sequential(Embeddings, PositionalEncoding) whose call runs both as ONE kernel
    (gather * sqrt(d) + pe, dropout).  Keys stay '0.lut.weight' / '1.pe'."""

    def forward(self, ids):
        emb, pos = self[0], self[1]
        return ops.EmbedFn.apply(ids, emb.lut.weight, pos.pe[0], pos._drop(pos.dropout.p, 0, ids.device))


# ------------------------------------------------------------------------------ encoder / decoder
class VAEEncoder(nn.Module):
    "N encoder layers -> norm -> conv bottleneck -> (mu, logvar) -> z, predicted length (reference :749-813)"

    def __init__(self, layer, N, d_latent, bypass_bottleneck, eps_scale):
        super().__init__()
        if bypass_bottleneck:
            raise NotImplementedError("bypass_bottleneck=True is broken in the reference too "
                                      "(pred_len undefined, trans_models.py:784-794)")
        self.layers = clones(layer, N)
        self.conv_bottleneck = ConvBottleneck(layer.size)
        self.z_means, self.z_var = nn.Linear(576, d_latent), nn.Linear(576, d_latent)
        self.norm = LayerNorm(layer.size)
        self.predict_len1 = nn.Linear(d_latent, d_latent * 2)
        self.predict_len2 = nn.Linear(d_latent * 2, d_latent)
        self.bypass_bottleneck = bypass_bottleneck
        self.eps_scale = eps_scale

    def _len_logits(self, z):
        y = ops.LinearFn.apply(z, self.predict_len1.weight, self.predict_len1.bias, False)
        return ops.LinearFn.apply(y, self.predict_len2.weight, self.predict_len2.bias, False)

    def predict_mask_length(self, mem):
        "arg max of the length head -> (n, 1) int64 (reference :763-769)"
        with torch.no_grad():
            logits = self._len_logits(mem)
            out = torch.empty(logits.shape[0], dtype=torch.int32, device=logits.device)
            ops.call("tvae_argmax_rows", ops._p(logits), logits.stride(0), logits.shape[1], ops._p(out),
                     logits.shape[0], ops._stream())
        return out.long().unsqueeze(1)

    def reparameterize(self, mu, logvar, eps_scale=1, eps=None):
        "mu + N(0,1) * eps_scale * exp(0.5 logvar); the normal draw comes from torch's generator"
        if eps is None:
            eps = torch.randn_like(mu)
        return ops.ReparamFn.apply(mu, logvar, eps, eps_scale)

    def _bottleneck(self, x, eps):
        mem = self.norm(x)
        flat = self.conv_bottleneck.forward_token_major(mem)
        mu = ops.LinearFn.apply(flat, self.z_means.weight, self.z_means.bias, False)
        logvar = ops.LinearFn.apply(flat, self.z_var.weight, self.z_var.bias, False)
        z = self.reparameterize(mu, logvar, self.eps_scale, eps)
        return z, mu, logvar, self._len_logits(mu)

    def forward(self, x, mask, eps=None):
        for layer in self.layers:
            x = layer(x, mask)
        return self._bottleneck(x, eps)

    def forward_w_attn(self, x, mask, eps=None):
        attn_wts = []
        for layer in self.layers:
            x, wts = layer(x, mask, return_attn=True)
            attn_wts.append(wts.detach().cpu())
        return self._bottleneck(x, eps) + (attn_wts,)


class VAEDecoder(nn.Module):
    "linear+deconv memory expansion -> final encoder layer -> norm -> N decoder layers -> norm (reference :834-880)"

    def __init__(self, encoder_layers, decoder_layers, N, d_latent, bypass_bottleneck):
        super().__init__()
        self.final_encodes = clones(encoder_layers, 1)
        self.layers = clones(decoder_layers, N)
        self.norm = LayerNorm(decoder_layers.size)
        self.bypass_bottleneck = bypass_bottleneck
        self.size = decoder_layers.size
        self.tgt_len = decoder_layers.tgt_len
        self.linear = nn.Linear(d_latent, 576)
        self.deconv_bottleneck = DeconvBottleneck(decoder_layers.size)

    def expand_memory(self, mem, src_mask, return_attn=False):
        "z (B, d_latent) -> normalised memory (B, 127, d) the source attention reads (reference :851-860)"
        h = ops.LinearFn.apply(mem, self.linear.weight, self.linear.bias, True)
        m = self.deconv_bottleneck.forward_token_major(h)
        wts = None
        for final_encode in self.final_encodes:
            if return_attn:
                m, wts = final_encode(m, src_mask, return_attn=True)
            else:
                m = final_encode(m, src_mask)
        m = self.norm(m)
        return (m, wts) if return_attn else m

    def forward(self, x, mem, src_mask, tgt_mask):
        mem = self.expand_memory(mem, src_mask)
        for layer in self.layers:
            x = layer(x, mem, mem, src_mask, tgt_mask)
        return self.norm(x)

    def forward_w_attn(self, x, mem, src_mask, tgt_mask):
        mem, deconv_wts = self.expand_memory(mem, src_mask, return_attn=True)
        src_attn_wts = []
        for layer in self.layers:
            x, wts = layer(x, mem, mem, src_mask, tgt_mask, return_attn=True)
            src_attn_wts.append(wts.detach().cpu())
        return self.norm(x), [deconv_wts.detach().cpu()], src_attn_wts


class EncoderDecoder(nn.Module):
    "encode -> decode -> generator (+ property head)  (reference :707-738)"

    def __init__(self, encoder, decoder, src_embed, tgt_embed, generator, property_predictor):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.src_embed = src_embed
        self.tgt_embed = tgt_embed
        self.generator = generator
        self.property_predictor = property_predictor

    def forward(self, src, tgt, src_mask, tgt_mask, eps=None):
        mem, mu, logvar, pred_len = self.encode(src, src_mask, eps)
        x = self.generator(self.decode(mem, src_mask, tgt, tgt_mask))
        prop = self.predict_property(mu) if self.property_predictor is not None else None
        return x, mu, logvar, pred_len, prop

    def encode(self, src, src_mask, eps=None):
        if self.training:
            ops.next_pass()              # fresh dropout masks for every training-mode forward
        return self.encoder(self.src_embed(src), src_mask, eps)

    def decode(self, mem, src_mask, tgt, tgt_mask):
        return self.decoder(self.tgt_embed(tgt), mem, src_mask, tgt_mask)

    def predict_property(self, mu):
        return self.property_predictor(mu)


def build_encoder_decoder(vocab_size, N=3, d_model=128, d_ff=512, d_latent=128, h=4, dropout=0.1,
                          bypass_bottleneck=False, property_predictor=False, d_pp=256, depth_pp=2,
                          eps_scale=1, src_len=126, tgt_len=125):
    """Assemble the module tree exactly as TransVAE.build_model does (reference :671-696)."""
    c = copy.deepcopy
    attn = MultiHeadedAttention(h, d_model)          # attention dropout stays 0.1 (reference :677, :910)
    ff = PositionwiseFeedForward(d_model, d_ff, dropout)
    position = PositionalEncoding(d_model, dropout)
    encoder = VAEEncoder(EncoderLayer(d_model, src_len, c(attn), c(ff), dropout), N, d_latent,
                         bypass_bottleneck, eps_scale)
    decoder = VAEDecoder(EncoderLayer(d_model, src_len, c(attn), c(ff), dropout),
                         DecoderLayer(d_model, tgt_len, c(attn), c(attn), c(ff), dropout),
                         N, d_latent, bypass_bottleneck)
    src_embed = EmbedSequential(Embeddings(d_model, vocab_size), c(position))
    tgt_embed = EmbedSequential(Embeddings(d_model, vocab_size), c(position))
    generator = Generator(d_model, vocab_size)
    pp = PropertyPredictor(d_pp, depth_pp, d_latent) if property_predictor else None
    model = EncoderDecoder(encoder, decoder, src_embed, tgt_embed, generator, pp)
    for p in model.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)
    for m in model.modules():                         # deep copies share nothing, dropout sites included
        if hasattr(m, "_new_sites"):
            m._new_sites()
    return model


# ------------------------------------------------------------------------------ model shell
import os  # noqa: E402
import collections  # noqa: E402

import numpy as np  # noqa: E402

from .data import DeviceDataset, DeviceLoader, make_src_mask, make_std_mask, vae_data_gen  # noqa: E402
from .decode import GreedyDecoder, GRAPH_MAX_N  # noqa: E402
from .loss import trans_vae_loss, vae_loss  # noqa: E402, F401
from .opt import Adam, NoamOpt  # noqa: E402
from .tvae_util import KLAnnealer, decode_mols  # noqa: E402

_DEFAULTS = (('BATCH_SIZE', 500), ('BATCH_CHUNKS', 5), ('BETA_INIT', 1e-8), ('BETA', 0.05),
             ('ANNEAL_START', 0), ('WARMUP_STEPS', 10000), ('EPS_SCALE', 1))
_LOG_HEADER = 'epoch,batch_idx,data_type,tot_loss,recon_loss,pred_loss,kld_loss,prop_mse_loss,run_time\n'


def _load_checkpoint(path):
    """torch.load on the CPU.  Tries the safe unpickler first (tensors, dicts, numpy scalars);
    a checkpoint that needs arbitrary unpickling is only opened when TVAE_TRUSTED_CKPT=1."""
    try:
        import numpy.core.multiarray as _ma  # noqa: F401
        allow = [np.float64, np.dtype, type(np.dtype(np.float64))]
        try:
            allow.append(np._core.multiarray.scalar)
        except AttributeError:
            allow.append(np.core.multiarray.scalar)
        with torch.serialization.safe_globals(allow):
            return torch.load(path, map_location=torch.device('cpu'), weights_only=True)
    except Exception as exc:
        if os.environ.get('TVAE_TRUSTED_CKPT') == '1':
            return torch.load(path, map_location=torch.device('cpu'), weights_only=False)
        raise ops._lib.TvaeError("checkpoint %s could not be read with the safe loader (%s); set "
                                 "TVAE_TRUSTED_CKPT=1 to unpickle a file you trust" % (path, exc))


class VAEShell():
    """Parameter defaults, training loop, logging, checkpoint save/load, sampling and decoding
    (reference transvae/trans_models.py:20-611; same method names and signatures)."""

    def __init__(self, params, name=None):
        self.params = params
        self.name = name
        for key, val in _DEFAULTS:
            if key not in self.params.keys():
                self.params[key] = val
        if 'LR' not in self.params.keys():
            # reference behaviour kept (trans_models.py:38-39): the key 'LR' is never set by the
            # CLI, so LR_SCALE is always forced to 1 whatever --lr_scale says.
            self.params['LR_SCALE'] = 1
        if 'CHAR_DICT' in self.params.keys():
            self.vocab_size = len(self.params['CHAR_DICT'].keys())
            self.pad_idx = self.params['CHAR_DICT']['_']
            if 'CHAR_WEIGHTS' in self.params.keys():
                self.params['CHAR_WEIGHTS'] = torch.as_tensor(self.params['CHAR_WEIGHTS']).detach().clone().float()
            else:
                self.params['CHAR_WEIGHTS'] = torch.ones(self.vocab_size, dtype=torch.float)
        self.loss_func = vae_loss          # as the reference (:49); the transformer's train loop calls trans_vae_loss itself
        self.data_gen = vae_data_gen
        self.src_len = 126
        self.tgt_len = 125
        self.n_epochs = 0
        self.best_loss = np.inf
        self.current_state = {'name': self.name, 'epoch': self.n_epochs, 'model_state_dict': None,
                              'optimizer_state_dict': None, 'best_loss': self.best_loss, 'params': self.params}
        self.loaded_from = None

    # ------------------------------------------------------------------ checkpoints
    def save(self, state, fn, path='checkpoints', use_name=True):
        "Write `state` as a .ckpt (reference :69-92: '<fn>_<name>.ckpt' under `path` unless use_name is False)"
        os.makedirs(path, exist_ok=True)
        if use_name:
            stem, ext = os.path.splitext(fn)
            if ext == '':
                ext = '.ckpt'
            if self.name is not None:
                stem += '_' + self.name
            save_path = os.path.join(path, stem + ext)
        else:
            save_path = fn
        torch.save(state, save_path)

    def load(self, checkpoint_path):
        "Restore params, weights and optimizer bookkeeping from a .ckpt (reference :94-124)"
        loaded = _load_checkpoint(checkpoint_path)
        self.loaded_from = checkpoint_path
        for k in self.current_state.keys():
            self.current_state[k] = loaded.get(k, None)
        if self.name is None:
            self.name = self.current_state['name']
        self.n_epochs = self.current_state['epoch']
        self.best_loss = self.current_state['best_loss']
        for k, v in self.current_state['params'].items():
            if k in self.arch_params or k not in self.params.keys():
                self.params[k] = v
        self.vocab_size = len(self.params['CHAR_DICT'].keys())
        self.pad_idx = self.params['CHAR_DICT']['_']
        self.build_model()
        self.model.load_state_dict(self.current_state['model_state_dict'])
        self.optimizer.load_state_dict(self.current_state['optimizer_state_dict'])

    # ------------------------------------------------------------------ training
    def _device(self):
        return next(self.model.parameters()).device

    def _chunk_inputs(self, chunk):
        mols = chunk[:, :-1]
        src = mols.long()
        tgt = src[:, :-1]
        return src, tgt, chunk[:, -1], make_src_mask(src, self.pad_idx), make_std_mask(tgt, self.pad_idx)

    def _model_forward(self, src, tgt, src_mask, tgt_mask, eps=None):
        """(x_out, mu, logvar, pred_len, pred_prop) of either model family: the RNN models have no length head and their
        loss no length term (reference :217-231: trans_vae_loss vs vae_loss), so pred_len is None there"""
        if self.model_type == 'transformer':
            return self.model(src, tgt, src_mask, tgt_mask, eps)
        x_out, mu, logvar, pred_prop = self.model(src, tgt, src_mask, tgt_mask, eps)
        return x_out, mu, logvar, None, pred_prop

    def _encode(self, src, src_mask):
        "(mem, mu, logvar) of either model family (reference :486-489, :593-596)"
        if self.model_type == 'transformer':
            return self.model.encode(src, src_mask)[:3]
        return self.model.encode(src)

    def _forward_backward(self, data, beta, eps, n_groups, table, row0):
        src, tgt, true_prop, src_mask, tgt_mask = self._chunk_inputs(data)
        x_out, mu, logvar, pred_len, pred_prop = self._model_forward(src, tgt, src_mask, tgt_mask, eps)
        out5, g = ops.loss_fwd_bwd(src, x_out, self.params['CHAR_WEIGHTS'], pred_len, mu, logvar, pred_prop,
                                   true_prop, beta, self.pad_idx, n_groups=n_groups)
        table[row0:row0 + n_groups].copy_(out5.view(n_groups, 5))
        outs = [x_out, mu, logvar]
        grads = [g['logits'].view(x_out.shape), g['mu'], g['logvar']]
        if pred_len is not None:
            outs.append(pred_len)
            grads.append(g['pred_len'])
        if pred_prop is not None:
            outs.append(pred_prop)
            grads.append(g['pred_prop'].view(pred_prop.shape))
        return outs, grads

    def train_step(self, data, beta, eps=None, dp=None):
        """One optimizer step on a (BATCH_SIZE, 128) batch.  The reference runs BATCH_CHUNKS
        forward/backward passes whose gradients ACCUMULATE (no 1/chunks scaling, :202-236); here
        the whole batch goes through the network ONCE and the loss kernel normalises every chunk
        on its own, which yields the same gradient sum with 5x fewer, 5x larger launches
        (params['FUSE_CHUNKS'] = False restores the literal chunk loop).  Weight gradients are
        accumulated by the kernels directly into param.grad, NoamOpt.step follows, and the
        gradient reset is fused into the Adam launch.  Returns the (chunks, 5) loss table on the
        device; nothing is copied to the host here.  `dp` (parallel.DataParallel) adds the
        bucketed gradient all-reduce, overlapped with the (last) backward."""
        dev = self._device()
        data = data.to(dev, non_blocking=True)
        chunks = self.params['BATCH_CHUNKS']
        csize = data.shape[0] // chunks
        table = torch.empty((chunks, 5), device=dev, dtype=torch.float32)
        sink = dp.sink if dp is not None else ops.GradSink()
        sink.pending.clear()
        ops.begin_step()
        with ops.direct_grads(sink):
            if self.params.get('FUSE_CHUNKS', True) and csize * chunks == data.shape[0]:
                outs, grads = self._forward_backward(data, beta, eps, chunks, table, 0)
                if dp is not None:
                    dp.arm()
                torch.autograd.backward(outs, grads)
            else:
                for i in range(chunks):
                    sl = slice(i * csize, (i + 1) * csize)
                    outs, grads = self._forward_backward(data[sl], beta, None if eps is None else eps[sl], 1, table, i)
                    if dp is not None and i == chunks - 1:
                        dp.arm()
                    torch.autograd.backward(outs, grads)
        if dp is not None:
            dp.finish()
        self.optimizer.step()
        if self.model.training:
            ops.advance_rng(dev)
        return table

    def _data_parallel(self):
        "the shell's DataParallel engine: one per shell, rebuilt only when the world size changes"
        import torch.distributed as dist
        from .parallel import DataParallel
        dp = getattr(self, '_dp', None)
        if dp is not None and (dp.closed or dp.world != dist.get_world_size()):
            dp.close()
            dp = None
        if dp is None:
            dp = self._dp = DataParallel(self)
        return dp

    @torch.no_grad()
    def eval_step(self, data, beta):
        "Forward + loss on a batch without gradients (the reference's validation pass, :267-307)"
        dev = self._device()
        data = data.to(dev, non_blocking=True)
        chunks = self.params['BATCH_CHUNKS']
        if data.shape[0] % chunks != 0:
            chunks = 1
        src, tgt, true_prop, src_mask, tgt_mask = self._chunk_inputs(data)
        x_out, mu, logvar, pred_len, pred_prop = self._model_forward(src, tgt, src_mask, tgt_mask)
        out5, _ = ops.loss_fwd_bwd(src, x_out, self.params['CHAR_WEIGHTS'], pred_len, mu, logvar, pred_prop,
                                   true_prop, beta, self.pad_idx, want_grads=False, n_groups=chunks)
        return out5.view(chunks, 5)

    def train(self, train_mols, val_mols, train_props=None, val_props=None,
              epochs=100, save=True, save_freq=None, log=True, log_dir='trials'):
        "Train and validate (reference :126-352: same batching, CSV log rows and checkpoint cadence)"
        train_data = self.data_gen(train_mols, train_props, char_dict=self.params['CHAR_DICT'])
        val_data = self.data_gen(val_mols, val_props, char_dict=self.params['CHAR_DICT'])
        # targets are ids 1 .. vocab-1 shifted down by one (loss.py:26); F.cross_entropy of the reference raises on
        # anything else, the loss kernel would give it zero weight: refuse bad vocabularies / data once, here
        for what, d in (('train', train_data), ('validation', val_data)):
            ids = d[:, 1:self.src_len + 1]
            if ids.numel() and (float(ids.min()) < 1 or float(ids.max()) > self.vocab_size - 1):
                raise IndexError("Target {} is out of bounds. ({} data: token ids after <start> must lie in 1..{})".format(
                    int(ids.min() if float(ids.min()) < 1 else ids.max()) - 1, what, self.vocab_size - 1))
        # one process per GPU (torch.distributed initialised by the launcher): every rank takes its own
        # shard of each epoch's shuffle and BATCH_SIZE is per rank; rank 0 logs and writes checkpoints
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        dp = None
        if world > 1:
            from .parallel import DataParallel, mean_over_ranks
            dp = self._data_parallel()
            log, save = log and rank == 0, save and rank == 0
        # dropout stream: continue the one a checkpoint carried; else, ONCE per process and device, follow torch's seed
        # (+ rank, so that the replicas of a data-parallel run do not all drop the same units) unless the caller seeded
        # it (ops.seed_rng).  A later train() call -- train(epochs=1) in a loop, or again after a resume -- continues
        # the stream {seed, step, pass} where the previous call left it instead of replaying the masks from step 0.
        if self.current_state['params'].get('RNG_STATE') is not None and self.loaded_from is not None \
                and not getattr(self, '_rng_resumed', False):
            ops.rng_restore(self._device(), self.current_state['params']['RNG_STATE'])
            self._rng_resumed = True
        elif not ops.rng_seeded(self._device()):
            ops.seed_rng(self._device(), (torch.initial_seed() + rank) & 0x7FFFFFFF, explicit=False)

        # the encoded molecules move to HBM once (one byte per token); a batch is one gather launch in the
        # order the reference's DataLoader (or a DistributedSampler) would have produced (data.DeviceLoader)
        def loader(d):
            return DeviceLoader(DeviceDataset.from_tensor(d, self._device()), self.params['BATCH_SIZE'], world, rank)
        train_iter, val_iter = loader(train_data), loader(val_data)
        self.chunk_size = self.params['BATCH_SIZE'] // self.params['BATCH_CHUNKS']
        if save_freq is None:
            save_freq = epochs
        log_fn = None
        if log:
            os.makedirs(log_dir, exist_ok=True)
            log_fn = '{}/log{}.txt'.format(log_dir, '_' + self.name if self.name is not None else '')
            if not os.path.exists(log_fn):
                with open(log_fn, 'a') as f:
                    f.write(_LOG_HEADER)
        kl_annealer = KLAnnealer(self.params['BETA_INIT'], self.params['BETA'], epochs, self.params['ANNEAL_START'])

        def run(iterator, step_fn, tag, beta, with_time):
            # The loss table of a step is read back two steps late (pinned buffer + event), so the host keeps
            # queueing work while the GPU runs; run_time is the GPU time from the end of the previous step
            # to the end of this one, i.e. what the reference's perf_counter pair brackets (:201-240).
            means, pending = [], collections.deque()
            chunks = self.params['BATCH_CHUNKS']
            hosts = [torch.empty((chunks, 5), dtype=torch.float32).pin_memory() for _ in range(3)]

            def flush(keep):
                while len(pending) > keep:
                    j, host, rows, start, done = pending.popleft()
                    done.synchronize()
                    avg = host[:rows].numpy().mean(0)
                    means.append(avg[0])
                    if log:
                        row = [self.n_epochs, j, tag, avg[0], avg[1], avg[2], avg[3], avg[4]]
                        if with_time:            # the reference's test rows drop run_time (:322-329)
                            row.append(round(start.elapsed_time(done) * 1e-3, 5))
                        with open(log_fn, 'a') as f:
                            f.write(','.join(str(v) for v in row) + '\n')

            last = torch.cuda.Event(enable_timing=True)
            last.record()
            for j, data in enumerate(iterator):
                table = step_fn(data, beta)
                host = hosts[j % 3]
                host[:table.shape[0]].copy_(table, non_blocking=True)
                done = torch.cuda.Event(enable_timing=True)
                done.record()
                pending.append((j, host, table.shape[0], last, done))
                last = done
                flush(2)
            flush(0)
            return np.mean(means) if means else float('nan')

        train_fn = self.train_step if dp is None else (lambda data, beta: self.train_step(data, beta, dp=dp))
        for epoch in range(epochs):
            self.model.train()
            beta = kl_annealer(epoch)
            if world > 1:
                train_iter.set_epoch(self.n_epochs)
                val_iter.set_epoch(self.n_epochs)
            train_loss = run(train_iter, train_fn, 'train', beta, True)
            self.model.eval()
            val_loss = run(val_iter, self.eval_step, 'test', beta, False)
            if world > 1:                   # every rank must take the same best-checkpoint decision
                train_loss, val_loss = mean_over_ranks([train_loss, val_loss])
            self.n_epochs += 1
            print('Epoch - {} Train - {} Val - {} KLBeta - {}'.format(self.n_epochs, train_loss, val_loss, beta))
            self.current_state['epoch'] = self.n_epochs
            snap = ops.rng_snapshot(self._device())                          # a resumed run continues the mask stream
            self.params['RNG_STATE'] = snap
            if isinstance(self.current_state.get('params'), dict):           # after load() this is the LOADED dict
                self.current_state['params']['RNG_STATE'] = snap
            self.current_state['model_state_dict'] = self.model.state_dict()
            self.optimizer.sync_state_dict()
            self.current_state['optimizer_state_dict'] = self.optimizer.state_dict
            if val_loss < self.best_loss:
                self.best_loss = float(val_loss)
                self.current_state['best_loss'] = self.best_loss
                if save:
                    self.save(self.current_state, 'best')
            if (self.n_epochs) % save_freq == 0 and save:
                self.save(self.current_state, str(self.n_epochs).zfill(3))
        if dp is not None:
            dp.close()          # hooks off, grad_scale back to 1: a later single-process train_step is unscaled

    # ------------------------------------------------------------------ sampling / decoding
    def sample_from_memory(self, size, mode='rand', sample_dims=None, k=5):
        "z ~ N(0, I) on all ('rand'), the given ('top_dims') or k randomly chosen ('k_dims') dimensions (reference :355-381)"
        dl = self.params['d_latent']
        if mode == 'rand':
            return torch.randn(size, dl)
        assert sample_dims is not None, "ERROR: Must provide sample dimensions"
        z = torch.zeros((size, dl))
        dims = sample_dims if mode == 'top_dims' else np.random.choice(sample_dims, size=k, replace=False)
        for d in dims:
            z[:, d] = torch.randn(size)
        return z

    def greedy_decode(self, mem, src_mask=None, condition=[]):
        """Greedy token ids (n, 125) from latent vectors (reference :383-436), computed by the
        KV-cached decoder.  `src_mask` (n,1,127), when given, replaces the predicted-length mask."""
        dev = self._device()
        mem = mem.to(dev).float()
        n = mem.shape[0]
        if self.model_type != 'transformer':
            return self._rnn_greedy_decode(mem, condition)
        if src_mask is None:
            lens = self.model.encoder.predict_mask_length(mem)            # (n, 1) int64, may be 0
            key_mask = (torch.arange(self.src_len + 1, device=dev)[None, :] < lens).to(torch.uint8).contiguous()
        else:
            key_mask = ops.mask_spec(src_mask, n, 1, self.src_len + 1, dev).key_mask
            if key_mask is None:
                raise ops._lib.TvaeError("greedy_decode: src_mask must be (n, 1, 127)")
        cond = [self.params['CHAR_DICT'][tok] for tok in condition]
        was_training = self.model.training
        self.model.eval()
        try:
            return GreedyDecoder(self.model, self.params['h']).run(
                mem, key_mask, self.params['CHAR_DICT']['<start>'], cond, self.tgt_len)
        finally:
            self.model.train(was_training)

    DECODE_SLAB = 65536     # latents in flight in decode_latents: ~110 GB of K/V caches for trans4x-256

    @torch.no_grad()
    def decode_latents(self, mem, condition=[], slab=None):
        """greedy_decode over ANY number of latents, `slab` rows (default DECODE_SLAB) in flight at a time; the token ids come
        back as ONE (n, 125) uint8 tensor on the device (the vocabulary has 26 entries: 1 M molecules are 125 MB instead of
        1 GB of int64).  Rows are independent and the kernels sum in a fixed order, so the tokens equal those of
        greedy_decode on any other split of `mem`.
          On the GPU a transformer's slab is decoded as FOUR quarter slabs whose step loops are issued in turn on four
        streams (GreedyDecoder.run_iter): one piece's attention over its K/V caches is HBM-bound, another's projections
        are MFMA-bound, and they overlap -- measured on 262,144 latents: 35.9 / 36.0 k molecules/s on one stream,
        36.5-37.4 k with two pieces, 37.3-37.8 k with three, 38.0 / 38.1 k with four (tools/debug/ab_decode.sh).
        TVAE_DECODE_STREAMS=1..4 sets the number; 1 decodes whole slabs one after the other on the current stream."""
        slab = int(slab or os.environ.get("TVAE_DECODE_SLAB") or self.DECODE_SLAB)
        dev = self._device()
        n = mem.shape[0]
        out = torch.empty((n, self.tgt_len), dtype=torch.uint8, device=dev)
        ways = max(1, min(4, int(os.environ.get("TVAE_DECODE_STREAMS", "4"))))
        half = (slab + ways - 1) // ways
        if self.model_type != 'transformer' or dev.type != 'cuda' or half <= GRAPH_MAX_N or n <= half or ways == 1:
            for s0 in range(0, n, slab):
                out[s0:s0 + slab] = self.greedy_decode(mem[s0:s0 + slab], condition=condition).to(torch.uint8)
            return out
        mem = mem.to(dev).float()
        cond = [self.params['CHAR_DICT'][tok] for tok in condition]
        start = self.params['CHAR_DICT']['<start>']
        main = torch.cuda.current_stream(dev)
        streams = (main, ops.second_stream(dev)) + tuple(ops.extra_stream(dev, k) for k in range(ways - 2))
        was_training = self.model.training
        self.model.eval()
        try:
            decoder = GreedyDecoder(self.model, self.params['h'])
            for p0 in range(0, n, ways * half):
                pieces = [(a, min(a + half, n)) for a in range(p0, p0 + ways * half, half) if a < n]
                for st in streams[1:]:
                    st.wait_stream(main)                     # inputs and the last round's buffers are behind us
                gens = []
                for (a, b), st in zip(pieces, streams):
                    with torch.cuda.stream(st):
                        z = mem[a:b]
                        lens = self.model.encoder.predict_mask_length(z)
                        key_mask = (torch.arange(self.src_len + 1, device=dev)[None, :] < lens).to(torch.uint8).contiguous()
                        gens.append(decoder.run_iter(z, key_mask, start, cond, self.tgt_len))
                live = list(range(len(gens)))
                while live:
                    for k in list(live):
                        with torch.cuda.stream(streams[k]):
                            ids = next(gens[k], None)
                            if ids is not None:              # the generator's last value
                                a, b = pieces[k]
                                out[a:b] = ids.to(torch.uint8)
                                live.remove(k)
                                gens[k] = None               # releases the slab's caches (to that stream's pool)
                for st in streams[1:]:
                    main.wait_stream(st)
        finally:
            self.model.train(was_training)
        return out

    @torch.no_grad()
    def _rnn_greedy_decode(self, mem, condition=[]):
        """Greedy decoding of the RNN models.  The reference re-runs the whole decoder over the 126-token buffer for every
        new token and reads position i (:425-431); a GRU is causal, so the same tokens come from ONE pass that carries
        the hidden states from step to step: per step the embedded previous token and memory row i go through one GRU
        step per layer, then norm -> generator -> arg max."""
        dev = self._device()
        model = self.model
        was_training = model.training
        model.eval()
        try:
            dec = model.decoder
            n, max_len = mem.shape[0], self.tgt_len
            rows = dec.expand_memory(mem)                                   # (n, 126, size)
            tgt = torch.full((n, max_len + 1), self.params['CHAR_DICT']['<start>'], dtype=torch.long, device=dev)
            for j, tok in enumerate(condition):
                tgt[:, j + 1] = self.params['CHAR_DICT'][tok]
            h = [torch.zeros((n, dec.size), device=dev) for _ in range(dec.n_layers)]
            nxt = torch.empty(n, dtype=torch.int32, device=dev)
            for i in range(max_len):
                x = rows[:, i, :]
                if dec.teacher_force:
                    x = torch.cat((model.src_embed(tgt[:, i:i + 1])[:, 0, :], x), dim=1)
                h = dec.gru.step(x.contiguous(), h)
                if i >= len(condition):
                    logits = model.generator(dec.norm(h[-1]))
                    ops.call("tvae_argmax_rows", ops._p(logits), logits.stride(0), logits.shape[1], ops._p(nxt), n,
                             ops._stream())
                    tgt[:, i + 1] = nxt.long() + 1
            return tgt[:, 1:]
        finally:
            model.train(was_training)

    def _encode_batches(self, data):
        "yield (offset, chunk data, src, src_mask) over full batches, as reconstruct / calc_mems iterate (reference :459-483)"
        bs, chunks = self.params['BATCH_SIZE'], self.params['BATCH_CHUNKS']
        self.batch_size, self.chunk_size = bs, bs // chunks
        dev = self._device()
        for j in range(data.shape[0] // bs):
            batch = data[j * bs:(j + 1) * bs].to(dev)
            for i in range(chunks):
                chunk = batch[i * self.chunk_size:(i + 1) * self.chunk_size]
                src = chunk[:, :-1].long()
                yield j, j * bs + i * self.chunk_size, src, make_src_mask(src, self.pad_idx)

    @torch.no_grad()
    def reconstruct(self, data, method='greedy', log=True, return_mems=True, return_str=True):
        "encode SMILES -> z -> greedy decode with the TRUE source mask (reference :438-509)"
        data = vae_data_gen(data, props=None, char_dict=self.params['CHAR_DICT'])
        self.model.eval()
        decoded_smiles = []
        mems = torch.empty((data.shape[0], self.params['d_latent']))
        last_j = -1
        for j, off, src, src_mask in self._encode_batches(data):
            if log and j != last_j:
                os.makedirs('calcs', exist_ok=True)
                with open('calcs/{}_progress.txt'.format(self.name), 'a') as f:
                    f.write('{}\n'.format(j))
                last_j = j
            _, mem, _ = self._encode(src, src_mask)              # reference quirk: keeps mu as "mem" (:487)
            mems[off:off + mem.shape[0]] = mem.cpu()
            decoded = self.greedy_decode(mem, src_mask=src_mask) if method == 'greedy' else None
            if return_str:
                decoded_smiles += decode_mols(decoded, self.params['ORG_DICT'])
            else:
                decoded_smiles.append(decoded)
        return (decoded_smiles, mems.numpy()) if return_mems else decoded_smiles

    def sample(self, n, method='greedy', sample_mode='rand', sample_dims=None, k=None, return_str=True,
               condition=[]):
        "sample z, decode greedily, optionally translate to SMILES strings (reference :511-545)"
        mem = self.sample_from_memory(n, mode=sample_mode, sample_dims=sample_dims, k=k)
        decoded = self.greedy_decode(mem, condition=condition) if method == 'greedy' else None
        return decode_mols(decoded, self.params['ORG_DICT']) if return_str else decoded

    @torch.no_grad()
    def calc_mems(self, data, log=True, save_dir='memory', save_fn='model_name', save=True):
        "z, mu, logvar for every full batch of `data` (reference :547-611)"
        data = vae_data_gen(data, props=None, char_dict=self.params['CHAR_DICT'])
        n_full = (data.shape[0] // self.params['BATCH_SIZE']) * self.params['BATCH_SIZE']
        out = [torch.empty((n_full, self.params['d_latent'])) for _ in range(3)]
        self.model.eval()
        last_j = -1
        for j, off, src, src_mask in self._encode_batches(data):
            if log and j != last_j:
                os.makedirs('memory', exist_ok=True)
                with open('memory/{}_progress.txt'.format(self.name), 'a') as f:
                    f.write('{}\n'.format(j))
                last_j = j
            mem, mu, logvar = self._encode(src, src_mask)
            for buf, t in zip(out, (mem, mu, logvar)):
                buf[off:off + t.shape[0]] = t.cpu()
        mems, mus, logvars = (t.numpy() for t in out)
        if not save:
            return mems, mus, logvars
        if save_fn == 'model_name':
            save_fn = self.name
        os.makedirs(save_dir, exist_ok=True)
        save_path = os.path.join(save_dir, save_fn)
        np.save('{}_mems.npy'.format(save_path), mems)
        np.save('{}_mus.npy'.format(save_path), mus)
        np.save('{}_logvars.npy'.format(save_path), logvars)


class TransVAE(VAEShell):
    "Transformer VAE with a conv/deconv latent bottleneck (reference transvae/trans_models.py:616-705)"

    def __init__(self, params={}, name=None, N=3, d_model=128, d_ff=512,
                 d_latent=128, h=4, dropout=0.1, bypass_bottleneck=False,
                 property_predictor=False, d_pp=256, depth_pp=2, load_fn=None):
        super().__init__(params, name)
        self.model_type = 'transformer'
        self.params['model_type'] = self.model_type
        arch = dict(N=N, d_model=d_model, d_ff=d_ff, d_latent=d_latent, h=h, dropout=dropout,
                    bypass_bottleneck=bypass_bottleneck, property_predictor=property_predictor,
                    d_pp=d_pp, depth_pp=depth_pp)
        self.params.update(arch)
        self.arch_params = list(arch.keys())
        if load_fn is None:
            self.build_model()
        else:
            self.load(load_fn)

    def build_model(self):
        "module tree + xavier init + device placement + NoamOpt(Adam) (reference :671-705)"
        p = self.params
        self.model = build_encoder_decoder(self.vocab_size, p['N'], p['d_model'], p['d_ff'], p['d_latent'],
                                           p['h'], p['dropout'], p['bypass_bottleneck'], p['property_predictor'],
                                           p['d_pp'], p['depth_pp'], p['EPS_SCALE'], self.src_len, self.tgt_len)
        self.use_gpu = torch.cuda.is_available()
        if self.use_gpu:
            self.model.cuda()
            self.params['CHAR_WEIGHTS'] = self.params['CHAR_WEIGHTS'].cuda()
            if os.environ.get("TVAE_PACK_QKV", "1") == "1":       # 0: keep the projections as separately allocated matrices
                for m in self.model.modules():
                    if isinstance(m, MultiHeadedAttention):
                        m.pack_qkv()
        adam = Adam(self.model.parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9)
        adam.fuse_zero_grad = True
        self.optimizer = NoamOpt(p['d_model'], p['LR_SCALE'], p['WARMUP_STEPS'], adam)
