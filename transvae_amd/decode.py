"""KV-cached greedy decoder: replaces the O(L^2) full-recompute loop of VAEShell.greedy_decode
(reference transvae/trans_models.py:383-436) with one memory expansion + 125 single-token steps.

Per step and decoder layer: LayerNorm -> one fused QKV GEMM (M = molecules) -> append K,V to the
cache -> one-query attention over the cache -> output projection (+residual) -> LayerNorm ->
query projection -> one-query attention over the PRE-PROJECTED memory keys/values -> projection
-> LayerNorm -> FFN.  The arg-max is taken on the device; token ids never visit the host.
The arithmetic per position is the same as the reference's (causal masking makes position i
independent of later positions), which is what makes the token sequences bit-identical.
"""
import collections
import math

import torch

from . import ops
from .ops import _p, _stream, call


GRAPH_MAX_N = 2048      # latent batches up to this size are launch-bound: their step loop is replayed as a HIP graph
GRAPH_CACHE_MAX = 4     # captured graphs kept per model (least recently used goes first): a graph owns its K/V caches,
#                         ~1.5 MB per latent for trans4x-256, so an unbounded cache grows by GBs with every new batch size


def clear_decode_graphs(model):
    "drop every captured decode graph of `model` (and the HBM their private pools hold)"
    model.__dict__.pop("_tvae_decode_graphs", None)


def _weights_stamp(model):
    "changes whenever a weight may have changed (in-place updates bump _version; .data writes bump the epoch)"
    return (ops._weights_epoch[0], sum(p._version for p in model.parameters()), ops.compute_dtype())


class GreedyDecoder:
    def __init__(self, model, h):
        self.model, self.h = model, h

    @staticmethod
    def _ln(x, norm):
        return ops.layernorm_fwd(x, norm.a_2, norm.b_2, norm.eps)

    @torch.no_grad()
    def run(self, z, key_mask, start_id, cond_ids=(), max_len=125):
        """z (n, d_latent) on the device; key_mask (n, 127) uint8 source mask; returns (n, max_len) int64.

        Small batches (n <= GRAPH_MAX_N) spend their time in ~5000 launches of tiny kernels, so from the second
        call with the same shape on the projections + 125 steps are captured once into a HIP graph (static
        buffers for the memory, the mask and the ids) and replayed; the arithmetic and the kernels are the same."""
        model = self.model
        n = z.shape[0]
        spec = ops.MaskSpec(key_mask=key_mask)
        mem = model.decoder.expand_memory(z, spec)            # (n, 127, d), already through decoder.norm
        cond_ids = tuple(int(t) for t in cond_ids)
        if n > GRAPH_MAX_N or not torch.cuda.is_available():
            return self._steps(mem, key_mask, None, start_id, cond_ids, max_len)[:, 1:]
        key = (n, tuple(mem.shape), cond_ids, int(start_id), max_len, z.device.index)
        cache = model.__dict__.setdefault("_tvae_decode_graphs", {"graphs": collections.OrderedDict(), "seen": {}})
        graphs, seen = cache["graphs"], cache["seen"]
        stamp = _weights_stamp(model)
        entry = graphs.get(key)
        if entry is not None and entry["stamp"] != stamp:
            del graphs[key]                                   # weights moved since the capture: take it again
            entry = None
        if entry is None:
            if seen.get(key) != stamp:                        # first call with this shape: plain launches
                if len(seen) > 64:
                    seen.clear()
                seen[key] = stamp
                return self._steps(mem, key_mask, None, start_id, cond_ids, max_len)[:, 1:]
            while len(graphs) >= GRAPH_CACHE_MAX:
                graphs.popitem(last=False)                    # least recently used graph and its pool
            entry = {"stamp": stamp, "mem": torch.empty_like(mem), "mask": torch.empty_like(key_mask),
                     "ids": torch.empty((n, max_len + 1), dtype=torch.int64, device=z.device)}
            entry["mem"].copy_(mem)
            entry["mask"].copy_(key_mask)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._steps(entry["mem"], entry["mask"], entry["ids"], start_id, cond_ids, max_len)
            entry["graph"] = graph
            graphs[key] = entry
        graphs.move_to_end(key)
        entry["mem"].copy_(mem)
        entry["mask"].copy_(key_mask)
        entry["graph"].replay()
        return entry["ids"][:, 1:].clone()

    def _steps(self, mem, key_mask, ids, start_id, cond_ids, max_len):
        "source K/V projections + max_len single-token steps; every launch goes to the current stream"
        for ids in self._steps_iter(mem, key_mask, ids, start_id, cond_ids, max_len):
            pass
        return ids

    def run_iter(self, z, key_mask, start_id, cond_ids=(), max_len=125):
        """run() for a large slab as a generator: yields after the set-up and after every token step, the last value is the
        (n, max_len) ids.  VAEShell.decode_latents resumes several of these in turn, each under its own stream, so that one
        piece's attention over its K/V caches (HBM-bound) runs beside another's projections (MFMA-bound).  The caller must
        make the generator's stream current around every next()."""
        spec = ops.MaskSpec(key_mask=key_mask)
        mem = self.model.decoder.expand_memory(z, spec)
        ids = None
        for ids in self._steps_iter(mem, key_mask, None, start_id, tuple(int(t) for t in cond_ids), max_len):
            yield None
        yield ids[:, 1:]

    def _steps_iter(self, mem, key_mask, ids, start_id, cond_ids, max_len):
        "_steps as a generator (yields the ids buffer after the set-up and after every step)"
        model, h = self.model, self.h
        dec = model.decoder
        dev = mem.device
        n, Ls, d = mem.shape
        dk = d // h
        scale = 1.0 / math.sqrt(dk)
        mem2 = mem.reshape(n * Ls, d)
        T = max_len + 1
        layers = list(dec.layers)
        # source-attention keys/values never change during decoding: project them once
        kc, vc, wqkv, bqkv = [], [], [], []
        for lyr in layers:
            lin = lyr.src_attn.linears
            kc.append(ops.linear_fwd(mem2, lin[1].weight, lin[1].bias))
            vc.append(ops.linear_fwd(mem2, lin[2].weight, lin[2].bias))
            sl = lyr.self_attn.linears
            wqkv.append(torch.cat([sl[0].weight, sl[1].weight, sl[2].weight], 0).contiguous())
            bqkv.append(torch.cat([sl[0].bias, sl[1].bias, sl[2].bias], 0).contiguous())
        ks = [torch.empty((n, T, d), device=dev, dtype=torch.float32) for _ in layers]
        vs = [torch.empty((n, T, d), device=dev, dtype=torch.float32) for _ in layers]
        if ids is None:
            ids = torch.empty((n, T), dtype=torch.int64, device=dev)
        ids.fill_(int(start_id))
        for j, t in enumerate(cond_ids):
            ids[:, j + 1] = int(t)
        emb, pos = model.tgt_embed[0], model.tgt_embed[1]
        lut, pe = emb.lut.weight, pos.pe[0]
        vocab = lut.shape[0]
        gen = model.generator.proj
        n_cls = gen.weight.shape[0]
        x = torch.empty((n, d), device=dev, dtype=torch.float32)
        qkv = torch.empty((n, 3 * d), device=dev, dtype=torch.float32)
        att = torch.empty((n, d), device=dev, dtype=torch.float32)
        yield ids
        for t in range(max_len):
            st = _stream()               # a generator may be resumed under another stream context than it was started in
            call("tvae_embed_step", _p(ids), T, t, _p(lut), _p(pe), _p(x), n, d, vocab, st)
            cur = x
            for li, lyr in enumerate(layers):
                sub = lyr.sublayer
                n0 = self._ln(cur, sub[0].norm)
                ops.linear_fwd(n0, wqkv[li], bqkv[li], out=qkv)
                call("tvae_scatter_rows", _p(qkv[:, d:]), 3 * d, _p(ks[li]), d, T, t, n, d, st)
                call("tvae_scatter_rows", _p(qkv[:, 2 * d:]), 3 * d, _p(vs[li]), d, T, t, n, d, st)
                call("tvae_decode_attn", _p(qkv), 3 * d, _p(ks[li]), _p(vs[li]), d, T, _p(att), d, None, 0,
                     t + 1, n, h, dk, scale, st)
                so = lyr.self_attn.linears[3]
                cur = ops.linear_fwd(att, so.weight, so.bias, residual=cur)
                n1 = self._ln(cur, sub[1].norm)
                cl = lyr.src_attn.linears
                q = ops.linear_fwd(n1, cl[0].weight, cl[0].bias)
                call("tvae_decode_attn", _p(q), d, _p(kc[li]), _p(vc[li]), d, Ls, _p(att), d, _p(key_mask), Ls,
                     Ls, n, h, dk, scale, st)
                cur = ops.linear_fwd(att, cl[3].weight, cl[3].bias, residual=cur)
                n2 = self._ln(cur, sub[2].norm)
                ff = lyr.feed_forward
                hmid = ops.linear_fwd(n2, ff.w_1.weight, ff.w_1.bias, relu=True)
                cur = ops.linear_fwd(hmid, ff.w_2.weight, ff.w_2.bias, residual=cur)
            if t >= len(cond_ids):
                y = self._ln(cur, dec.norm)
                logits = ops.linear_fwd(y, gen.weight, gen.bias)
                call("tvae_argmax_next", _p(logits), n_cls, n_cls, _p(ids), T, t + 1, n, st)
            yield ids
