"""Deterministic synthetic checkpoints and SMILES-shaped token batches.

The published TransVAE checkpoints are git-LFS pointers, so every
parity test, golden fixture and benchmark in this repo runs on a *synthetic checkpoint*:
for each ``state_dict`` key the values come from a generator seeded by crc32(key) ^ seed,
which makes the tensor bits independent of module construction order and identical in the
build container and on the GPU box (same numpy).  Scales follow what
``TransVAE.build_model`` does (reference transvae/trans_models.py:694-696: xavier_uniform
on every dim>1 tensor) with LayerNorm gains/offsets perturbed so that gradients through
them are informative.

Nothing here touches the reference or the oracle; the same functions are used by
tools/make_goldens.py (container, beside the reference) and by tests/bench on the GPU box.
"""
import json
import os
import zlib

import numpy as np
import torch

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_vocab(source="zinc"):
    """Return (char_dict, org_dict, char_weights float64 (25,)) for 'zinc' or 'pubchem'.

    Mirrors what scripts/train.py:64-77 of the reference builds from data/char_dict_*.pkl
    and data/char_weights_*.npy; the values were transcribed to JSON (no pickle needed).
    """
    with open(os.path.join(_DATA_DIR, "vocab_%s.json" % source)) as f:
        blob = json.load(f)
    char_dict = {tok: i for i, tok in enumerate(blob["tokens"])}
    org_dict = {i - 1: tok for i, tok in enumerate(blob["tokens"]) if i > 0}
    return char_dict, org_dict, np.asarray(blob["char_weights"], dtype=np.float64)


def _fans(shape):
    # torch.nn.init._calculate_fan_in_and_fan_out convention
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    return shape[1] * rf, shape[0] * rf


def synth_tensor(key, shape, seed=0):
    """Values for one state_dict entry (float32 numpy array)."""
    rng = np.random.default_rng((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF)
    shape = tuple(shape)
    if key.endswith(".a_2"):
        return (1.0 + 0.1 * rng.uniform(-1, 1, shape)).astype(np.float32)
    if key.endswith(".b_2"):
        return (0.1 * rng.uniform(-1, 1, shape)).astype(np.float32)
    if len(shape) > 1:
        fan_in, fan_out = _fans(shape)
        bound = np.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-bound, bound, shape).astype(np.float32)
    return (0.05 * rng.standard_normal(shape)).astype(np.float32)


def synth_state_dict(template, seed=0):
    """Fill a state_dict-shaped mapping {key: tensor} with synthetic values.

    ``template`` supplies keys/shapes (any model with the reference's key names).  The
    analytic positional-encoding buffers (``*.pe``) are left as they are.
    """
    out = {}
    for key, val in template.items():
        if key.endswith(".pe"):
            out[key] = val.detach().clone()
        else:
            out[key] = torch.from_numpy(synth_tensor(key, val.shape, seed)).to(val.dtype)
    return out


def synth_tokens(n, seed=0, mean_len=35.0, std_len=6.0, min_len=10, max_len=125,
                 lengths=None):
    """(n, 127) int64 rows ``[<start>, t_1..t_k, <end>, pad...]``.

    Lengths ~ clip(round(N(35, 6^2)), 10, 125) match ZINC statistics; tokens uniform on
    ids 1..23.  A row with k == 126 has neither <end> nor pad (the reference's
    encode_smiles, transvae/tvae_util.py:114-122, appends nothing when the SMILES is full).
    """
    rng = np.random.default_rng(seed)
    if lengths is None:
        lengths = np.clip(np.rint(rng.normal(mean_len, std_len, n)), min_len, max_len)
    lengths = np.asarray(lengths).astype(np.int64)
    toks = np.full((n, 127), 24, dtype=np.int64)
    toks[:, 0] = 0
    body = rng.integers(1, 24, size=(n, 126))
    for i, k in enumerate(lengths):
        toks[i, 1:1 + k] = body[i, :k]
        if k < 126:
            toks[i, 1 + k] = 25
    return torch.from_numpy(toks)


ARCHS = {
    # name -> TransVAE ctor kwargs (reference scripts/parsers.py:8-11 naming scheme)
    "trans1x-128": dict(d_model=128, d_ff=128),
    "trans4x-256": dict(d_model=256, d_ff=1024),
}
