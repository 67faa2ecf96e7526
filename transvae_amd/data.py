"""Data helpers with the reference's names (transvae/data.py), plus the device-resident dataset the
train loop here feeds from.

The reference tokenizes every SMILES in a Python loop into a float32 (N, 128) tensor and lets a
``torch.utils.data.DataLoader`` index and stack 500 rows per step on the host (data.py:10-33,
trans_models.py:150-157).  Here the column is encoded by the threaded C++ tokenizer of libtvae_hip.so
(one byte per token), the uint8 rows live in HBM for the whole run (1 M molecules = 127 MB) and a
batch is one gather launch that writes the float32 (B, 128) rows ``train_step`` takes.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib, ops
from .tvae_util import subsequent_mask, tokenizer

_TOK_UNMATCHED, _TOK_UNKNOWN, _TOK_TOO_LONG = 1, 2, 3
_E_DATA = -5


def _text_blob(smiles):
    "concatenated UTF-8 bytes of the strings and their (n+1) byte offsets"
    for s in smiles:
        if not isinstance(s, str):
            raise TypeError("expected string or bytes-like object, got %r" % type(s).__name__)
    joined = "".join(smiles)
    blob = joined.encode("utf-8")
    if len(blob) == len(joined):                         # pure ASCII: byte length == character count
        lens = np.fromiter((len(s) for s in smiles), dtype=np.int64, count=len(smiles))
    else:
        parts = [s.encode("utf-8") for s in smiles]
        lens = np.fromiter((len(b) for b in parts), dtype=np.int64, count=len(parts))
    offsets = np.zeros(len(smiles) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    return blob, offsets


def encode_smiles_batch(smiles, char_dict, max_len=126, n_threads=None):
    """Token-id rows for a sequence of SMILES strings: uint8 (n, max_len + 1), row =
    [0, ids..., <end>, _ ...] -- what the reference builds per molecule with tokenizer + encode_smiles and
    the leading 0 of vae_data_gen (tvae_util.py:97-122, data.py:26-31), computed by tvae_encode_smiles.
    Raises what the reference raises: AssertionError for text the tokenizer cannot split, KeyError for a
    token outside char_dict, RuntimeError for more than max_len tokens."""
    smiles = list(smiles)
    n = len(smiles)
    out = np.empty((n, max_len + 1), dtype=np.uint8)
    if n == 0:
        return out
    blob, offsets = _text_blob(smiles)
    toks = [(t.encode("utf-8"), int(i)) for t, i in char_dict.items()]
    vocab = b"".join(t for t, _ in toks)
    voff = np.zeros(len(toks) + 1, dtype=np.int32)
    np.cumsum([len(t) for t, _ in toks], out=voff[1:])
    vids = np.asarray([i for _, i in toks], dtype=np.int32)
    if n_threads is None:
        n_threads = min(16, os.cpu_count() or 1)
    bad_row, bad_code = C.c_int64(-1), C.c_int(0)
    lib = _lib.lib()
    rc = lib.tvae_encode_smiles(C.c_char_p(blob), offsets.ctypes.data, n, C.c_char_p(vocab), voff.ctypes.data,
                                vids.ctypes.data, len(toks), 0, int(char_dict['<end>']), int(char_dict['_']),
                                max_len, out.ctypes.data, out.strides[0], int(n_threads),
                                C.addressof(bad_row), C.addressof(bad_code))
    if rc == _E_DATA:
        smi = smiles[bad_row.value]
        if bad_code.value == _TOK_UNMATCHED:
            raise AssertionError("{} could not be joined".format(smi))
        if bad_code.value == _TOK_UNKNOWN:
            raise KeyError(next(t for t in tokenizer(smi) if t not in char_dict))
        raise RuntimeError("row %d: %d tokens do not fit max_len %d: %s"
                           % (bad_row.value, len(tokenizer(smi)), max_len, smi))
    if rc != 0:
        raise _lib.TvaeError("tvae_encode_smiles failed (%d): %s" % (rc, lib.tvae_last_error().decode()))
    return out


def _props_column(props, n):
    if props is None:
        return np.zeros(n, dtype=np.float32)
    return np.asarray(props, dtype=np.float64).reshape(n, -1)[:, 0].astype(np.float32)


def vae_data_gen(mols, props, char_dict):
    """SMILES column -> float32 (N, 128): [<start>, 126 token ids, property] (reference data.py:10-33)."""
    smiles = mols[:, 0]
    ids = encode_smiles_batch(smiles, char_dict, 126)
    data = np.empty((len(smiles), 128), dtype=np.float32)
    data[:, :127] = ids
    data[:, 127] = _props_column(props, len(smiles))
    return torch.from_numpy(data)


class DeviceDataset:
    """Encoded molecules resident in HBM: uint8 ids (N, 127) + float32 property (N,)."""

    def __init__(self, ids, props, device):
        ids = np.ascontiguousarray(ids, dtype=np.uint8)
        self.n, self.row_len = ids.shape
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TvaeError("DeviceDataset lives in GPU memory; got device %s" % device)
        self.ids = torch.from_numpy(ids).to(self.device)
        self.props = torch.from_numpy(_props_column(props, self.n)).to(self.device)

    @classmethod
    def from_smiles(cls, mols, props, char_dict, device):
        return cls(encode_smiles_batch(mols[:, 0], char_dict, 126), props, device)

    @classmethod
    def from_tensor(cls, data, device):
        "from the float (N, 128) tensor of vae_data_gen (ids are small integers, exact in one byte)"
        arr = data.detach().cpu().numpy()
        return cls(arr[:, :-1].astype(np.uint8), arr[:, -1], device)

    def __len__(self):
        return self.n

    def batch(self, index=None, first=0, size=None):
        "float32 (B, row_len + 1) rows for `index` (int64 device tensor) or for rows first .. first+size"
        if index is not None:
            if not (index.is_cuda and index.dtype == torch.int64 and index.is_contiguous()):
                raise _lib.TvaeError("DeviceDataset.batch: index must be a contiguous int64 CUDA tensor")
            size = index.numel()
        out = torch.empty((size, self.row_len + 1), device=self.device, dtype=torch.float32)
        _lib.call("tvae_gather_batch", ops._p(self.ids), self.row_len, ops._p(self.props),
                  ops._p(index) if index is not None else None, int(first), self.n, int(size), self.row_len,
                  ops._p(out), self.row_len + 1, ops._stream())
        return out


def epoch_order(n, world=1, rank=0, epoch=0, seed=0):
    """The row order one rank walks in one epoch, as the reference's loaders produce it.

    world == 1: ``DataLoader(shuffle=True)`` -- its iterator draws a base seed, then RandomSampler draws its
    own seed, both from torch's default generator, and permutes with a generator seeded by the second
    (torch/utils/data/dataloader.py, sampler.py), so under the same ``torch.manual_seed`` the order is the
    DataLoader's.  world > 1: ``DistributedSampler(shuffle=True, drop_last=True, seed)``: a permutation
    seeded by seed + epoch, cut to a multiple of world and dealt round-robin."""
    g = torch.Generator()
    if world == 1:
        torch.empty((), dtype=torch.int64).random_()                     # _BaseDataLoaderIter._base_seed
        g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        return torch.randperm(n, generator=g)
    g.manual_seed(seed + epoch)
    perm = torch.randperm(n, generator=g)
    per_rank = math.ceil((n - world) / world) if n % world != 0 else n // world
    return perm[:per_rank * world][rank::world].contiguous()


class DeviceLoader:
    """Iterates float32 (batch_size, 128) device batches over a DeviceDataset in the order of
    ``DataLoader(dataset, batch_size, shuffle=True, drop_last=True)`` (with a DistributedSampler when
    world > 1).  One host->device copy of the permutation per epoch; every batch is one gather launch."""

    def __init__(self, dataset, batch_size, world=1, rank=0, seed=0):
        self.dataset, self.batch_size = dataset, int(batch_size)
        self.world, self.rank, self.seed, self.epoch = world, rank, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        n = len(self.dataset)
        if self.world > 1:
            n = math.ceil((n - self.world) / self.world) if n % self.world != 0 else n // self.world
        return n // self.batch_size

    def __iter__(self):
        order = epoch_order(len(self.dataset), self.world, self.rank, self.epoch, self.seed)
        order = order.to(self.dataset.device)
        for j in range(order.numel() // self.batch_size):
            yield self.dataset.batch(order[j * self.batch_size:(j + 1) * self.batch_size])


_causal = {}


def _causal_on(size, device):
    "subsequent_mask(size) resident on `device` (built once: torch.tril on the host + the copy cost 0.8 ms of every step)"
    key = (int(size), device.type, device.index)
    m = _causal.get(key)
    if m is None:
        if len(_causal) > 64:
            _causal.clear()
        m = _causal[key] = subsequent_mask(size).to(device)
    return m


def make_std_mask(tgt, pad):
    """(B, L, L) bool: key is not pad AND key <= query (reference data.py:35-48).  The returned
    tensor also carries the kernel form (pad vector + causal flag), so the attention kernel never
    reads the L x L mask."""
    key_ok = (tgt != pad)
    mask = key_ok.unsqueeze(-2) & _causal_on(tgt.size(-1), tgt.device)
    return ops.tag_mask(mask, ops.MaskSpec(key_mask=key_ok.to(torch.uint8).contiguous(), causal=True))


def make_src_mask(src, pad):
    """(B, 1, L) bool source mask as built inline by the reference (trans_models.py:214), tagged
    with its kernel form."""
    key_ok = (src != pad)
    return ops.tag_mask(key_ok.unsqueeze(-2), ops.MaskSpec(key_mask=key_ok.to(torch.uint8).contiguous()))
