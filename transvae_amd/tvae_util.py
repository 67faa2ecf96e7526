"""Host-side helpers with the reference's names (transvae/tvae_util.py), re-implemented.

Only what the train / sample / attention paths touch: module cloning, masks, the KL annealer,
the SMILES tokenizer + encoder, decode_mols and calc_entropy.  RDKit metrics are out of scope
(SURVEY.md section 2, rows 7 and 10).
"""
import copy
import re

import numpy as np
import torch
import torch.nn as nn

from . import ops


def clones(module, N):
    """N independent deep copies in a ModuleList (reference tvae_util.py:27-30)."""
    out = nn.ModuleList([copy.deepcopy(module) for _ in range(N)])
    for m in out.modules():          # every copy gets dropout sites of its own
        if hasattr(m, "_new_sites"):
            m._new_sites()
    return out


def subsequent_mask(size):
    """(1, size, size) bool, True where key <= query (reference tvae_util.py:32-37).
    The tensor carries its kernel form (causal flag) so attention needs no mask read."""
    m = torch.tril(torch.ones((1, size, size), dtype=torch.bool))
    return ops.tag_mask(m, ops.MaskSpec(causal=True))


class ListModule(nn.Module):
    """Modules registered under '0', '1', ... (reference tvae_util.py:50-71); keeps the
    state_dict key scheme conv_layers.{i}.0.* of the bottlenecks."""

    def __init__(self, *args):
        super().__init__()
        for i, m in enumerate(args):
            self.add_module(str(i), m)

    def __getitem__(self, idx):
        if idx < 0 or idx >= len(self._modules):
            raise IndexError('index {} is out of range'.format(idx))
        return list(self._modules.values())[idx]

    def __iter__(self):
        return iter(self._modules.values())

    def __len__(self):
        return len(self._modules)


class KLAnnealer:
    """Linear beta schedule per epoch, clipped at kl_high (reference tvae_util.py:73-92)."""

    def __init__(self, kl_low, kl_high, n_epochs, start_epoch):
        self.kl_low, self.kl_high = kl_low, kl_high
        self.n_epochs, self.start_epoch = n_epochs, start_epoch
        self.kl = (kl_high - kl_low) / (n_epochs - start_epoch)

    def __call__(self, epoch):
        k = (epoch - self.start_epoch) if epoch >= self.start_epoch else 0
        return min(self.kl_low + k * self.kl, self.kl_high)


_SMILES_RE = re.compile(
    r"(\[[^\]]+]|Br?|Cl?|N|O|S|P|F|I|b|c|n|o|s|p|\(|\)|\.|=|#|-|\+|\\|\/|_|:|~|@|\?|>|\*|\$|\%[0-9]{2}|[0-9])")


def tokenizer(smile):
    """Regex SMILES tokenizer (reference tvae_util.py:97-103)."""
    tokens = _SMILES_RE.findall(smile)
    assert smile == ''.join(tokens), ("{} could not be joined".format(smile))
    return tokens


def build_org_dict(char_dict):
    """id-1 -> token for every token but the first (reference tvae_util.py:105-112)."""
    return {int(v - 1): k for i, (k, v) in enumerate(char_dict.items()) if i > 0}


def encode_smiles(smile, max_len, char_dict):
    """Token list -> ids, '<end>' then '_' padding up to max_len (reference tvae_util.py:114-122)."""
    pad = max_len - len(smile)
    if pad > 0:
        smile = smile + ['<end>'] + ['_'] * (pad - 1)
    return [char_dict[c] for c in smile]


def decode_mols(encoded_tensors, org_dict):
    """ids (n, L) -> strings: tokens until '<end>', pads stripped (reference tvae_util.py:152-166).
    One device->host copy for the whole batch instead of one per row; the cut at '<end>' is found for all rows at once
    and a row is one table lookup + join (a million rows of uint8 ids take seconds, not minutes)."""
    arr = encoded_tensors.detach().cpu().numpy().astype(np.int16) - 1
    if arr.size == 0:
        return ['' for _ in range(arr.shape[0])]
    n_tok = max(org_dict.keys()) + 1
    if arr.min() < 0 or arr.max() >= n_tok:
        raise KeyError(int(arr.min() if arr.min() < 0 else arr.max()))      # what org_dict[idx] raises in the reference
    lut = np.array([org_dict[i] for i in range(n_tok)], dtype=object)       # KeyError on a gap, as the reference
    is_end = arr == [i for i in range(n_tok) if org_dict[i] == '<end>'][0]
    ends = np.where(is_end.any(1), is_end.argmax(1), arr.shape[1])
    return [''.join(lut[row[:e]]).strip('_') for row, e in zip(arr, ends)]


def calc_entropy(sample):
    """Shannon entropy of a 1000-bin histogram per latent dimension (reference tvae_util.py:198-204)."""
    from scipy.stats import entropy
    return np.array([entropy(np.histogram(sample[:, i], bins=1000, range=(-5., 5.), density=True)[0])
                     for i in range(sample.shape[1])])
