"""Data helpers with the reference's names (transvae/data.py)."""
import numpy as np
import torch

from . import ops
from .tvae_util import encode_smiles, subsequent_mask, tokenizer


def vae_data_gen(mols, props, char_dict):
    """SMILES column -> float32 (N, 128): [<start>, 126 token ids, property] (reference data.py:10-33)."""
    smiles = mols[:, 0]
    if props is None:
        props = np.zeros(smiles.shape)
    ids = np.zeros((len(smiles), 128), dtype=np.float32)
    for j, smi in enumerate(smiles):
        ids[j, 1:127] = encode_smiles(tokenizer(smi), 126, char_dict)
        ids[j, 127] = float(np.asarray(props[j]).reshape(-1)[0])
    return torch.from_numpy(ids)


def make_std_mask(tgt, pad):
    """(B, L, L) bool: key is not pad AND key <= query (reference data.py:35-48).  The returned
    tensor also carries the kernel form (pad vector + causal flag), so the attention kernel never
    reads the L x L mask."""
    key_ok = (tgt != pad)
    mask = key_ok.unsqueeze(-2) & subsequent_mask(tgt.size(-1)).to(tgt.device)
    return ops.tag_mask(mask, ops.MaskSpec(key_mask=key_ok.to(torch.uint8).contiguous(), causal=True))


def make_src_mask(src, pad):
    """(B, 1, L) bool source mask as built inline by the reference (trans_models.py:214), tagged
    with its kernel form."""
    key_ok = (src != pad)
    return ops.tag_mask(key_ok.unsqueeze(-2), ops.MaskSpec(key_mask=key_ok.to(torch.uint8).contiguous()))
