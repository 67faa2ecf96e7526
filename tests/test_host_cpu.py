"""CPU-side checks: the C-ABI library loads and exports every declared symbol; host helpers
(tokenizer, encode, masks, schedules, checkpoint naming) agree with golden data from the reference."""
import os
import re

import numpy as np
import pytest
import torch

from helpers import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from transvae_amd import _lib
    header = open(os.path.join(ROOT, "include", "tvae_kernels.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(tvae_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    handle = _lib.lib()                      # loads without a GPU; no compute call is made
    for name in declared:
        assert hasattr(handle, name), name
    assert declared - {"tvae_last_error", "tvae_version"} == set(_lib.SIGNATURES), \
        (declared ^ set(_lib.SIGNATURES))
    assert b"gfx950" in handle.tvae_version()


def test_host_tensor_is_refused_not_faulted():
    from transvae_amd import ops, _lib
    with pytest.raises(_lib.TvaeError):
        ops._p(torch.zeros(4))


def test_tokenizer_and_encoding_match_reference():
    from transvae_amd import synth
    from transvae_amd.data import vae_data_gen
    from transvae_amd.tvae_util import build_org_dict, decode_mols, encode_smiles, tokenizer
    g = load("units")
    cd, od, _ = synth.load_vocab("zinc")
    assert build_org_dict(cd) == od
    smiles = [str(s) for s in g["smiles"]]
    enc = np.stack([[0] + encode_smiles(tokenizer(s), 126, cd) for s in smiles])
    assert np.array_equal(enc, g["smiles_ids"].astype(np.int64))
    data = vae_data_gen(np.array(smiles)[:, None], None, cd)
    assert data.shape == (len(smiles), 128) and data.dtype == torch.float32
    assert np.array_equal(data[:, :127].numpy().astype(np.int64), enc) and float(data[:, 127].abs().max()) == 0
    ids = torch.from_numpy(g["decode_ids"].astype(np.int64))
    assert decode_mols(ids, od) == [str(s) for s in g["decode_strs"]]
    # round trip: ids of a real SMILES decode back to the SMILES
    assert decode_mols(torch.from_numpy(enc[:, 1:126]), od) == smiles
    with pytest.raises(AssertionError):
        tokenizer("C%")


def test_schedules_and_masks():
    from transvae_amd.data import make_src_mask, make_std_mask
    from transvae_amd.opt import NoamOpt
    from transvae_amd.tvae_util import KLAnnealer, calc_entropy, subsequent_mask
    g = load("units")
    ann = KLAnnealer(1e-8, 0.05, 100, 0)
    np.testing.assert_allclose([ann(e) for e in range(0, 120, 7)], g["anneal_0_100"], rtol=1e-14)
    ann = KLAnnealer(0.001, 0.05, 60, 10)
    np.testing.assert_allclose([ann(e) for e in range(0, 80, 5)], g["anneal_10_60"], rtol=1e-14)
    opt = NoamOpt(256, 1, 10000, torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0))
    np.testing.assert_allclose([opt.rate(int(s)) for s in g["noam_steps"]], g["noam_256"], rtol=1e-14)
    np.testing.assert_allclose(calc_entropy(g["entropy_sample"]), g["entropy"], rtol=1e-12)
    m = subsequent_mask(5)
    assert m.shape == (1, 5, 5) and m.dtype == torch.bool and bool(m[0, 3, 3]) and not bool(m[0, 2, 3])
    toks = torch.tensor([[0, 3, 4, 25, 24, 24], [0, 5, 25, 24, 24, 24]])
    tm = make_std_mask(toks[:, :-1], 24)
    ref = (toks[:, :-1] != 24).unsqueeze(-2) & torch.tril(torch.ones(1, 5, 5, dtype=torch.bool))
    assert torch.equal(tm, ref) and tm._tvae_spec.causal
    sm = make_src_mask(toks, 24)
    assert sm.shape == (2, 1, 6) and torch.equal(sm[:, 0], toks != 24)


def test_shell_defaults_and_state_dict_keys(tmp_path):
    from helpers import state_shapes
    from transvae_amd import synth
    from transvae_amd.trans_models import TransVAE
    cd, od, w = synth.load_vocab("pubchem")
    vae = TransVAE(params={"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w, "LR_SCALE": 7}, name="x",
                   d_model=256, d_ff=1024, property_predictor=True)
    assert vae.params["LR_SCALE"] == 1                       # reference quirk: always reset (trans_models.py:38-39)
    assert (vae.params["BATCH_SIZE"], vae.params["BATCH_CHUNKS"], vae.params["WARMUP_STEPS"]) == (500, 5, 10000)
    assert vae.src_len == 126 and vae.tgt_len == 125 and vae.pad_idx == 24 and vae.vocab_size == 26
    sd = vae.model.state_dict()
    shapes = state_shapes(256, 1024, pp=True)
    assert {k for k in sd if not k.endswith(".pe")} == set(shapes)
    for k, shp in shapes.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    assert tuple(sd["src_embed.1.pe"].shape) == (1, 5000, 256)
    assert sum(p.numel() for p in vae.model.parameters()) == 7876205 + 128 * 256 + 256 + 256 + 1
    # checkpoint naming rules of VAEShell.save (reference :78-91)
    vae.current_state["model_state_dict"] = sd
    vae.save(vae.current_state, "best", path=str(tmp_path))
    vae.save(vae.current_state, "007.ckpt", path=str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ["007_x.ckpt", "best_x.ckpt"]
