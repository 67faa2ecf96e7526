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


REF_CKPT = os.path.join(ROOT, "tests", "golden", "ref_tiny.ckpt")


def test_checkpoint_written_by_the_reference_loads_with_the_safe_loader():
    """tests/golden/ref_tiny.ckpt was written by the REFERENCE's own VAEShell.save after one NoamOpt.step
    (tools/make_goldens.py golden_ckpt; reference trans_models.py:69-92).  TransVAE(load_fn=...) must read it with the
    safe unpickler (np.float64 best_loss and all), rebuild the architecture from the stored params, restore every
    tensor bit for bit and carry the Noam step on (reference load, :94-124, opt.py:38-39).  The greedy tokens that
    state decodes to are checked on the GPU (tests/test_model_gpu.py)."""
    from transvae_amd.trans_models import TransVAE
    g = load("ref_ckpt")
    assert os.environ.get("TVAE_TRUSTED_CKPT") != "1"            # the SAFE loader is what is being tested
    vae = TransVAE(load_fn=REF_CKPT)
    assert vae.name == "tiny" and vae.n_epochs == int(g["epoch"]) and float(vae.best_loss) == float(g["best_loss"])
    assert [vae.params[k] for k in ("N", "d_model", "d_ff", "d_latent", "h")] == [1, 64, 64, 32, 2]
    assert vae.params["WARMUP_STEPS"] == 10000          # not an arch param: the constructor's default wins (reference :115-119)
    sd = vae.model.state_dict()
    assert list(sd) == [str(k) for k in g["keys"]]
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g["sums"], rtol=0, atol=0)
    st = vae.optimizer.state_dict
    assert st["step"] == int(g["noam_step"]) == 1 and st["rate"] == float(g["noam_rate"])
    assert len(st["state"]) == int(g["opt_state_entries"])
    # the next step continues the schedule from the stored step
    assert abs(vae.optimizer.rate(st["step"] + 1) - 64 ** -0.5 * 2 * 10000 ** -1.5) < 1e-18


def test_checkpoint_written_by_the_build_was_loaded_by_the_reference():
    """The other direction, recorded in the container by tools/make_goldens.py (the reference cannot run on the GPU
    box): a checkpoint written by this package's VAEShell.save went through the reference's VAEShell.load; it saw
    the same state_dict keys in the same order and every tensor bit for bit."""
    from transvae_amd import synth
    from transvae_amd.trans_models import TransVAE
    g = load("ref_ckpt")
    assert bool(g["reverse_loaded"]) and bool(g["reverse_keys_equal"]) and bool(g["reverse_tensors_equal"])
    assert int(g["reverse_epoch"]) == 7 and g["reverse_arch"].tolist() == [1, 64, 64, 32, 2]
    cd, od, w = synth.load_vocab("zinc")
    mine = TransVAE(params={"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w}, name="mine", N=1, d_model=64, d_ff=64,
                    d_latent=32, h=2)
    assert [str(k) for k in g["reverse_keys"]] == list(mine.model.state_dict())


def test_sample_from_memory_matches_the_reference_draw_for_draw():
    """VAEShell.sample_from_memory is host code over torch's and numpy's global generators (reference
    trans_models.py:355-381): under the same seeds the three modes must give the reference's z bit for bit."""
    from transvae_amd import synth
    from transvae_amd.trans_models import TransVAE
    g = load("sampling")
    cd, od, w = synth.load_vocab("zinc")
    vae = TransVAE(params={"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w}, name="s", **synth.ARCHS["trans1x-128"])
    torch.manual_seed(int(g["torch_seed"]))
    np.random.seed(int(g["numpy_seed"]))
    dims, pool = g["dims"].tolist(), g["pool"].tolist()
    assert np.array_equal(vae.sample_from_memory(7, mode="rand").numpy(), g["rand"])
    assert np.array_equal(vae.sample_from_memory(7, mode="top_dims", sample_dims=dims).numpy(), g["top_dims"])
    assert np.array_equal(vae.sample_from_memory(7, mode="k_dims", sample_dims=pool, k=3).numpy(), g["k_dims"])
    assert np.array_equal(vae.sample_from_memory(5, mode="k_dims", sample_dims=pool, k=5).numpy(), g["k_dims_again"])
    with pytest.raises(AssertionError):
        vae.sample_from_memory(3, mode="top_dims")


def test_pack_qkv_keeps_names_values_and_adjacency():
    """MultiHeadedAttention.pack_qkv lays q, k, v (weights, biases, gradients) out as row blocks of one matrix: parameter
    names, shapes and values are untouched, ops._adjacent sees the three as one run, load_state_dict (an in-place copy) keeps it;
    deepcopy clones every Parameter on its own, which un-packs the copy (ops then runs the three GEMMs of before), and a
    tensor that is merely equal but allocated elsewhere is not taken for adjacent."""
    import copy
    import torch
    from transvae_amd import ops
    from transvae_amd.trans_models import MultiHeadedAttention
    torch.manual_seed(0)
    m = MultiHeadedAttention(4, 64)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    lin = m.linears
    assert not ops._adjacent(lin[0].weight, lin[1].weight, lin[2].weight)
    m.pack_qkv()
    after = m.state_dict()
    assert list(after) == list(before) and all(torch.equal(after[k], before[k]) for k in before)
    assert ops._adjacent(lin[0].weight, lin[1].weight, lin[2].weight) and ops._adjacent(lin[0].bias, lin[1].bias, lin[2].bias)
    assert ops._adjacent(lin[0].weight.grad, lin[1].weight.grad, lin[2].weight.grad)
    assert not ops._adjacent(lin[2].weight, lin[3].weight)              # the output projection is not part of the run
    run = ops._run([lin[0].weight, lin[1].weight, lin[2].weight])
    assert run.shape == (192, 64) and torch.equal(run, torch.cat([before["linears.%d.weight" % i] for i in range(3)], 0))
    m.load_state_dict({k: v + 1 for k, v in before.items()})
    assert ops._adjacent(lin[0].weight, lin[1].weight, lin[2].weight) and torch.equal(run[64:128], before["linears.1.weight"] + 1)
    m2 = copy.deepcopy(m)
    l2 = m2.linears
    assert not ops._adjacent(l2[0].weight, l2[1].weight, l2[2].weight)
    m2.pack_qkv()
    assert ops._adjacent(l2[0].weight, l2[1].weight, l2[2].weight) and torch.equal(l2[1].weight, lin[1].weight)
    assert not ops._adjacent(lin[0].weight, lin[1].weight.detach().clone())
