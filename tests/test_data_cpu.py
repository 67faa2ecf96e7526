"""Data pipeline in front of the train step, CPU side.

* the oracle's restatement (oracle.tvae_oracle.tokenize / encode_tokens / data_rows) is pinned by
  tests/golden/data.npz: the reference's own vae_data_gen on its 1000-row ZINC example file and on strings
  chosen to trip a scanner (tools/make_goldens.py golden_data);
* tvae_encode_smiles (host C++ in libtvae_hip.so, no GPU needed) is bit-exact against those goldens and,
  on random strings, against the oracle -- ids and the exception class the reference raises;
* epoch_order walks a dataset in the order of the reference's DataLoader / of a DistributedSampler.
"""
import os

import numpy as np
import pytest
import torch

from oracle import tvae_oracle as O
from transvae_amd import synth
from transvae_amd.data import DeviceLoader, encode_smiles_batch, epoch_order, vae_data_gen

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def _outcome(fn, *args):
    try:
        return "ok", fn(*args)
    except Exception as e:                                  # noqa: BLE001 -- the class is what is compared
        return type(e).__name__, None


def test_oracle_rows_match_reference(g):
    cd = synth.load_vocab("zinc")[0]
    rows = O.data_rows([str(s) for s in g["zinc_smiles"]], g["zinc_props"], cd)
    assert np.array_equal(rows[:, :127], g["zinc_ids"].astype(np.float32))
    assert np.array_equal(rows[:, 127], g["zinc_prop_column"])
    for vocab in ("zinc", "pubchem"):
        cd = synth.load_vocab(vocab)[0]
        for smi, want, ids in zip(g["tricky_smiles"], g["tricky_outcome_" + vocab], g["tricky_ids_" + vocab]):
            got, rows = _outcome(O.data_rows, [str(smi)], None, cd)
            assert got == str(want), (vocab, str(smi), got, str(want))
            if got == "ok":
                assert np.array_equal(rows[0, :127], ids.astype(np.float32)), str(smi)


def test_native_encoder_matches_reference_rows(g):
    cd = synth.load_vocab("zinc")[0]
    smiles = [str(s) for s in g["zinc_smiles"]]
    ids = encode_smiles_batch(smiles, cd)
    assert ids.dtype == np.uint8 and np.array_equal(ids, g["zinc_ids"])
    data = vae_data_gen(np.array(smiles, dtype=object)[:, None], g["zinc_props"], cd)
    assert data.dtype == torch.float32 and tuple(data.shape) == (1000, 128)
    assert np.array_equal(data.numpy()[:, :127], g["zinc_ids"].astype(np.float32))
    assert np.array_equal(data.numpy()[:, 127], g["zinc_prop_column"])
    assert np.array_equal(vae_data_gen(np.array(smiles[:3], dtype=object)[:, None], None, cd).numpy()[:, 127], np.zeros(3))


@pytest.mark.parametrize("vocab", ["zinc", "pubchem"])
def test_native_encoder_on_tricky_strings(g, vocab):
    cd = synth.load_vocab(vocab)[0]
    for smi, want, ids in zip(g["tricky_smiles"], g["tricky_outcome_" + vocab], g["tricky_ids_" + vocab]):
        got, rows = _outcome(encode_smiles_batch, [str(smi)], cd)
        assert got == str(want), (str(smi), got, str(want))
        if got == "ok":
            assert np.array_equal(rows[0], ids), str(smi)


def test_native_encoder_reports_what_the_reference_reports_first(g):
    cd = synth.load_vocab("zinc")[0]
    for key in ("unknown_then_unmatched", "long_then_unknown", "unknown_and_long_in_one", "long_then_unmatched"):
        col = [str(s) for s in g["column_input_" + key]]
        got, _ = _outcome(encode_smiles_batch, col, cd)
        assert got == str(g["column_" + key]), (key, got)
    with pytest.raises(KeyError) as e:
        encode_smiles_batch(["CC", "C[N+]C"], cd)
    assert e.value.args[0] == "[N+]"
    with pytest.raises(AssertionError, match="CCX could not be joined"):
        encode_smiles_batch(["CC", "CCX"], cd)
    with pytest.raises(TypeError):
        encode_smiles_batch(["CC", float("nan")], cd)
    assert encode_smiles_batch([], cd).shape == (0, 127)


def _random_strings(rng, n, alphabet, lo, hi):
    lens = rng.integers(lo, hi, n)
    return ["".join(rng.choice(alphabet, k)) for k in lens]


def test_native_encoder_against_oracle_on_random_strings():
    rng = np.random.default_rng(5)
    cd = synth.load_vocab("pubchem")[0]
    # mostly-valid strings: pieces of the vocabulary plus a few pieces that need the scanner's care
    pieces = np.array([t for t in cd if t not in ("<start>", "<end>", "_")] + ["Br", "Cl", "[nH]", "[N+]", "[O-]", "%12"], dtype=object)
    valid = _random_strings(rng, 6000, pieces, 0, 60)
    ok_rows, ok_strings = [], []
    for s in valid:
        got, rows = _outcome(O.data_rows, [s], None, cd)
        if got == "ok":
            ok_rows.append(rows[0, :127])
            ok_strings.append(s)
        else:
            assert _outcome(encode_smiles_batch, [s], cd)[0] == got, s
    assert len(ok_strings) > 2000
    for threads in (1, 3, 8):
        ids = encode_smiles_batch(ok_strings * 3, cd, n_threads=threads)       # > 4096 rows: several threads at work
        assert np.array_equal(ids.astype(np.float32), np.stack(ok_rows * 3)), threads
    # junk: any printable byte, brackets and percent signs over-represented
    junk_alphabet = np.array(list("CcNnOoSs()[]%%[]123=#-+\\/@HBrl. XZ*$:~?>_e") + ["Cl", "Br", "[nH]"], dtype=object)
    for s in _random_strings(rng, 3000, junk_alphabet, 1, 14):
        want, rows = _outcome(O.data_rows, [s], None, cd)
        got, ids = _outcome(encode_smiles_batch, [s], cd)
        assert got == want, (s, got, want)
        if got == "ok":
            assert np.array_equal(ids[0].astype(np.float32), rows[0, :127]), s


def test_bad_row_found_from_any_thread():
    cd = synth.load_vocab("zinc")[0]
    col = ["CCO"] * 20000
    col[17321] = "CC!"
    col[9000] = "C[N+]"
    with pytest.raises(AssertionError, match="CC! could not be joined"):
        encode_smiles_batch(col, cd, n_threads=8)
    col[17321] = "CCO"
    with pytest.raises(KeyError):
        encode_smiles_batch(col, cd, n_threads=8)


def test_epoch_order_is_the_dataloaders():
    n, bs = 1037, 50
    ds = torch.arange(n)
    for seed in (0, 7):
        torch.manual_seed(seed)
        want = torch.cat(list(torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=True, drop_last=True)))
        torch.manual_seed(seed)
        got = epoch_order(n)
        assert torch.equal(got[:n // bs * bs], want)
    # two epochs in a row draw from the same default generator, as two passes over one DataLoader do
    torch.manual_seed(3)
    dl = torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=True, drop_last=True)
    want = [torch.cat(list(dl)), torch.cat(list(dl))]
    torch.manual_seed(3)
    got = [epoch_order(n), epoch_order(n)]
    assert all(torch.equal(a[:n // bs * bs], b) for a, b in zip(got, want))


@pytest.mark.parametrize("n,world", [(1037, 2), (1000, 4), (1003, 8)])
def test_epoch_order_is_the_distributed_samplers(n, world):
    ds = torch.arange(n)
    seen = []
    for rank in range(world):
        for epoch in (0, 5):
            smp = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=True, seed=0, drop_last=True)
            smp.set_epoch(epoch)
            want = torch.tensor(list(smp))
            got = epoch_order(n, world, rank, epoch, seed=0)
            assert torch.equal(got, want), (rank, epoch)
            if epoch == 0:
                seen.append(got)
    allrows = torch.cat(seen)
    assert allrows.unique().numel() == allrows.numel()            # ranks hold disjoint rows


def test_loader_length():
    class _DS:
        device = torch.device("cpu")

        def __len__(self):
            return 1037
    assert len(DeviceLoader(_DS(), 50)) == 20
    assert len(DeviceLoader(_DS(), 50, world=2, rank=1)) == 10
