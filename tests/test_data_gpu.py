"""Data pipeline on the GPU: the gather launch against plain indexing of the float
tensor the reference feeds its DataLoader, the loader against the DataLoader itself, and VAEShell.train's log
against a step-by-step replay of the same batches."""
import os
import sys

import numpy as np
import pandas as pd
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

from transvae_amd import synth  # noqa: E402
from transvae_amd.data import DeviceDataset, DeviceLoader, vae_data_gen  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data.npz")


def _zinc():
    g = np.load(GOLD)
    smiles = np.array([str(s) for s in g["zinc_smiles"]], dtype=object)[:, None]
    return g, smiles, synth.load_vocab("zinc")[0]


def test_gather_batch_is_the_rows_of_the_reference_tensor():
    g, smiles, cd = _zinc()
    want = torch.from_numpy(np.concatenate([g["zinc_ids"].astype(np.float32), g["zinc_prop_column"][:, None]], 1))
    ds = DeviceDataset.from_smiles(smiles, g["zinc_props"], cd, "cuda:0")
    assert len(ds) == 1000
    idx = torch.from_numpy(np.random.default_rng(0).permutation(1000)[:500].copy()).cuda()
    assert torch.equal(ds.batch(idx).cpu(), want[idx.cpu()])                       # bit-exact: small integers
    assert torch.equal(ds.batch(first=990, size=10).cpu(), want[990:])
    assert torch.equal(ds.batch(idx[:1]).cpu(), want[idx[:1].cpu()])
    # the float tensor of vae_data_gen round-trips through one byte per token
    ds2 = DeviceDataset.from_tensor(vae_data_gen(smiles, g["zinc_props"], cd), "cuda:0")
    assert torch.equal(ds2.batch(first=0, size=1000).cpu(), want)
    # an index outside the dataset must not become a wild read
    bad = torch.tensor([3, 1000, -1, 999], device="cuda")
    out = ds.batch(bad).cpu()
    assert torch.equal(out[[0, 3]], want[[3, 999]]) and float(out[[1, 2]].abs().sum()) == 0.0
    from transvae_amd._lib import TvaeError
    with pytest.raises(TvaeError):
        ds.batch(first=995, size=10)
    with pytest.raises(TvaeError):
        ds.batch(idx.cpu())
    with pytest.raises(TvaeError):
        DeviceDataset(g["zinc_ids"], None, "cpu")


def test_loader_yields_the_dataloaders_batches():
    g, smiles, cd = _zinc()
    data = vae_data_gen(smiles, g["zinc_props"], cd)
    torch.manual_seed(11)
    want = list(torch.utils.data.DataLoader(data, batch_size=64, shuffle=True, drop_last=True))
    torch.manual_seed(11)
    got = list(DeviceLoader(DeviceDataset.from_tensor(data, "cuda:0"), 64))
    assert len(got) == len(want) == 15
    assert all(torch.equal(a.cpu(), b) for a, b in zip(got, want))
    # world 2: the two ranks' batches are the DistributedSampler's
    for rank in (0, 1):
        smp = torch.utils.data.distributed.DistributedSampler(data, 2, rank, shuffle=True, drop_last=True)
        smp.set_epoch(3)
        want = list(torch.utils.data.DataLoader(data, batch_size=64, sampler=smp, drop_last=True))
        ld = DeviceLoader(DeviceDataset.from_tensor(data, "cuda:0"), 64, world=2, rank=rank)
        ld.set_epoch(3)
        got = list(ld)
        assert len(got) == len(want) == len(ld) == 7
        assert all(torch.equal(a.cpu(), b) for a, b in zip(got, want))


def test_train_log_equals_a_replay_of_the_same_batches(tmp_path, monkeypatch):
    """VAEShell.train (device loader, loss tables read back two steps late) writes the losses that calling
    train_step / eval_step batch by batch on the DataLoader's batches gives: same order, same numbers."""
    from transvae_amd.trans_models import TransVAE
    monkeypatch.chdir(tmp_path)
    g, smiles, cd = _zinc()
    cdict, org, w = synth.load_vocab("zinc")

    def build():
        torch.manual_seed(0)
        vae = TransVAE(params={"CHAR_DICT": cdict, "ORG_DICT": org, "CHAR_WEIGHTS": w, "BATCH_SIZE": 40,
                               "BATCH_CHUNKS": 2, "BETA_INIT": 1e-3, "BETA": 0.05, "ANNEAL_START": 0},
                       name="t", d_model=128, d_ff=128, dropout=0.0)
        vae.model.load_state_dict(synth.synth_state_dict(vae.model.state_dict(), 3))
        return vae
    train_mols, val_mols = smiles[:200], smiles[200:290]
    from transvae_amd import ops
    vae = build()
    torch.manual_seed(5)
    ops.seed_rng(torch.device("cuda:0"), 77)      # attention dropout is 0.1 whatever `dropout` says (reference quirk)
    vae.train(train_mols, val_mols, epochs=2, save=False, log=True, log_dir="trials")
    log = pd.read_csv("trials/log_t.txt")
    assert list(log["data_type"]) == (["train"] * 5 + ["test"] * 2) * 2
    assert list(log["batch_idx"]) == ([0, 1, 2, 3, 4] + [0, 1]) * 2
    assert (log["run_time"][log["data_type"] == "train"] > 0).all() and log["run_time"][log["data_type"] == "test"].isna().all()

    ref = build()
    for ma, mb in zip(vae.model.modules(), ref.model.modules()):     # same dropout site ids -> same keep masks
        if hasattr(ma, "_sites"):
            mb._sites = list(ma._sites)
    torch.manual_seed(5)
    ops.seed_rng(torch.device("cuda:0"), 77)
    from transvae_amd.tvae_util import KLAnnealer
    ann = KLAnnealer(1e-3, 0.05, 2, 0)
    tr = torch.utils.data.DataLoader(vae_data_gen(train_mols, None, cdict), batch_size=40, shuffle=True, drop_last=True)
    va = torch.utils.data.DataLoader(vae_data_gen(val_mols, None, cdict), batch_size=40, shuffle=True, drop_last=True)
    rows = []
    for epoch in range(2):
        ref.model.train()
        for data in tr:
            rows.append(ref.train_step(data, ann(epoch)).cpu().numpy().mean(0))
        ref.model.eval()
        for data in va:
            rows.append(ref.eval_step(data, ann(epoch)).cpu().numpy().mean(0))
    want = np.stack(rows)
    got = log[["tot_loss", "recon_loss", "pred_loss", "kld_loss", "prop_mse_loss"]].to_numpy()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-7)
