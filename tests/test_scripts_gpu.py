"""End-to-end through the command-line entry points on the GPU: train (custom data source) ->
checkpoint -> sample -> attention dump, the way a user of the reference's scripts/ would run them."""
import os
import sys

import numpy as np
import pandas as pd
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

BASE = ["CCO", "c1ccccc1", "CC(=O)O", "CCN(CC)CC", "C1CCCCC1", "CC(C)Cc1ccc(cc1)C(C)C(=O)O", "O=C(O)c1ccccc1O",
        "CCOC(=O)C", "c1ccncc1", "CC(C)O", "ClCCl", "FC(F)F", "CSC", "NC(=O)C", "C#N", "Brc1ccccc1"]


def test_train_sample_attention_scripts(tmp_path, monkeypatch):
    from scripts import attention, parsers, sample, train
    monkeypatch.chdir(tmp_path)
    mols = [BASE[i % len(BASE)] + "C" * (i // len(BASE)) for i in range(48)]
    pd.DataFrame(mols, columns=["mol"]).to_csv("train.txt", index=False)
    pd.DataFrame(mols[:24], columns=["mol"]).to_csv("test.txt", index=False)
    vocab = os.path.join(ROOT, "transvae_amd", "data", "vocab_zinc.json")
    args = parsers.train_parser().parse_args(
        ["--model", "transvae", "--data_source", "custom", "--train_mols_path", "train.txt", "--test_mols_path",
         "test.txt", "--vocab_path", vocab, "--batch_size", "12", "--batch_chunks", "2", "--epochs", "2",
         "--save_freq", "1", "--save_name", "tiny", "--d_model", "128", "--d_feedforward", "128"])
    train.train(args)
    assert os.path.exists("checkpoints/001_tiny.ckpt") and os.path.exists("checkpoints/002_tiny.ckpt")
    log = pd.read_csv("trials/log_tiny.txt")
    assert set(log["data_type"]) == {"train", "test"} and len(log) == 2 * (4 + 2)
    assert np.isfinite(log["tot_loss"]).all()

    # resume for one more epoch: weights, optimizer moments and the epoch counter carry over
    args2 = parsers.train_parser().parse_args(
        ["--model", "transvae", "--data_source", "custom", "--train_mols_path", "train.txt", "--test_mols_path",
         "test.txt", "--vocab_path", vocab, "--batch_size", "12", "--batch_chunks", "2", "--epochs", "1",
         "--save_freq", "1", "--save_name", "tiny", "--checkpoint", "checkpoints/002_tiny.ckpt"])
    train.train(args2)
    assert os.path.exists("checkpoints/003_tiny.ckpt")

    sargs = parsers.sample_parser().parse_args(
        ["--model", "transvae", "--model_ckpt", "checkpoints/003_tiny.ckpt", "--sample_mode", "rand",
         "--n_samples", "10", "--n_samples_per_batch", "5", "--save_path", "gen.csv"])
    sample.sample(sargs)
    assert len(pd.read_csv("gen.csv", keep_default_na=False)) == 10

    aargs = parsers.attn_parser().parse_args(
        ["--model", "transvae", "--model_ckpt", "checkpoints/003_tiny.ckpt", "--mols", "train.txt",
         "--n_samples", "12", "--batch_size", "6", "--batch_chunks", "2", "--save_path", "attn"])
    attention.calc_attention(aargs)
    sa, xa = np.load("attn_self_attn.npy"), np.load("attn_src_attn.npy")
    assert sa.shape == (12, 4, 4, 127, 127) and xa.shape == (12, 3, 4, 126, 127)
    np.testing.assert_allclose(sa.sum(-1), 1.0, atol=1e-4)          # every dumped row is a softmax
    np.testing.assert_allclose(xa.sum(-1), 1.0, atol=1e-4)
