"""scripts/: flag parity with the reference's parsers (names + defaults transcribed from
/root/reference/scripts/parsers.py:36-122) and the plain-data vocabulary loader."""
import os
import pickle
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts import parsers  # noqa: E402

TRAIN_DEFAULTS = dict(d_model=128, d_feedforward=128, d_latent=128, property_predictor=False, d_property_predictor=256,
                      depth_property_predictor=2, batch_size=500, batch_chunks=5, beta=0.05, beta_init=1e-8,
                      anneal_start=0, adam_lr=3e-4, lr_scale=1, warmup_steps=10000, eps_scale=1, epochs=100,
                      train_mols_path=None, test_mols_path=None, train_props_path=None, test_props_path=None,
                      vocab_path=None, char_weights_path=None, checkpoint=None, save_name=None, save_freq=5)
SAMPLE_DEFAULTS = dict(mols=None, k=15, condition='', entropy_cutoff=5, n_samples=30000, n_samples_per_batch=100,
                       save_path=None)
ATTN_DEFAULTS = dict(n_samples=5000, batch_size=500, batch_chunks=5, shuffle=False, save_path=None)


def test_train_flags():
    a = parsers.train_parser().parse_args(['--model', 'transvae', '--data_source', 'zinc'])
    for k, v in TRAIN_DEFAULTS.items():
        assert getattr(a, k) == v, k
    assert parsers.default_save_name(a) == 'trans1x-128_zinc'
    a = parsers.train_parser().parse_args(['--model', 'transvae', '--data_source', 'pubchem', '--d_model', '256',
                                           '--d_feedforward', '1024'])
    assert parsers.default_save_name(a) == 'trans4x-256_pubchem'
    with pytest.raises(SystemExit):
        parsers.train_parser().parse_args(['--model', 'transvae'])                 # --data_source is required


def test_sample_and_attn_flags():
    a = parsers.sample_parser().parse_args(['--model', 'transvae', '--model_ckpt', 'c', '--sample_mode', 'rand'])
    for k, v in SAMPLE_DEFAULTS.items():
        assert getattr(a, k) == v, k
    with pytest.raises(SystemExit):
        parsers.sample_parser().parse_args(['--model', 'transvae', '--model_ckpt', 'c', '--sample_mode', 'bogus'])
    a = parsers.attn_parser().parse_args(['--model', 'transvae', '--model_ckpt', 'c', '--mols', 'm'])
    for k, v in ATTN_DEFAULTS.items():
        assert getattr(a, k) == v, k


def test_default_save_names_of_the_three_model_families():
    "reference scripts/parsers.py:7-17: trans{d_ff/d_model}x-{d_model}_{source} | {model}-{d_model}_{source} | --save_name"
    a = parsers.train_parser().parse_args(['--model', 'transvae', '--d_model', '256', '--d_feedforward', '1024', '--data_source', 'zinc'])
    assert parsers.default_save_name(a) == 'trans4x-256_zinc'
    a = parsers.train_parser().parse_args(['--model', 'rnnattn', '--d_model', '256', '--data_source', 'pubchem'])
    assert parsers.default_save_name(a) == 'rnnattn-256_pubchem'
    a = parsers.train_parser().parse_args(['--model', 'rnn', '--data_source', 'zinc', '--save_name', 'mine'])
    assert parsers.default_save_name(a) == 'mine'


def test_vocab_loader_reads_plain_pickles_only(tmp_path):
    good = tmp_path / 'v.pkl'
    good.write_bytes(pickle.dumps({'<start>': 0, 'C': 1, '_': 2}))
    assert parsers.load_char_dict(str(good)) == {'<start>': 0, 'C': 1, '_': 2}
    bad = tmp_path / 'b.pkl'
    bad.write_bytes(pickle.dumps({'x': os.getcwd}))                                 # needs a global: refused
    with pytest.raises(pickle.UnpicklingError):
        parsers.load_char_dict(str(bad))
    a = parsers.train_parser().parse_args(['--model', 'transvae', '--data_source', 'zinc'])
    cd, od, w = parsers.load_vocab(a)
    assert cd['<start>'] == 0 and len(w) == len(cd) - 1 and od[0] != '<start>'


def test_bench_and_entry_points_are_self_contained():
    """bench.py / __graft_entry__.py run on a GPU box where /root/reference does not exist: neither may name it,
    and both keep the command-line contract (--gpus / --steps / --warmup)."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("bench.py", "__graft_entry__.py"):
        src = open(os.path.join(root, name)).read()
        ast.parse(src)
        assert "/root/reference" not in src, name
    src = open(os.path.join(root, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src
