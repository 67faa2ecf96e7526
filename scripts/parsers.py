"""Command-line flags of the TransVAE scripts (same names and defaults as the reference's
scripts/parsers.py:36-122, so existing job files keep working) and the small loaders the scripts share.
All three model families are built: `--model transvae | rnnattn | rnn` (the attention dump: transvae | rnnattn)."""
import argparse
import io
import json
import pickle

import numpy as np

from transvae_amd import synth
from transvae_amd.rnn_models import RNN, RNNAttn
from transvae_amd.trans_models import TransVAE
from transvae_amd.tvae_util import build_org_dict

MODELS = ['transvae', 'rnnattn', 'rnn']


def _add(parser, spec):
    for flag, kw in spec:
        parser.add_argument(flag, **kw)
    return parser


def _opt(default, typ):
    return dict(default=default, type=typ)


_SWITCH = dict(default=False, action='store_true')


def train_parser():
    return _add(argparse.ArgumentParser(), [
        ('--model', dict(choices=MODELS, required=True, type=str)),
        ('--d_model', _opt(128, int)), ('--d_feedforward', _opt(128, int)), ('--d_latent', _opt(128, int)),
        ('--property_predictor', _SWITCH), ('--d_property_predictor', _opt(256, int)),
        ('--depth_property_predictor', _opt(2, int)),
        ('--batch_size', _opt(500, int)), ('--batch_chunks', _opt(5, int)),
        ('--beta', _opt(0.05, float)), ('--beta_init', _opt(1e-8, float)), ('--anneal_start', _opt(0, int)),
        ('--adam_lr', _opt(3e-4, float)), ('--lr_scale', _opt(1, float)), ('--warmup_steps', _opt(10000, int)),
        ('--eps_scale', _opt(1, float)), ('--epochs', _opt(100, int)),
        ('--data_source', dict(choices=['zinc', 'pubchem', 'custom'], required=True, type=str)),
        ('--train_mols_path', _opt(None, str)), ('--test_mols_path', _opt(None, str)),
        ('--train_props_path', _opt(None, str)), ('--test_props_path', _opt(None, str)),
        ('--vocab_path', _opt(None, str)), ('--char_weights_path', _opt(None, str)),
        ('--checkpoint', _opt(None, str)),
        ('--save_name', _opt(None, str)), ('--save_freq', _opt(5, int)),
        # additions (no counterpart in the reference)
        ('--compute_dtype', dict(default=None, choices=['fp32', 'f32x3', 'bf16'], type=str)),   # None: TVAE_COMPUTE_DTYPE, else fp32
    ])


def sample_parser():
    return _add(argparse.ArgumentParser(), [
        ('--model', dict(choices=MODELS, required=True, type=str)),
        ('--model_ckpt', dict(required=True, type=str)), ('--mols', _opt(None, str)),
        ('--sample_mode', dict(choices=['rand', 'high_entropy', 'k_high_entropy'], required=True, type=str)),
        ('--k', _opt(15, int)), ('--condition', _opt('', str)), ('--entropy_cutoff', _opt(5, float)),
        ('--n_samples', _opt(30000, int)), ('--n_samples_per_batch', _opt(100, int)),
        ('--save_path', _opt(None, str)),
    ])


def attn_parser():
    return _add(argparse.ArgumentParser(), [
        ('--model', dict(choices=['transvae', 'rnnattn'], required=True, type=str)),
        ('--model_ckpt', dict(required=True, type=str)), ('--mols', dict(required=True, type=str)),
        ('--n_samples', _opt(5000, int)), ('--batch_size', _opt(500, int)), ('--batch_chunks', _opt(5, int)),
        ('--shuffle', _SWITCH), ('--save_path', _opt(None, str)),
    ])


def vocab_parser():
    return _add(argparse.ArgumentParser(), [
        ('--mols', dict(required=True, type=str)),
        ('--freq_penalty', _opt(0.5, float)), ('--pad_penalty', _opt(0.1, float)),
        ('--vocab_name', _opt('custom_char_dict', str)), ('--weights_name', _opt('custom_char_weights', str)),
        ('--save_dir', _opt('data', str)),
    ])


def default_save_name(args):
    "trans{d_ff/d_model}x-{d_model}_{source}, or {model}-{d_model}_{source} for the RNN models (reference scripts/parsers.py:7-17)"
    if args.save_name is not None:
        return args.save_name
    if args.model == 'transvae':
        return 'trans{}x-{}_{}'.format(args.d_feedforward // args.d_model, args.d_model, args.data_source)
    return '{}-{}_{}'.format(args.model, args.d_model, args.data_source)


def model_init(args, params={}):
    "the model class --model names, built from the flags (reference scripts/parsers.py:19-34)"
    common = dict(params=params, name=default_save_name(args), d_model=args.d_model, d_latent=args.d_latent,
                  property_predictor=args.property_predictor, d_pp=args.d_property_predictor,
                  depth_pp=args.depth_property_predictor)
    if args.model == 'transvae':
        return TransVAE(d_ff=args.d_feedforward, **common)
    return {'rnnattn': RNNAttn, 'rnn': RNN}[args.model](**common)


def model_load(model, ckpt_fn):
    "the model class --model names, restored from a checkpoint (reference scripts/sample.py:15-21)"
    return {'transvae': TransVAE, 'rnnattn': RNNAttn, 'rnn': RNN}[model](load_fn=ckpt_fn)


class _PlainUnpickler(pickle.Unpickler):
    "a char_dict pickle holds a dict of str -> int: nothing in it needs a global, so none is resolved"

    def find_class(self, module, name):
        raise pickle.UnpicklingError("vocabulary file references %s.%s; expected a plain dict" % (module, name))


def load_char_dict(path):
    "token -> id dict from a .json file or from the reference's char_dict .pkl (plain-data pickle only)"
    with open(path, 'rb') as f:
        raw = f.read()
    if path.endswith('.json'):
        obj = json.loads(raw.decode())
        if 'tokens' in obj:                                   # the layout of transvae_amd/data/vocab_*.json
            return {tok: i for i, tok in enumerate(obj['tokens'])}
        return obj['char_dict'] if 'char_dict' in obj else obj
    obj = _PlainUnpickler(io.BytesIO(raw)).load()
    if not isinstance(obj, dict):
        raise ValueError("%s does not hold a dict" % path)
    return {str(k): int(v) for k, v in obj.items()}


def load_vocab(args):
    "(char_dict, org_dict, char_weights|None) for --data_source zinc|pubchem (bundled tables) or custom files"
    if args.data_source != 'custom':
        return synth.load_vocab(args.data_source)
    assert args.train_mols_path is not None and args.test_mols_path is not None and args.vocab_path is not None, \
        "ERROR: Must specify files for train/test data and vocabulary"
    char_dict = load_char_dict(args.vocab_path)
    weights = None
    if args.char_weights_path is not None:
        weights = np.load(args.char_weights_path)
    return char_dict, build_org_dict(char_dict), weights
