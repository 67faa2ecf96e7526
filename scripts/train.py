"""Train a TransVAE / RNNAttn / RNN on MI355X (flags: scripts/parsers.py; behaviour: reference scripts/train.py:14-84).

    python scripts/train.py --model transvae --d_model 256 --d_feedforward 1024 --data_source zinc
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train.py ...   # data parallel
"""
import os
import sys

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.parsers import load_vocab, model_init, train_parser  # noqa: E402
from transvae_amd import ops  # noqa: E402
from transvae_amd.trans_models import _load_checkpoint  # noqa: E402


def _read_column_file(path):
    return pd.read_csv(path).to_numpy()


def train(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    if args.compute_dtype is not None:             # otherwise TVAE_COMPUTE_DTYPE (read at import) or exact fp32
        ops.set_compute_dtype(args.compute_dtype)

    # resuming: the KL annealer restarts from where the finished epochs left beta (reference :16-21)
    if args.checkpoint is not None:
        done = _load_checkpoint(args.checkpoint)['epoch']
        args.beta_init = (args.beta - args.beta_init) / (done + args.epochs) * done

    params = {'ADAM_LR': args.adam_lr, 'ANNEAL_START': args.anneal_start, 'BATCH_CHUNKS': args.batch_chunks,
              'BATCH_SIZE': args.batch_size, 'BETA': args.beta, 'BETA_INIT': args.beta_init,
              'EPS_SCALE': args.eps_scale, 'LR_SCALE': args.lr_scale, 'WARMUP_STEPS': args.warmup_steps}
    char_dict, org_dict, weights = load_vocab(args)
    if weights is not None:
        params['CHAR_WEIGHTS'] = np.asarray(weights, dtype=np.float32)
    params['CHAR_DICT'], params['ORG_DICT'] = char_dict, org_dict

    if args.data_source == 'custom':
        train_path, test_path = args.train_mols_path, args.test_mols_path
    else:
        train_path, test_path = ('data/{}_{}.txt'.format(args.data_source, s) for s in ('train', 'test'))
    train_mols, test_mols = _read_column_file(train_path), _read_column_file(test_path)
    train_props = test_props = None
    if args.property_predictor:
        assert args.train_props_path is not None and args.test_props_path is not None, \
            "ERROR: Must specify files with train/test properties if training a property predictor"
        train_props, test_props = _read_column_file(args.train_props_path), _read_column_file(args.test_props_path)

    vae = model_init(args, params)
    if args.checkpoint is not None:
        vae.load(args.checkpoint)
    vae.train(train_mols, test_mols, train_props, test_props, epochs=args.epochs, save_freq=args.save_freq)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    train(train_parser().parse_args())
