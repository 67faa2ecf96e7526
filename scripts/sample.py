"""Sample molecules from a trained TransVAE (flags: scripts/parsers.py; behaviour: reference
scripts/sample.py:13-59).  Under torch.distributed every rank decodes its share of --n_samples and
rank 0 writes the CSV."""
import os
import sys

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.parsers import require_transformer, sample_parser  # noqa: E402
from transvae_amd.trans_models import TransVAE  # noqa: E402
from transvae_amd.tvae_util import calc_entropy  # noqa: E402


def sample(args):
    require_transformer(args)
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
        torch.manual_seed(torch.initial_seed() + rank)
        np.random.seed(rank)
    vae = TransVAE(load_fn=args.model_ckpt)
    condition = [] if args.condition == '' else args.condition.split(',')

    sample_mode, sample_dims = 'rand', None
    if args.sample_mode != 'rand':
        _, mus, _ = vae.calc_mems(pd.read_csv(args.mols).to_numpy(), log=False, save=False)
        sample_dims = np.where(np.array(calc_entropy(mus)) > args.entropy_cutoff)[0]
        sample_mode = 'top_dims' if args.sample_mode == 'high_entropy' else 'k_dims'

    mine = args.n_samples // world + (1 if rank < args.n_samples % world else 0)
    samples = []
    while len(samples) < mine:
        samples.extend(vae.sample(args.n_samples_per_batch, sample_mode=sample_mode, sample_dims=sample_dims,
                                  k=args.k, condition=condition))
    if world > 1:
        parts = [None] * world if rank == 0 else None
        dist.gather_object(samples, parts, dst=0)
        if rank == 0:
            samples = [s for part in parts for s in part]
        dist.destroy_process_group()
    if rank == 0:
        save_path = args.save_path
        if save_path is None:
            os.makedirs('generated', exist_ok=True)
            save_path = 'generated/{}_{}.csv'.format(vae.name, args.sample_mode)
        pd.DataFrame(samples, columns=['mol']).to_csv(save_path, index=False)


if __name__ == '__main__':
    sample(sample_parser().parse_args())
