"""Sample molecules from a trained TransVAE / RNNAttn / RNN (flags: scripts/parsers.py; behaviour: reference
scripts/sample.py:13-59).  Under torch.distributed every rank decodes its share of --n_samples and
rank 0 writes the CSV."""
import os
import sys

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.parsers import model_load, sample_parser  # noqa: E402
from transvae_amd.tvae_util import calc_entropy, decode_mols  # noqa: E402


def sample(args):
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group(os.environ.get('TVAE_DIST_BACKEND', 'nccl'))
        torch.manual_seed(torch.initial_seed() + rank)
        np.random.seed(rank)
    vae = model_load(args.model, args.model_ckpt)
    condition = [] if args.condition == '' else args.condition.split(',')

    sample_mode, sample_dims = 'rand', None
    if args.sample_mode != 'rand':
        _, mus, _ = vae.calc_mems(pd.read_csv(args.mols).to_numpy(), log=False, save=False)
        sample_dims = np.where(np.array(calc_entropy(mus)) > args.entropy_cutoff)[0]
        sample_mode = 'top_dims' if args.sample_mode == 'high_entropy' else 'k_dims'

    # The reference decodes batch by batch (:47-51, over-shooting to whole batches of n_samples_per_batch); a 100-latent
    # decode is launch-bound, so here only the LATENTS are drawn batch by batch -- same calls in the same order, hence the
    # same z for the same seed -- and all of a rank's latents are decoded together in slabs of up to 65,536
    # (VAEShell.decode_latents: rows are independent, the tokens are those of the per-batch loop).  TVAE_SAMPLE_PER_BATCH=1
    # restores the literal loop.  What travels between ranks is ONE gather of token ids, one byte each (26-token vocabulary).
    def n_batches(r):
        share = args.n_samples // world + (1 if r < args.n_samples % world else 0)
        return -(-share // args.n_samples_per_batch)
    import torch
    dev = vae._device()
    if os.environ.get('TVAE_SAMPLE_PER_BATCH') == '1':
        ids = [vae.sample(args.n_samples_per_batch, sample_mode=sample_mode, sample_dims=sample_dims, k=args.k,
                          condition=condition, return_str=False).to(torch.uint8) for _ in range(n_batches(rank))]
        ids = torch.cat(ids) if ids else torch.empty((0, vae.tgt_len), dtype=torch.uint8, device=dev)
    else:
        zs = [vae.sample_from_memory(args.n_samples_per_batch, mode=sample_mode, sample_dims=sample_dims, k=args.k)
              for _ in range(n_batches(rank))]
        ids = vae.decode_latents(torch.cat(zs), condition=condition) if zs \
            else torch.empty((0, vae.tgt_len), dtype=torch.uint8, device=dev)     # on the model's device even when empty
    if world > 1:
        rows = [n_batches(r) * args.n_samples_per_batch for r in range(world)]
        padded = torch.zeros((max(rows), ids.shape[1]), dtype=torch.uint8, device=ids.device)
        padded[:ids.shape[0]] = ids
        if dist.get_backend() == 'gloo':
            padded = padded.cpu()
        parts = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
        dist.gather(padded, parts, dst=0)
        if rank == 0:
            ids = torch.cat([part[:n] for part, n in zip(parts, rows)])
        dist.destroy_process_group()
    samples = decode_mols(ids, vae.params['ORG_DICT']) if rank == 0 else []
    if rank == 0:
        save_path = args.save_path
        if save_path is None:
            os.makedirs('generated', exist_ok=True)
            save_path = 'generated/{}_{}.csv'.format(vae.name, args.sample_mode)
        pd.DataFrame(samples, columns=['mol']).to_csv(save_path, index=False)


if __name__ == '__main__':
    sample(sample_parser().parse_args())
