"""Data parallelism on the real kernels: two ranks (both on cuda:0, gloo moves the buckets) must
take exactly the step one process takes on the concatenated batch with twice the chunks -- the
reference's accumulate-the-chunk-means semantics (trans_models.py:202-236) on the global batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

ARCH = "trans1x-128"
PER_RANK, CHUNKS = 8, 2


def _set_case(arch, per_rank, chunks):
    "module-level so that spawned workers (which re-import this file) can be told the case through their arguments"
    global ARCH, PER_RANK, CHUNKS
    ARCH, PER_RANK, CHUNKS = arch, per_rank, chunks


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(batch, chunks):
    from transvae_amd import synth
    from transvae_amd.trans_models import TransVAE
    cd, od, w = synth.load_vocab("zinc")
    params = {"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w, "BATCH_SIZE": batch, "BATCH_CHUNKS": chunks}
    if ARCH == "rnnattn-128":           # the recurrent family: AdamOpt, no length head, one parameter (tgt_embed) never used
        from transvae_amd.rnn_models import RNNAttn
        vae = RNNAttn(params=params, name="dp", d_model=128, N=2)
    else:
        vae = TransVAE(params=params, name="dp", **synth.ARCHS[ARCH])
    vae.model.load_state_dict(synth.synth_state_dict(vae.model.state_dict(), 0))
    vae.model.eval()                    # dropout off: the step is a deterministic function of (data, eps)
    return vae


def _inputs(world):
    from transvae_amd import synth
    n = PER_RANK * world
    toks = synth.synth_tokens(n, seed=3).float()
    data = torch.cat([toks, torch.zeros(n, 1)], 1)
    eps = torch.randn(n, 128, generator=torch.Generator().manual_seed(9))
    return data, eps


def _moments(vae):
    opt = vae.optimizer.optimizer
    return {k: opt.state[p]["exp_avg"].detach().cpu().numpy().copy() for k, p in vae.model.named_parameters()}


def _worker(rank, world, port, reduce, overlap, q, case=("trans1x-128", 8, 2)):
    _set_case(*case)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transvae_amd.parallel import DataParallel
    vae = _build(PER_RANK, CHUNKS)
    if rank == 1:                                        # the constructor must bring rank 0's weights over
        with torch.no_grad():
            for p in vae.model.parameters():
                p.add_(0.01)
    dp = DataParallel(vae, reduce=reduce, overlap=overlap)
    data, eps = _inputs(world)
    sl = slice(rank * PER_RANK, (rank + 1) * PER_RANK)
    table = dp.step(data[sl].cuda(), 0.03, eps=eps[sl].cuda())
    q.put((rank, table.cpu().numpy().copy(), _moments(vae)))      # numpy: no shared-memory handles that die with the worker
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("reduce,overlap,case", [("sum", True, ("trans1x-128", 8, 2)), ("mean", True, ("trans1x-128", 8, 2)),
                                                 ("sum", False, ("trans1x-128", 8, 2)),
                                                 ("mean", True, ("rnnattn-128", 8, 2)),
                                                 # BASELINE config 3's model at the reference's chunk size: trans4x-256, 100 per rank
                                                 ("mean", True, ("trans4x-256", 100, 1))])
def test_two_ranks_equal_one_process_on_the_global_batch(reduce, overlap, case):
    world = 2
    _set_case(*case)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, reduce, overlap, q, case)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=200) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    vae = _build(PER_RANK * world, CHUNKS * world)
    data, eps = _inputs(world)
    table = vae.train_step(data.cuda(), 0.03, eps=eps.cuda()).cpu().numpy()
    ref = _moments(vae)
    scale = 1.0 if reduce == "sum" else 1.0 / world
    # per-chunk loss rows: rank r owns chunks r*CHUNKS .. (r+1)*CHUNKS-1 of the global batch
    for rank, tab, _ in res:
        np.testing.assert_allclose(tab, table[rank * CHUNKS:(rank + 1) * CHUNKS], rtol=1e-6, atol=1e-7)
    # the only difference is the order of the cross-rank sum (the fp32 atomics of the weight
    # gradients are order-dependent too): a few ulp.  Key-projection biases have an analytically
    # zero gradient (softmax is shift invariant): pure rounding noise, bounded absolutely.
    worst = 0.0
    for k, m_ref in ref.items():
        for rank, _, mom in res:
            num = float(np.linalg.norm(mom[k] - scale * m_ref))
            den = float(np.linalg.norm(scale * m_ref))
            assert num <= 2e-5 * den + 1e-8, (k, rank, num, den)
            if den > 1e-6:
                worst = max(worst, num / den)
    print("worst relative difference of the first Adam moment: %.2e" % worst)
    for k in ref:                                         # both ranks hold the same reduced gradient
        assert np.array_equal(res[0][2][k], res[1][2][k]), k


def _train_worker(rank, world, port, tmp, q):
    _set_case("trans1x-128", 8, 2)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.chdir(tmp)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "data.npz"))
    mols = np.array([str(s) for s in g["zinc_smiles"][:100]], dtype=object)[:, None]
    vae = _build(PER_RANK, CHUNKS)
    vae.model.train()
    torch.manual_seed(100 + rank)                       # the reparameterisation noise differs per rank, as in a real run
    vae.train(mols, mols[:40], epochs=2, save=True, save_freq=2, log=True, log_dir="trials")
    sd = {k: v.detach().cpu().numpy().copy() for k, v in vae.model.state_dict().items()}
    # train() closed its data-parallel engine: no hooks left behind, the optimizer's gradient scale back to 1
    assert vae._dp.closed and vae.optimizer.optimizer.grad_scale == 1.0
    assert all(len(p._post_accumulate_grad_hooks or {}) == 0 for p in vae.model.parameters())
    q.put((rank, sd, vae.n_epochs, float(vae.best_loss)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_loop(tmp_path):
    _set_case("trans1x-128", 8, 2)
    """VAEShell.train under torch.distributed: every rank walks its DistributedSampler share of the HBM-resident
    dataset, gradients are all-reduced, only rank 0 logs and saves, and the ranks end with identical weights and
    the same best-loss decision."""
    import pandas as pd
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, sd0, ep0, best0), (_, sd1, ep1, best1) = res
    assert ep0 == ep1 == 2 and best0 == best1
    assert all(np.array_equal(sd0[k], sd1[k]) for k in sd0)
    log = pd.read_csv(tmp_path / "trials" / "log_dp.txt")
    # 100 molecules over 2 ranks = 50 each = 6 batches of 8; 40 validation molecules = 20 each = 2 batches
    assert list(log["data_type"]) == (["train"] * 6 + ["test"] * 2) * 2 and np.isfinite(log["tot_loss"]).all()
    assert sorted(os.listdir(tmp_path / "checkpoints")) == ["002_dp.ckpt", "best_dp.ckpt"]
