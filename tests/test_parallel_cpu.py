"""Data-parallel gradient arena + bucketed all-reduce on CPU (gloo, world_size 2).  The HIP model
cannot run here, so a small torch model stands in for it: what is under test is the arena layout,
the bucket bookkeeping (hook route and direct-accumulation 'sink' route) and the reduction."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Opt:
    def __init__(self):
        self.optimizer = type("O", (), {"grad_scale": 1.0})()


class _Shell:
    def __init__(self, model):
        self.model = model
        self.optimizer = _Opt()


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                               torch.nn.Linear(32, 4))


def _worker(rank, world, port, overlap, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transvae_amd.parallel import DataParallel
    torch.manual_seed(100 + rank)
    model = _model()
    if rank == 1:                      # ranks start different: the constructor must broadcast rank 0's weights
        for p in model.parameters():
            p.data.add_(1.0)
    dp = DataParallel(_Shell(model), n_buckets=3, overlap=overlap)
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for step in range(2):              # two steps: the arena must be reusable after zeroing
        dp.arena.flat.zero_()
        # chunk 1 (not armed), chunk 2 (armed): accumulate like VAEShell.train_step
        ((model(xs[:2]) - ys[:2]) ** 2).mean().backward()
        dp.arm()
        ((model(xs[2:]) - ys[2:]) ** 2).mean().backward()
        dp.finish()
    scale = dp.vae.optimizer.optimizer.grad_scale
    q.put((rank, (dp.arena.packed() * scale).clone(), [p.grad.data_ptr() for p in model.parameters()],
           dp.arena.flat.data_ptr(), dp.arena.flat.numel(), len(dp.arena.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def _run(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process truth: average over ranks of (sum over the rank's two chunk-mean gradients)
    model = _model()
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)
    for r in range(2):
        xs, ys = x[r * 4:(r + 1) * 4], y[r * 4:(r + 1) * 4]
        ((model(xs[:2]) - ys[:2]) ** 2).mean().backward()
        ((model(xs[2:]) - ys[2:]) ** 2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in reversed(list(model.parameters()))]) / 2
    for rank, flat, ptrs, base, numel, nb in res:
        assert 2 <= nb <= 3
        assert torch.allclose(flat, ref, rtol=1e-5, atol=1e-7), (rank, (flat - ref).abs().max())
        assert all(base <= p < base + numel * 4 and (p - base) % 256 == 0 for p in ptrs)   # every grad is an aligned arena view
    assert torch.equal(res[0][1], res[1][1])


def test_bucketed_allreduce_overlapped_hooks():
    _run(True)


def test_bucketed_allreduce_no_overlap():
    _run(False)


def test_grad_sink_counts_multi_use_parameters():
    from transvae_amd import ops
    done = []
    sink = ops.GradSink(on_complete=lambda p: done.append(p.data_ptr()))
    a, b = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2))
    sink.use(a); sink.use(a); sink.use(b)
    sink.done(a)
    assert done == []
    sink.done(b); sink.done(a)
    assert done == [b.data_ptr(), a.data_ptr()]
    assert sink.target(a) is a.grad and a.grad.shape == a.shape


# ---- a parameter whose gradient is added into param.grad by its own "kernel" (the sink route) ------
class _DirectLinear(torch.autograd.Function):
    """y = x W^T whose backward adds dW straight into W.grad and returns None for W, the way the HIP
    Functions do inside VAEShell.train_step.  autograd still runs W's post-accumulate hook, which the
    bucket bookkeeping must not count as a second completion (that launched buckets early once)."""

    @staticmethod
    def forward(ctx, x, W):
        from transvae_amd import ops
        ops._note_use(W)
        ctx.save_for_backward(x, W)
        return x @ W.t()

    @staticmethod
    def backward(ctx, dy):
        from transvae_amd import ops
        x, W = ctx.saved_tensors
        sink = ops._sink[0]
        sink.target(W).add_(dy.t() @ x)
        sink.done(W)
        return dy @ W, None


class _Mixed(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Linear(16, 32)
        self.w1 = torch.nn.Parameter(torch.randn(32, 32) * 0.2)
        self.w2 = torch.nn.Parameter(torch.randn(32, 32) * 0.2)
        self.b = torch.nn.Linear(32, 4)

    def forward(self, x):
        h = torch.tanh(self.a(x))
        h = torch.tanh(_DirectLinear.apply(h, self.w1))
        h = torch.tanh(_DirectLinear.apply(h, self.w2))
        return self.b(h)

    def plain(self, x):
        h = torch.tanh(self.a(x))
        return self.b(torch.tanh(torch.tanh(h @ self.w1.t()) @ self.w2.t()))


def _mixed_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transvae_amd import ops
    from transvae_amd.parallel import DataParallel
    model = _Mixed()
    dp = DataParallel(_Shell(model), n_buckets=2, overlap=True, reduce="sum")
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    dp.sink.pending.clear()
    with ops.direct_grads(dp.sink):
        loss = ((model(xs) - ys) ** 2).mean()
        dp.arm()
        loss.backward()
    launched_early = len(dp._launched)
    dp.finish()
    q.put((rank, dp.arena.packed().numpy().copy(), launched_early))
    dist.barrier()
    dist.destroy_process_group()


def test_direct_accumulation_parameters_complete_once():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mixed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _Mixed()
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)
    for r in range(2):
        ((model.plain(x[r * 4:(r + 1) * 4]) - y[r * 4:(r + 1) * 4]) ** 2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in reversed(list(model.parameters()))]).numpy()
    for rank, flat, early in res:
        assert early >= 1                                   # at least one bucket went out from a hook
        assert abs(flat - ref).max() < 1e-6, (rank, abs(flat - ref).max())


def test_arena_keeps_packed_qkv_gradients_adjacent():
    """GradArena lays the q | k | v projections that MultiHeadedAttention.pack_qkv packed into one matrix out the same way
    for their gradients (weights q, k, v consecutive in forward order, then the three biases), so that under data
    parallelism ops.AttnSublayerFn still issues ONE weight-gradient launch per attention block; every slice stays
    256-byte aligned, every parameter gets exactly one slice and the buckets still tile the arena."""
    from transvae_amd import ops, synth
    from transvae_amd.parallel import GradArena
    from transvae_amd.trans_models import MultiHeadedAttention, build_encoder_decoder
    model = build_encoder_decoder(26, 3, 128, 128, 128, 4, 0.1, False, False, 256, 2, 1, 126, 125)
    mhas = [m for m in model.modules() if isinstance(m, MultiHeadedAttention)]
    assert len(mhas) == 10
    for m in mhas:
        m.pack_qkv()
    arena = GradArena(model, n_buckets=4)
    params = [p for p in model.parameters() if p.requires_grad]
    assert len(arena.params) == len(params) and {id(p) for p in arena.params} == {id(p) for p in params}
    for m in mhas:
        gw = [l.weight.grad for l in m.linears[:3]]
        gb = [l.bias.grad for l in m.linears[:3]]
        assert ops._adjacent(*gw) and ops._adjacent(*gb)
        assert ops._adjacent(gw[1], gw[2]) and ops._adjacent(gb[1], gb[2])        # the k | v run of source attention
        assert ops._run(gw).shape == (3 * 128, 128)
    base = arena.flat.data_ptr()
    spans = sorted((p.grad.data_ptr() - base, p.numel() * 4) for p in params)
    for (o, n), (o2, _) in zip(spans, spans[1:]):
        assert o % 256 == 0 and o + n <= o2                                    # aligned, disjoint
    assert arena.buckets[0][0] == 0 and arena.buckets[-1][1] == arena.flat.numel()
    assert all(a[1] == b[0] for a, b in zip(arena.buckets, arena.buckets[1:]))
    assert sum(b[2] for b in arena.buckets) == len(params)
