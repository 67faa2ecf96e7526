"""Data-parallel training: one process per GPU, gradients all-reduced over RCCL/xGMI.

Layout: all parameter gradients live in ONE flat fp32 arena; ``p.grad`` of every parameter is a
view into it, so autograd accumulates straight into the arena and no pack/unpack copy exists.
The arena is cut into a few contiguous buckets in the order backward finishes them
(generator + decoder layers -> deconv/linear -> encoder bottleneck -> encoder layers + embeddings).
During the LAST accumulation chunk a bucket's all-reduce is launched (async, on RCCL's own
stream) as soon as its last gradient has been accumulated, so communication overlaps the rest of
backward; xGMI is point-to-point, so few large messages (3-4 buckets of ~8 MB for trans4x-256)
is the shape that suits it.  The 1/world averaging is folded into the fused Adam launch
(grad_scale), so no extra pass over the gradients is needed.

Semantics.  The reference normalises the loss per CHUNK (BATCH_SIZE / BATCH_CHUNKS molecules, weighted
CE denominator included) and SUMS the chunk gradients (trans_models.py:202-236).  Ranks hold whole
chunks, so no normaliser ever spans two ranks and nothing but the gradients has to be exchanged:
  reduce='sum'   the step is exactly the reference's step with BATCH_SIZE x world molecules in
                 BATCH_CHUNKS x world chunks (checked on the GPU by tests/test_parallel_gpu.py);
  reduce='mean'  (default) the summed gradient divided by world, i.e. the usual data-parallel
                 convention; Adam's update is invariant to that factor up to its eps of 1e-9.
"""
import torch
import torch.distributed as dist


class _Done:
    def wait(self):
        return True


def _host_staged():
    "gloo moves host memory: device buffers are staged through the host (CPU-side tests and rehearsals only)"
    return dist.get_backend() == "gloo"


def _all_reduce_sum(t):
    "launch a SUM all-reduce of a flat fp32 tensor; returns an object with .wait()"
    if t.is_cuda and _host_staged():
        host = t.cpu()                     # synchronises with the stream that produced the gradients
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        t.copy_(host)
        return _Done()
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


def _broadcast(t, src=0):
    if t.is_cuda and _host_staged():
        host = t.cpu()
        dist.broadcast(host, src=src)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src)


def mean_over_ranks(values):
    "average a few Python floats over the ranks (epoch losses: every rank must take the same checkpoint decision)"
    t = torch.tensor(list(values), dtype=torch.float64)
    if not _host_staged():
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return (t / dist.get_world_size()).tolist()


class GradArena:
    ALIGN = 64                      # floats

    def packed(self):
        "the gradients without the alignment padding, in arena order"
        return torch.cat([self.flat[o:o + n] for o, n in self.slices])

    def __init__(self, model, n_buckets=4):
        params = [p for p in model.parameters() if p.requires_grad]
        # backward reaches parameters roughly in reverse registration order -- except that the q | k | v projections of an
        # attention block that MultiHeadedAttention.pack_qkv laid out as row blocks of one matrix keep that layout for
        # their GRADIENTS too (weights q, k, v consecutive, then biases q, k, v, in forward order): ops.AttnSublayerFn
        # then still finds them adjacent and issues ONE split-K launch for the three weight and bias gradients, as in the
        # single-GPU step (256 x 256 blocks and 256-float biases are whole numbers of 64-float slices: no padding between)
        from . import ops
        runs = {}
        for m in model.modules():
            lin = getattr(m, "linears", None)
            if lin is None or len(lin) < 3:
                continue
            Ws, bs = [l.weight for l in lin[:3]], [l.bias for l in lin[:3]]
            if all(w.requires_grad for w in Ws + bs) and ops._adjacent(*[w.data for w in Ws]) and ops._adjacent(*[b.data for b in bs]):
                for run in (Ws, bs):
                    for q in run:
                        runs[id(q)] = run
        order, seen = [], set()
        for p in reversed(params):
            if id(p) in seen:
                continue
            for q in runs.get(id(p), [p]):
                order.append(q)
                seen.add(id(q))
        # every slice starts on a 256-byte boundary: the weight-gradient kernels add 128-byte runs with fp32 atomics,
        # and a slice that starts mid cache line makes every run straddle two (measured: +16 us on each such launch)
        pad = lambda n: (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        total = sum(pad(p.numel()) for p in order)
        dev = order[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.buckets = []           # (start, end, set(param ids))
        self.slices = []            # (offset, numel) per parameter, in arena order
        target = (total + n_buckets - 1) // n_buckets
        off, bstart, members = 0, 0, []
        self.bucket_of = {}
        for p in order:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self.slices.append((off, n))
            members.append(p)
            off += pad(n)
            if off - bstart >= target or p is order[-1]:
                bi = len(self.buckets)
                for q in members:
                    self.bucket_of[q.data_ptr()] = bi
                self.buckets.append((bstart, off, len(members)))
                bstart, members = off, []
        self.params = order


class DataParallel:
    """Wraps a VAEShell: ``step(data, beta)`` = VAEShell.train_step with bucketed gradient all-reduce."""

    def __init__(self, vae, n_buckets=4, overlap=True, reduce="mean"):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if reduce not in ("mean", "sum"):
            raise ValueError("reduce must be 'mean' or 'sum'")
        self.vae = vae
        self.world = dist.get_world_size()
        self.arena = GradArena(vae.model, n_buckets)
        self.overlap = overlap
        self._armed = False
        self._pending = []
        self._count = [0] * len(self.arena.buckets)
        self._launched = set()
        # two routes tell us that a parameter's gradient is final for this step: the kernels that
        # accumulate straight into param.grad report through the sink; the few parameters whose
        # gradient still travels through autograd (conv / deconv kernels) through a hook
        from . import ops
        self.sink = ops.GradSink(on_complete=self._hook if overlap else None)
        self._seen = set()
        self._handles = []
        self.closed = False
        self.timing = False              # bench.py: record an event pair around the waits of finish()
        self.wait_events = []
        if overlap:
            for p in self.arena.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._autograd_hook))
        # parameters start identical on every rank
        for p in vae.model.parameters():
            _broadcast(p.data, src=0)
        ops.weights_changed()                # written through .data: invalidate the pre-split copies
        self._set_grad_scale(1.0 / self.world if reduce == "mean" else 1.0)

    def _set_grad_scale(self, s):
        self.vae.optimizer.optimizer.grad_scale = s

    def close(self):
        """Detach from the shell: autograd hooks removed, the optimizer's gradient scale back to 1.  The parameters'
        .grad stay views of the arena (harmless: a later single-process step accumulates into them as into any
        buffer).  A closed engine must not be stepped again; VAEShell._data_parallel builds a new one."""
        if self.closed:
            return
        for h in self._handles:
            h.remove()
        self._handles = []
        self.sink.on_complete = None
        self._set_grad_scale(1.0)
        self.closed = True

    def _autograd_hook(self, p):
        # autograd runs the post-accumulate hook of EVERY parameter it visits, including those whose
        # Function returned no gradient because its kernels add into param.grad themselves; for
        # those only the sink knows when the last contribution has been launched
        if p.data_ptr() in self.sink.pending:
            return
        self._hook(p)

    def _hook(self, p):
        if not self._armed:
            return
        bi = self.arena.bucket_of.get(p.data_ptr())
        if bi is None or p.data_ptr() in self._seen:
            return
        self._seen.add(p.data_ptr())
        self._count[bi] += 1
        if self._count[bi] >= self.arena.buckets[bi][2] and bi not in self._launched:
            self._launch(bi)

    def _launch(self, bi):
        s, e, _ = self.arena.buckets[bi]
        self._launched.add(bi)
        from . import ops
        buf = self.arena.flat[s:e]
        side = ops.side_stream(buf.device)
        if side is None:
            self._pending.append(_all_reduce_sum(buf))
            return
        # Weight-gradient products run on a second stream (ops._side_begin).  The collective must see them done, the backward
        # pass need not: the all-reduce is issued FROM that stream (which first waits for what the compute stream has
        # launched: LayerNorm / embedding / conv gradients of the bucket), so RCCL's stream orders itself behind both and
        # the compute stream is not held up.  Joining the compute stream here instead cost 1.9 ms per step (4 buckets,
        # each waiting out the weight gradients still queued: 32.1 vs 30.2 ms single-process on one box).
        side.wait_stream(torch.cuda.current_stream(buf.device))
        with torch.cuda.stream(side):
            self._pending.append(_all_reduce_sum(buf))

    def arm(self):
        "call before the backward of the last accumulation chunk"
        self._armed = True
        self._count = [0] * len(self.arena.buckets)
        self._pending = []
        self._launched = set()
        self._seen = set()

    def finish(self):
        "after the last backward: reduce whatever was not launched from hooks, wait for all buckets"
        if not self._armed:
            self._pending, self._launched = [], set()
        for bi in range(len(self.arena.buckets)):
            if bi not in self._launched:                  # not launched from a hook (or overlap off)
                self._launch(bi)
        ev = None
        if self.timing and self.arena.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._pending:
            w.wait()                      # RCCL: the compute stream waits for the collective's stream, the host does not
        if ev is not None:
            ev[1].record()
            self.wait_events.append(ev)
        self._armed = False
        self._pending = []

    def step(self, data, beta, eps=None):
        if self.closed:
            raise RuntimeError("DataParallel.step on a closed engine")
        return self.vae.train_step(data, beta, eps=eps, dp=self)
