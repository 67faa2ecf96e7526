"""Thin Python layer over the C ABI: raw launch helpers + block-level autograd Functions.

torch is used for device memory (allocation of outputs/workspaces), the current HIP stream and
the autograd graph that chains the block-level Functions.  Every piece of arithmetic on the
TransVAE hot path runs in libtvae_hip.so; nothing here falls back to ATen math.
"""
import ctypes as C
import math

import os

import torch

from . import _lib
from ._lib import (F32, BF16, F32X3, GEMM_B_SPLIT3, GEMM_B_BF16, GEMM_RELU, GEMM_ACCUM, GEMM_A_KMAJOR, GEMM_B_KMAJOR, GEMM_SPLITK,
                   GEMM_A_DROPMASK, GEMM_RELUMASK, call)

# ----------------------------------------------------------------------------- configuration
_DTYPES = {"fp32": F32, "f32": F32, "f32x3": F32X3, "bf16": BF16}
# TVAE_COMPUTE_DTYPE picks the GEMM arithmetic at import (default: exact fp32 MFMA)
_state = {"dtype": _DTYPES[os.environ.get("TVAE_COMPUTE_DTYPE", "fp32")], "site": 0, "wgrad": None, "dgrad": None}
# bf16 mode: arithmetic of the weight-gradient / data-gradient products (None = bf16 like the forward)
_BF16_ROLES = {"wgrad": os.environ.get("TVAE_BF16_WGRAD"), "dgrad": os.environ.get("TVAE_BF16_DGRAD")}


def set_compute_dtype(name, wgrad=None, dgrad=None):
    """GEMM arithmetic: 'fp32' (exact, v_mfma_f32_32x32x2_f32), 'f32x3' (fp32-accurate: six bf16 MFMAs
    on exact 3-way bf16 splits of the operands) or 'bf16' (operands rounded to bf16, fp32 accumulate).
    In 'bf16' mode the weight-gradient and data-gradient products may run in another arithmetic
    (`wgrad` / `dgrad` = 'f32x3' | 'fp32' | None; default from TVAE_BF16_WGRAD / TVAE_BF16_DGRAD)."""
    _state["dtype"] = _DTYPES[name]
    _state["wgrad"] = _state["dgrad"] = None
    if _state["dtype"] == BF16:
        for role, given in (("wgrad", wgrad), ("dgrad", dgrad)):
            pick = given if given is not None else _BF16_ROLES[role]
            _state[role] = _DTYPES[pick] if pick else None


def _role_dtype(role):
    return _state[role] if _state[role] is not None else _state["dtype"]


def compute_dtype():
    return {F32: "fp32", BF16: "bf16", F32X3: "f32x3"}[_state["dtype"]]


def next_site():
    _state["site"] += 1
    return _state["site"]


_rng = {}
# Dropout randomness.  A keep decision is a hash of (seed, step, site, element):
#   seed, step  device uint32[2] per GPU (below): the kernels read them when they RUN, so they may only change between a
#               forward and ITS backward never -- VAEShell.train_step bumps `step` after the optimizer step;
#   site        16 bits of call-site id (one per dropout module) | 16 bits of `pass`, a host counter that
#               EncoderDecoder.encode bumps on every training-mode forward.  The Drop object made in the forward carries
#               the combined number into the backward, so a plain `loss.backward()` loop over vae.model (no train_step),
#               and the unfused chunk loop, draw fresh masks on every forward all the same.
# Nothing here follows torch.manual_seed by itself: the FIRST VAEShell.train of a process on a device seeds it from
# torch.initial_seed() + rank unless seed_rng / rng_restore already did; every later train() call continues the stream
# {seed, step, pass} where it stands (the reference continues torch's global generator the same way), and
# rng_snapshot / rng_restore carry it through a checkpoint.
_pass = {"n": 0, "explicit_seed": False}
_seeded = set()


def rng_state(device):
    """Device uint32[2] {seed, step} shared by every dropout site on `device`."""
    key = (device.type, device.index)
    if key not in _rng:
        _rng[key] = torch.tensor([0x1234567, 0], dtype=torch.int32, device=device)
    return _rng[key]


def seed_rng(device, seed, step=0, n_pass=0, explicit=True):
    "seed the dropout stream of `device` (and restart the pass counter): same seed, same masks"
    st = rng_state(device)
    st.copy_(torch.tensor([seed & 0x7FFFFFFF, step & 0x7FFFFFFF], dtype=torch.int32))
    _pass["n"] = int(n_pass)
    _pass["explicit_seed"] = bool(explicit)
    _seeded.add((device.type, device.index))


def rng_seeded(device):
    "has the dropout stream of `device` been seeded (by the caller, a checkpoint or an earlier train())?"
    return (device.type, device.index) in _seeded


def seeded_explicitly():
    return _pass["explicit_seed"]


def next_pass():
    "a new training-mode forward pass begins: its dropout sites draw masks of their own"
    _pass["n"] = (_pass["n"] + 1) & 0xFFFF


def rng_snapshot(device):
    "{seed, step, pass} as plain ints (synchronises: checkpoint time only)"
    seed, step = (int(v) for v in rng_state(device).cpu().tolist())
    return {"seed": seed, "step": step, "pass": _pass["n"]}


def rng_restore(device, snap):
    seed_rng(device, int(snap["seed"]), int(snap.get("step", 0)), int(snap.get("pass", 0)))


def advance_rng(device):
    call("tvae_rng_advance", _p(rng_state(device)), _stream())


class Drop:
    """One dropout site: probability + site id (+ the device rng).  ``off`` when p == 0."""
    __slots__ = ("p", "site", "rng")

    def __init__(self, p, site, rng):
        self.p, self.site, self.rng = float(p), int(site), rng

    @staticmethod
    def make(p, site, training, device):
        if not training or p <= 0.0:
            return None
        return Drop(p, (site & 0xFFFF) | (_pass["n"] << 16), rng_state(device))


def _drop_args(drop):
    if drop is None:
        return None, 0, 0.0
    return _p(drop.rng), drop.site, drop.p


def _p(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: handing a host address to
    a kernel is a GPU memory fault, not an exception, so it is caught here."""
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.TvaeError("host tensor (shape %s, dtype %s) passed where a device buffer is required"
                             % (tuple(t.shape), t.dtype))
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """the current stream's handle for a launch.  torch.cuda.current_stream() builds a Stream object and resolves the device
    through three Python layers (9 us per call, ~200 calls per train step: 2 ms of a trans1x-128 step that is bound by the
    host's launch rate); the raw getter is the same lookup without them"""
    if _raw_stream is not None and _cur_device is not None:
        return C.c_void_p(_raw_stream(_cur_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise _lib.TvaeError("%s must be a float32 CUDA tensor (got %s on %s)" % (name, t.dtype, t.device))


# ----------------------------------------------------------------------------- raw launchers
def gemm(A, lda, Bm, ldb, Cm, ldc, M, N, K, bias=None, residual=None, ldr=0, alpha=1.0, flags=0,
         drop=None, dtype=None, a_rowsum=None, stream=None):
    r, s, p = _drop_args(drop)
    call("tvae_gemm", _p(A), lda, _p(Bm), ldb, _p(Cm), ldc, M, N, K, _p(bias), _p(residual), ldr,
         alpha, flags, r, s, p, _p(a_rowsum), _state["dtype"] if dtype is None else dtype,
         _stream() if stream is None else stream)


# ---- direct gradient accumulation -------------------------------------------------------------
# Inside VAEShell.train_step the weight/bias/LayerNorm/embedding gradients are accumulated by the
# kernels STRAIGHT into param.grad (split-K atomics and column sums add into it anyway) and the
# Functions return None for those inputs: no zero-filled temporaries, no AccumulateGrad add per
# parameter and chunk.  Outside of it (plain loss.backward() on the modules) gradients are returned
# to autograd as usual.
class GradSink:
    """Hands out param.grad buffers; `uses`/`done` let data parallelism know when a parameter's
    gradient is complete (a parameter can be used more than once per step: decoder.norm)."""

    def __init__(self, on_complete=None):
        self.pending = {}
        self.on_complete = on_complete

    def use(self, p):
        self.pending[p.data_ptr()] = self.pending.get(p.data_ptr(), 0) + 1

    def target(self, p):
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        return p.grad

    def done(self, p):
        n = self.pending.get(p.data_ptr(), 0) - 1
        self.pending[p.data_ptr()] = n
        if n <= 0 and self.on_complete is not None:
            self.on_complete(p)


_sink = [None]


class direct_grads:
    def __init__(self, sink):
        self.sink = sink

    def __enter__(self):
        self.prev = _sink[0]
        _sink[0] = self.sink
        return self.sink

    def __exit__(self, *exc):
        join_side()
        _sink[0] = self.prev


# ---- weight gradients on a second stream ------------------------------------------------------------------------------
# With direct accumulation nothing in the backward pass READS a weight gradient: the split-K product only adds into param.grad.
# So the weight-gradient GEMMs of the Linear layers go to a second stream that runs beside the backward pass proper: its tiles
# (MFMA-bound, a few atomics at the end) fill the rounds a data-gradient GEMM leaves part empty and run under its epilogue, which
# is all HBM traffic.  Measured (tools/debug/overlap_bench.py, MI355X, fp32): a (dgrad, wgrad) pair of a 63000-row 256 -> 256
# layer 176 us back to back, 151 us on two streams; 256 -> 1024: 545 -> 514 us.  The second stream waits for an event of the
# compute stream before every product (its operands are ready by then: the compute stream is in order) and the compute stream
# waits for it ONCE: when the gradients are about to be read (direct_grads.__exit__, a data-parallel bucket, Adam).
# Joining per product would give the gain back (182 us for the pair above).
#   Operands read there are marked with record_stream: the caching allocator does not hand their memory to the compute stream
# again before the second stream has passed the point where they were freed.
# Stream priority makes no difference (29.9 / 30.0 ms per step at equal / raised priority).
# (f32x3 mode: 23.5 -> 24.0 ms per step while every product ran on the register-staged 3 x bf16 kernel at 74 KB of LDS, 23.2 -> 22.9
# since the forward / data-gradient products have the streamed kernel.)  TVAE_WGRAD_STREAM=0 launches everything on the compute stream.
_side = {}            # device index -> [stream, products launched since the last join]


def _runs_beside(main, cand, cycles=300000):
    """True if a kernel on `cand` overlaps one on `main`.  HIP maps its streams onto a few hardware queues (GPU_MAX_HW_QUEUES,
    4 here) in creation order; two streams on one queue run strictly one after the other.  Which pool stream shares the
    compute stream's queue depends on what else the process created first (RCCL's streams under data parallelism: measured,
    the first pool stream then serialised with the compute stream and the step LOST 1 ms), so it is measured: two spin
    kernels of `cycles`, one per stream, take about one spin when the queues differ and two when they do not."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    cand.wait_stream(main)
    ev[0].record(main)
    torch.cuda._sleep(cycles)
    ev[1].record(main)
    with torch.cuda.stream(cand):
        torch.cuda._sleep(cycles)
    main.wait_stream(cand)
    ev[2].record(main)
    ev[2].synchronize()
    return ev[0].elapsed_time(ev[2]) < 1.5 * ev[0].elapsed_time(ev[1])


def _pick_side_stream(dev):
    main = torch.cuda.current_stream(dev)
    first = None
    for _ in range(6):
        cand = torch.cuda.Stream(device=dev)
        if first is None:
            first = cand
        if _runs_beside(main, cand):
            return cand
    return first


def _side_begin(*operands):
    "stream pointer for a weight-gradient product that may run beside the backward pass, or None (compute stream)"
    if _sink[0] is None or os.environ.get("TVAE_WGRAD_STREAM", "1") == "0" or not operands[0].is_cuda:
        return None
    dev = operands[0].device
    ent = _side.get(dev.index)
    if ent is None:
        ent = _side[dev.index] = [_pick_side_stream(dev), 0]
    ent[0].wait_stream(torch.cuda.current_stream(dev))
    for t in operands:
        t.record_stream(ent[0])
    ent[1] += 1
    return C.c_void_p(ent[0].cuda_stream)


def second_stream(device):
    "the second stream of `device` (created on first use; see _side_begin / _pick_side_stream)"
    ent = _side.get(device.index)
    if ent is None:
        ent = _side[device.index] = [_pick_side_stream(device), 0]
    return ent[0]


_extra = {}


def extra_stream(device, k):
    "further streams of `device`, e.g. for decode_latents' slab pieces (taken from torch's pool as they come: not probed for their hardware queue)"
    key = (device.index, k)
    if key not in _extra:
        _extra[key] = torch.cuda.Stream(device=device)
    return _extra[key]


def side_stream(device):
    "the weight-gradient stream of `device` if one is in use (products launched since the last join), else None"
    ent = _side.get(device.index) if device.type == "cuda" else None
    return ent[0] if ent is not None and ent[1] else None


class _side_block:
    """`with _side_block(*operands) as on:` -- when `on`, everything launched (and allocated) inside goes to the second stream:
    weight-gradient work of the conv / deconv bottleneck that ends in an add into param.grad and that nobody waits for.
    Same rules as _side_begin (direct accumulation active, operands marked record_stream, joined before gradients are read)."""

    def __init__(self, *operands):
        self.operands, self.ctx = operands, None

    def __enter__(self):
        if not self.operands or os.environ.get("TVAE_CONV_WGRAD_STREAM", "1") != "1" or _side_begin(*self.operands) is None:
            return False
        self.ctx = torch.cuda.stream(_side[self.operands[0].device.index][0])
        self.ctx.__enter__()
        return True

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def _accumulate_direct(pairs):
    "param.grad += g for (param, g) pairs on the current stream, and report the parameters' gradients complete"
    for p, g_ in pairs:
        _sink[0].target(p).add_(g_.view_as(p))
    for p, _ in pairs:
        _sink[0].done(p)


def join_side():
    "the compute stream of every device waits for the weight-gradient products launched beside it"
    for idx, ent in _side.items():
        if ent[1]:
            torch.cuda.current_stream(idx).wait_stream(ent[0])
            ent[1] = 0


def _direct(*params):
    """True if every given parameter can take its gradient directly (leaf, requires grad, sink on)."""
    if _sink[0] is None:
        return False
    return all(p is not None and p.is_leaf and p.requires_grad and p.is_contiguous() for p in params)


def _note_use(*params):
    if _sink[0] is not None:
        for p in params:
            if p is not None and p.is_leaf and p.requires_grad:
                _sink[0].use(p)


def rows_buffer(M, N, device):
    """(M, N) fp32 activation buffer.  When a row is a multiple of 4 KB the rows of a 128-row GEMM
    tile all fall on the same few HBM channels (measured: 90 vs 107 TFLOP/s on the FFN w_1 launch),
    so such buffers get 128 floats of padding per row; every kernel takes explicit row strides."""
    if N % 1024 == 0:
        return torch.empty((M, N + 128), device=device, dtype=torch.float32)[:, :N]
    return torch.empty((M, N), device=device, dtype=torch.float32)


# ---- pre-split weights for the f32x3 arithmetic -------------------------------------------------
# In f32x3 mode the GEMM splits its fp32 operands into three bf16 planes on the way into LDS.  The
# weights only change once per optimizer step, so nn.Linear weights are split ONCE per step (both
# orientations in one launch: W for the forward, W^T for the data gradient) and handed to the kernel
# pre-split; the kernel then only splits the activation operand.
_weights_epoch = [0]


def weights_changed():
    "every cached split is stale (called after an optimizer step that wrote the weights through raw pointers)"
    _weights_epoch[0] += 1


def _split_weight(W, dtype=None):
    """(planes of W, planes of W^T) as int16 tensors in tvae_split3's tile-major layout (rows padded to a multiple of 64:
    (ceil64(N), 3K) and (ceil64(K), 3N) elements), or None if W does not qualify."""
    if (_state["dtype"] if dtype is None else dtype) != F32X3 or not isinstance(W, torch.nn.Parameter) or W.dim() != 2 or not W.is_contiguous():
        return None
    N, K = W.shape
    if N % 16 or K % 16:
        return None
    # the planes live ON the parameter object: a cache keyed by address would hand a new model that
    # reuses a freed address the old model's (possibly smaller) buffers
    stamp = (_weights_epoch[0], W._version, W.data_ptr(), N, K)
    hit = getattr(W, "_tvae_split3", None)
    if hit is not None and hit[0] == stamp:
        return hit[1], hit[2]
    pad = lambda n: (n + 63) // 64 * 64
    if hit is not None and hit[1].shape == (pad(N), 3 * K) and hit[1].device == W.device:
        wn, wt = hit[1], hit[2]                       # same shape: reuse the buffers
    else:
        wn = torch.empty((pad(N), 3 * K), device=W.device, dtype=torch.int16)      # the kernel writes the pad rows (zeros) too
        wt = torch.empty((pad(K), 3 * N), device=W.device, dtype=torch.int16)
    call("tvae_split3", _p(W), K, _p(wn), _p(wt), N, K, _stream())
    W._tvae_split3 = (stamp, wn, wt)
    return wn, wt


def _bf16_weight(W, dtype=None):
    """(bf16 copy of W (N, K), bf16 copy of W^T (K, N)) as int16 tensors for the streamed bf16 kernel, or None if W
    does not qualify.  Made once per optimizer step (tvae_cast_bf16), kept on the parameter object."""
    if (_state["dtype"] if dtype is None else dtype) != BF16 or not isinstance(W, torch.nn.Parameter) or W.dim() != 2 \
            or not W.is_contiguous() or os.environ.get("TVAE_BF16_STREAM", "1") != "1":
        return None
    N, K = W.shape
    if N % 8 or K % 8:
        return None
    stamp = (_weights_epoch[0], W._version, W.data_ptr(), N, K)
    hit = getattr(W, "_tvae_bf16", None)
    if hit is not None and hit[0] == stamp:
        return hit[1], hit[2]
    if hit is not None and hit[1].shape == (N, K) and hit[1].device == W.device:
        wn, wt = hit[1], hit[2]
    else:
        wn = torch.empty((N, K), device=W.device, dtype=torch.int16)
        wt = torch.empty((K, N), device=W.device, dtype=torch.int16)
    call("tvae_cast_bf16", _p(W), K, _p(wn), _p(wt), N, K, _stream())
    W._tvae_bf16 = (stamp, wn, wt)
    return wn, wt


def linear_fwd(x, W, b=None, relu=False, residual=None, drop=None, out=None):
    """y[M,N] = drop(relu(x W^T + b)) + residual.  x: (M,K) contiguous rows, W: (N,K)."""
    M, K = x.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    flags = GEMM_RELU if relu else 0
    sp = _split_weight(W) if x.stride(0) % 4 == 0 else None
    Bm, ldb = W, W.stride(0)
    if sp is not None:
        Bm, ldb, flags = sp[0], 3 * K, flags | GEMM_B_SPLIT3
    elif K % 32 == 0 and x.stride(0) % 4 == 0:
        bw = _bf16_weight(W)
        if bw is not None:                 # streamed bf16 kernel: the weight is bf16 in memory already
            Bm, ldb, flags = bw[0], K, flags | GEMM_B_BF16
    gemm(x, x.stride(0), Bm, ldb, out, out.stride(0), M, N, K, bias=b, residual=residual,
         ldr=0 if residual is None else residual.stride(0), flags=flags, drop=drop)
    return out


def linear_dgrad(dy, W, out=None, accumulate=False, relumask=None, alpha=1.0, a_drop=None):
    """dx[M,K] (+)= (dy * keepscale) W, optionally masked by (relumask > 0).  W: (N,K)."""
    M, N = dy.shape
    K = W.shape[1]
    if out is None:
        out = torch.empty((M, K), device=dy.device, dtype=torch.float32)
    flags = GEMM_ACCUM if accumulate else 0
    if a_drop is not None:
        flags |= GEMM_A_DROPMASK
    if relumask is not None:
        flags |= GEMM_RELUMASK
    dt = _role_dtype("dgrad")
    sp = _split_weight(W, dt) if dy.stride(0) % 4 == 0 else None
    bw = _bf16_weight(W, dt) if (sp is None and N % 32 == 0 and dy.stride(0) % 4 == 0) else None
    if bw is not None and a_drop is not None:          # the streamed kernel copies A as it is: mask the gradient first
        dy, a_drop = masked_grad(dy, a_drop, force=True)
        flags &= ~GEMM_A_DROPMASK
    if sp is not None:                     # W^T pre-split: the contraction index (out features) is contiguous
        Bm, ldb, flags = sp[1], 3 * N, flags | GEMM_B_SPLIT3
    elif bw is not None:                   # dY W as an NT product against the bf16 copy of W^T
        Bm, ldb, flags = bw[1], N, flags | GEMM_B_BF16
    else:
        Bm, ldb, flags = W, W.stride(0), flags | GEMM_B_KMAJOR
    gemm(dy, dy.stride(0), Bm, ldb, out, out.stride(0), M, K, N, residual=relumask,
         ldr=0 if relumask is None else relumask.stride(0), alpha=alpha, flags=flags, drop=a_drop, dtype=dt)
    return out


def linear_wgrad(dy, x, a_drop=None, want_bias=True, W=None, b=None, side=True):
    """dW[N,K] (+)= (dy*keepscale)^T x by split-K fp32 atomics, db[N] (+)= column sums of dy, the
    latter fused into the same launch.  With W (and b) given and direct accumulation active the
    sums go straight into W.grad / b.grad and (None, None) is returned; the product is then launched
    on the second stream (_side_begin) unless `side` is False: the caller must not write dy or x
    again before join_side()."""
    M, N = dy.shape
    K = x.shape[1]
    direct = _direct(W) and (not want_bias or _direct(b))
    if direct:
        dW = _sink[0].target(W)
        db = _sink[0].target(b) if want_bias else None
    else:
        dW = torch.zeros((N, K), device=dy.device, dtype=torch.float32)
        db = torch.zeros((N,), device=dy.device, dtype=torch.float32) if want_bias else None
    dt = _role_dtype("wgrad")
    if dt == F32X3 and a_drop is not None:         # the 3xbf16 kernel cannot regenerate the mask of a k-major operand
        dy, a_drop = masked_grad(dy, a_drop, force=True)
    flags = GEMM_A_KMAJOR | GEMM_B_KMAJOR | GEMM_SPLITK | (GEMM_A_DROPMASK if a_drop is not None else 0)
    gemm(dy, dy.stride(0), x, x.stride(0), dW, K, N, K, M, flags=flags, drop=a_drop, a_rowsum=db, dtype=dt,
         stream=_side_begin(dy, x) if direct and side else None)
    if direct:
        _sink[0].done(W)
        if want_bias:
            _sink[0].done(b)
        return None, None
    return dW, db


def colsum(X, drop=None):
    M, N = X.shape
    out = torch.zeros((N,), device=X.device, dtype=torch.float32)
    r, s, p = _drop_args(drop)
    call("tvae_colsum", _p(X), X.stride(0), _p(out), M, N, 1, r, s, p, _stream())
    return out


def layernorm_fwd(x2, a, b, eps=1e-6):
    y = torch.empty_like(x2)
    call("tvae_layernorm_fwd", _p(x2), _p(a), _p(b), _p(y), x2.shape[0], x2.shape[1], eps, _stream())
    return y


# ---- residual-dropout masks applied where the gradient is PRODUCED -----------------------------------------------
# The gradient arriving at the output of a SublayerConnection, y = x + dropout(sublayer(norm(x))), is needed twice with that
# dropout's mask on it (data gradient and weight gradient of the sublayer's last projection).  It is written by the
# LayerNorm backward of the NEXT sublayer (pre-norm: that LayerNorm reads y), so that kernel also writes the masked copy
# (tvae_layernorm_bwd_masked: one more store stream) instead of both GEMMs regenerating the mask in their k-loops.
#   _res_drop   output buffer of a sublayer (data_ptr) -> its residual Drop, noted by the forward pass
#   _premasked  gradient buffer (data_ptr) -> (masked copy, site, the gradient tensor itself: keeps the address taken)
# A sublayer only uses a masked copy made for exactly the tensor it receives and for its own site; if autograd summed
# several gradients (an output with two consumers) the tensor is another one and the GEMMs mask while staging, as before.
_res_drop, _premasked = {}, {}
premask_stats = {"written": 0, "used": 0}


def begin_step():
    "forget the notes of earlier passes (VAEShell.train_step calls this before a forward pass)"
    _res_drop.clear()
    _premasked.clear()


def _note_res_drop(y2, drop):
    # (grad mode is off inside Function.forward, so it cannot be asked here; a Drop only exists in training mode)
    if drop is not None and os.environ.get("TVAE_PREMASK", "1") == "1":
        if len(_res_drop) > 256:
            _res_drop.clear()
        _res_drop[y2.data_ptr()] = drop


def _take_premasked(dy2, drop):
    hit = _premasked.pop(dy2.data_ptr(), None)
    if hit is not None and drop is not None and hit[1] == drop.site and hit[0].shape == dy2.shape and hit[2] is not None:
        premask_stats["used"] += 1
        return hit[0]
    return None


def layernorm_bwd(x2, a, dy, add_to=None, eps=1e-6, a_param=None, b_param=None, up_drop=None):
    dx = torch.empty_like(x2)
    direct = _direct(a_param, b_param)
    if direct:
        da, db = _sink[0].target(a_param), _sink[0].target(b_param)
    else:
        dab = torch.zeros((2, x2.shape[1]), device=x2.device, dtype=torch.float32)
        da, db = dab[0], dab[1]
    if up_drop is not None and x2.shape[0] * x2.shape[1] < (1 << 32):
        dxm = torch.empty_like(x2)
        r, s, p = _drop_args(up_drop)
        call("tvae_layernorm_bwd_masked", _p(x2), _p(a), _p(dy), _p(add_to), _p(dx), _p(dxm), _p(da), _p(db),
             x2.shape[0], x2.shape[1], eps, r, s, p, _stream())
        if len(_premasked) > 8:
            _premasked.clear()
        _premasked[dx.data_ptr()] = (dxm, up_drop.site, dx)
        premask_stats["written"] += 1
    else:
        call("tvae_layernorm_bwd", _p(x2), _p(a), _p(dy), _p(add_to), _p(dx), _p(da), _p(db),
             x2.shape[0], x2.shape[1], eps, _stream())
    if direct:
        _sink[0].done(a_param)
        _sink[0].done(b_param)
        return dx, None, None
    return dx, da, db


class MaskSpec:
    """Attention mask in kernel form: key_mask (B,Lk) uint8 and/or causal flag and/or a full
    (B,Lq,Lk) uint8 mask.  Built by ``mask_spec`` from the reference's bool/float masks."""
    __slots__ = ("key_mask", "causal", "full")

    def __init__(self, key_mask=None, causal=False, full=None):
        self.key_mask, self.causal, self.full = key_mask, bool(causal), full


def tag_mask(mask, spec):
    """Attach the kernel form to a reference-style mask tensor (so callers keep the tensor API)."""
    mask._tvae_spec = spec
    return mask


def mask_spec(mask, B, Lq, Lk, device=None):
    """Kernel form of a reference-style mask ((B|1, 1, Lk) or (B|1, Lq, Lk), zero = masked),
    moved to `device` if it lives elsewhere (the reference builds decode masks on the host)."""
    if mask is None:
        return MaskSpec()
    if isinstance(mask, MaskSpec):
        return mask
    spec = getattr(mask, "_tvae_spec", None)
    if spec is not None and (device is None or all(t is None or t.device == device for t in (spec.key_mask, spec.full))):
        return spec
    m = mask if device is None else mask.to(device)
    if m.dim() == 4:                    # already unsqueezed for heads
        m = m[:, 0]
    if m.dim() != 3:
        raise _lib.TvaeError("attention mask must be (B,1,Lk) or (B,Lq,Lk), got %s" % (tuple(mask.shape),))
    nz = (m != 0)
    if m.shape[1] == 1:
        km = nz[:, 0, :].expand(B, Lk).to(torch.uint8).contiguous()
        return MaskSpec(key_mask=km)
    return MaskSpec(full=nz.expand(B, Lq, Lk).to(torch.uint8).contiguous())


def attn_fwd(q, k, v, B, h, Lq, Lk, dk, spec, drop=None, want_p=False):
    """q: (B*Lq, >=h*dk) row views (stride(0) = ld); returns o (B*Lq, h*dk), lse, p."""
    o = torch.empty((B * Lq, h * dk), device=q.device, dtype=torch.float32)
    lse = torch.empty((B, h, Lq, 2), device=q.device, dtype=torch.float32)
    p_out = torch.empty((B, h, Lq, Lk), device=q.device, dtype=torch.float32) if want_p else None
    r, s, p = _drop_args(drop)
    call("tvae_attn_fwd", _p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
         _p(lse), _p(p_out), _p(spec.key_mask), _p(spec.full), int(spec.causal), B, h, Lq, Lk, dk,
         1.0 / math.sqrt(dk), r, s, p, _state["dtype"], _stream())
    return o, lse, p_out


def attn_bwd(q, k, v, o, do, lse, B, h, Lq, Lk, dk, spec, drop=None, out=None):
    """out = (dq, dk, dv) to write into given (possibly column-strided) buffers"""
    if out is not None:
        dq, dkk, dv = out
    else:
        dq = torch.empty((B * Lq, h * dk), device=q.device, dtype=torch.float32)
        dkk = torch.empty((B * Lk, h * dk), device=q.device, dtype=torch.float32)
        dv = torch.empty((B * Lk, h * dk), device=q.device, dtype=torch.float32)
    rows = B * h * Lq                                   # workspace: dsum per row + 4 dropout keep words per row + 1 word per (b, head)
    dsum = torch.empty(((rows + 3) // 4 * 4 + 4 * rows + B * h,), device=q.device, dtype=torch.float32)
    r, s, p = _drop_args(drop)
    call("tvae_attn_bwd", _p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
         _p(do), do.stride(0), _p(lse), _p(dq), dq.stride(0), _p(dkk), dkk.stride(0), _p(dv), dv.stride(0),
         _p(dsum), _p(spec.key_mask), _p(spec.full), int(spec.causal), B, h, Lq, Lk, dk,
         1.0 / math.sqrt(dk), r, s, p, _state["dtype"], _stream())
    return dq, dkk, dv


# ----------------------------------------------------------------------------- autograd blocks
def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a, b, eps):
        x2 = _c(x).view(-1, x.shape[-1])
        _chk(x2, "x")
        y = layernorm_fwd(x2, a, b, eps)
        ctx.save_for_backward(x2, a)
        ctx.eps, ctx.P = eps, (a, b)
        ctx.up_drop = _res_drop.get(x2.data_ptr())        # the norm after a stack of layers reads the last sublayer's output
        _note_use(a, b)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, a = ctx.saved_tensors
        dx, da, db = layernorm_bwd(x2, a, _c(dy).view(x2.shape), None, ctx.eps, *ctx.P, up_drop=ctx.up_drop)
        return dx.view(dy.shape), da, db, None


def relu_mask(dy2, y):
    """dy * [y > 0] in one pass (the stride-1 form of the upsample-backward kernel)."""
    out = torch.empty_like(dy2)
    M, N = dy2.shape
    call("tvae_upsample_bwd", _p(dy2), _p(y), _p(out), M, 0, 1, M, N, 1, 0, M, _stream())
    return out


class LinearFn(torch.autograd.Function):
    """y = [relu](x W^T + b): z_means / z_var / predict_len / decoder.linear / generator /
    property-predictor layers (reference trans_models.py:747, 790-793, 852, 1030)."""

    @staticmethod
    def forward(ctx, x, W, b, relu):
        x2 = _c(x).view(-1, x.shape[-1])
        _chk(x2, "x")
        y = linear_fwd(x2, W, b, relu=relu)
        ctx.save_for_backward(x2, W, y if relu else None)
        ctx.relu, ctx.has_bias, ctx.xshape, ctx.P = relu, b is not None, x.shape, (W, b)
        _note_use(W, b)
        return y.view(x.shape[:-1] + (W.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        x2, W, y = ctx.saved_tensors
        dy2 = _c(dy).view(-1, W.shape[0])
        if ctx.relu:
            dy2 = relu_mask(dy2, y)
        # the parameter object itself (not the unpacked saved tensor) carries the pre-split planes of f32x3 mode
        dx = linear_dgrad(dy2, ctx.P[0]).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        dW, db = linear_wgrad(dy2, x2, want_bias=ctx.has_bias, W=ctx.P[0], b=ctx.P[1])
        return dx, dW, db, None


class EmbedFn(torch.autograd.Function):
    """Embeddings * sqrt(d) + PositionalEncoding (+dropout), trans_models.py:1042-1043, 1060-1063."""

    @staticmethod
    def forward(ctx, ids, lut, pe, drop):
        B, L = ids.shape
        vocab, d = lut.shape
        ids = _c(ids)
        out = torch.empty((B, L, d), device=lut.device, dtype=torch.float32)
        r, s, p = _drop_args(drop)
        pe2 = pe.view(-1, d)
        if L > pe2.shape[0]:
            raise _lib.TvaeError("sequence length %d exceeds positional table %d" % (L, pe2.shape[0]))
        call("tvae_embed_fwd", _p(ids), _p(lut), _p(pe2), _p(out), B, L, d, vocab, r, s, p, _stream())
        ctx.save_for_backward(ids)
        ctx.drop, ctx.shape, ctx.P = drop, (vocab, d), lut
        _note_use(lut)
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        vocab, d = ctx.shape
        B, L = ids.shape
        direct = _direct(ctx.P)
        dlut = _sink[0].target(ctx.P) if direct else torch.zeros((vocab, d), device=dout.device, dtype=torch.float32)
        r, s, p = _drop_args(ctx.drop)
        dout = _c(dout)
        with _side_block(dout, ids) if direct else _side_block():      # direct: atomics into lut.grad, nobody waits for them (ids: a temporary of the forward pass, read there too)
            call("tvae_embed_bwd", _p(ids), _p(dout), _p(dlut), B, L, d, vocab, r, s, p, _stream())
        if direct:
            _sink[0].done(ctx.P)
            return None, None, None, None
        return None, dlut, None, None


class AttnCoreFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(dk), masked) v on (B, L, h*dk) tensors (heads are column blocks)."""

    @staticmethod
    def forward(ctx, q, k, v, spec, h, drop):
        B, Lq, d = q.shape
        Lk = k.shape[1]
        dk = d // h
        q2, k2, v2 = _c(q).view(-1, d), _c(k).view(-1, d), _c(v).view(-1, d)
        o, lse, _ = attn_fwd(q2, k2, v2, B, h, Lq, Lk, dk, spec, drop)
        ctx.save_for_backward(q2, k2, v2, o, lse)
        ctx.meta = (B, h, Lq, Lk, dk, spec, drop)
        return o.view(B, Lq, d)

    @staticmethod
    def backward(ctx, do):
        q2, k2, v2, o, lse = ctx.saved_tensors
        B, h, Lq, Lk, dk, spec, drop = ctx.meta
        dq, dkk, dv = attn_bwd(q2, k2, v2, o, _c(do).view(-1, h * dk), lse, B, h, Lq, Lk, dk, spec, drop)
        return dq.view(B, Lq, -1), dkk.view(B, Lk, -1), dv.view(B, Lk, -1), None, None, None


def attention_weights(q, k, spec, h):
    """P = softmax(mask(q k^T/sqrt(dk))) as (B, h, Lq, Lk) -- no gradient (forward_w_attn path)."""
    B, Lq, d = q.shape
    Lk = k.shape[1]
    q2, k2 = _c(q).view(-1, d), _c(k).view(-1, d)
    _, _, p = attn_fwd(q2, k2, k2, B, h, Lq, Lk, d // h, spec, None, want_p=True)
    return p


def _adjacent(*ts):
    """True if the tensors are consecutive row blocks of one contiguous fp32 matrix in one storage
    (MultiHeadedAttention.pack_qkv lays the q, k, v projections out like that)."""
    t0 = ts[0]
    ptr, st = t0.data_ptr(), t0.untyped_storage().data_ptr()
    for t in ts:
        if (t is None or t.dtype != torch.float32 or not t.is_contiguous() or t.shape[1:] != t0.shape[1:]
                or t.data_ptr() != ptr or t.untyped_storage().data_ptr() != st):
            return False
        ptr += t.numel() * 4
    return True


def _run(ts):
    "one tensor over adjacent row blocks (see _adjacent)"
    t0 = ts[0].detach()
    return t0.as_strided((sum(t.shape[0] for t in ts),) + tuple(t0.shape[1:]), t0.stride(), t0.storage_offset())


def _run_param(ts):
    """The adjacent row blocks `ts` as ONE nn.Parameter over the same memory, cached on the first block: linear_fwd /
    linear_dgrad key their f32x3 planes and bf16 copies of a weight on the Parameter object."""
    t0 = ts[0]
    key = tuple(t.data_ptr() for t in ts)
    hit = getattr(t0, "_tvae_run", None)
    if hit is None or hit[0] != key:
        hit = (key, torch.nn.Parameter(_run(ts), requires_grad=False))
        t0._tvae_run = hit
    return hit[1]


def _proj_fused(x_rows, Ws, bs):
    """[x W_0^T + b_0 | x W_1^T + b_1 | ...] as ONE product against the row-stacked weights (which must be _adjacent):
    a 63500-token d_model x d_model projection is one round of 994 tiles -- nothing overlaps its prologue and epilogue --
    and runs at ~95 TFLOP/s in fp32 where the same arithmetic as one 768-wide product reaches ~115."""
    return linear_fwd(x_rows, _run_param(Ws), _run(bs))


def _proj_fused_bwd(dcat, x_rows, Ws, bs, accumulate_into=None):
    """gradients of _proj_fused: dx = dcat [W_0; W_1; ...] in one product (K = the stacked width), and dW / db in one split-K
    launch when the parameters' gradients are adjacent too (else one launch per block on column views of dcat)."""
    M, N = dcat.shape
    K = Ws[0].shape[1]
    direct = all(_direct(w) for w in Ws) and all(_direct(b) for b in bs)
    grads = None
    if direct:
        gW, gb = [_sink[0].target(w) for w in Ws], [_sink[0].target(b) for b in bs]
        if _adjacent(*gW) and _adjacent(*gb):
            gemm(dcat, N, x_rows, x_rows.stride(0), _run(gW), K, N, K, M, flags=GEMM_A_KMAJOR | GEMM_B_KMAJOR | GEMM_SPLITK,
                 a_rowsum=_run(gb), dtype=_role_dtype("wgrad"), stream=_side_begin(dcat, x_rows))
            for t in list(Ws) + list(bs):
                _sink[0].done(t)
            grads = [(None, None)] * len(Ws)
    if grads is None:
        grads, c = [], 0
        for w, b in zip(Ws, bs):
            grads.append(linear_wgrad(dcat[:, c:c + w.shape[0]], x_rows, W=w, b=b))
            c += w.shape[0]
    dx = linear_dgrad(dcat, _run_param(Ws), out=accumulate_into, accumulate=accumulate_into is not None)
    return dx, grads


class AttnSublayerFn(torch.autograd.Function):
    """One SublayerConnection around MultiHeadedAttention, fused end to end:
        y = x + dropout( W_o attn( W_q LN(x), W_k m, W_v m ) + b_o ),   m = LN(x) or `mem`
    (reference trans_models.py:1099-1101 around :921-941).  Saves x, LN(x), q, k, v, attn-out, lse.
    """

    @staticmethod
    def forward(ctx, x, mem, ln_a, ln_b, Wq, bq, Wk, bk, Wv, bv, Wo, bo, spec, h, drop_attn, drop_res, eps):
        B, Lq, d = x.shape
        x2 = _c(x).view(-1, d)
        _chk(x2, "x")
        n = layernorm_fwd(x2, ln_a, ln_b, eps)
        if mem is None:
            kv_in, Lk = n, Lq
        else:
            kv_in, Lk = _c(mem).view(-1, d), mem.shape[1]
        dk = d // h
        # the projections laid out as row blocks of one matrix (pack_qkv): q | k | v (self-attention)
        # or k | v (source attention) from ONE product; the attention kernels take column views of its output
        fuse = 0
        if _adjacent(Wk, Wv) and _adjacent(bk, bv):
            fuse = 3 if (mem is None and _adjacent(Wq, Wk) and _adjacent(bq, bk)) else 2
        if fuse == 3:
            qkv = _proj_fused(n, (Wq, Wk, Wv), (bq, bk, bv))
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        elif fuse == 2:
            q = linear_fwd(n, Wq, bq)
            kv = _proj_fused(kv_in, (Wk, Wv), (bk, bv))
            k, v = kv[:, :d], kv[:, d:]
        else:
            q = linear_fwd(n, Wq, bq)
            k = linear_fwd(kv_in, Wk, bk)
            v = linear_fwd(kv_in, Wv, bv)
        o, lse, _ = attn_fwd(q, k, v, B, h, Lq, Lk, dk, spec, drop_attn)
        y = linear_fwd(o, Wo, bo, residual=x2, drop=drop_res)
        ctx.up_drop = _res_drop.get(x2.data_ptr())        # x is the output of the sublayer in front: its residual dropout
        _note_res_drop(y, drop_res)
        ctx.save_for_backward(x2, n, kv_in if mem is not None else None, q, k, v, o, lse, ln_a, Wq, Wk, Wv, Wo)
        ctx.meta = (B, h, Lq, Lk, dk, spec, drop_attn, drop_res, eps, mem is not None)
        ctx.fuse = fuse
        ctx.P = (ln_a, ln_b, Wq, bq, Wk, bk, Wv, bv, Wo, bo)
        _note_use(*ctx.P)
        return y.view(B, Lq, d)

    @staticmethod
    def backward(ctx, dy):
        x2, n, memv, q, k, v, o, lse, ln_a, Wq, Wk, Wv, Wo = ctx.saved_tensors
        B, h, Lq, Lk, dk, spec, drop_attn, drop_res, eps, cross = ctx.meta
        d = h * dk
        dy2 = _c(dy).view(-1, d)
        pa, pb, pWq, pbq, pWk, pbk, pWv, pbv, pWo, pbo = ctx.P
        Wq, Wk, Wv, Wo = pWq, pWk, pWv, pWo       # the parameter objects carry the pre-split planes of f32x3 mode
        dym, a_drop = masked_grad(dy2, drop_res)
        do = linear_dgrad(dym, Wo, a_drop=a_drop)
        dWo, dbo = linear_wgrad(dym, o, a_drop=a_drop, W=pWo, b=pbo)
        kv_in = memv if cross else n
        fuse = ctx.fuse
        if fuse == 3 and _adjacent(pWq, pWk, pWv) and _adjacent(pbq, pbk, pbv):
            dqkv = torch.empty((q.shape[0], 3 * d), device=q.device, dtype=torch.float32)
            attn_bwd(q, k, v, o, do, lse, B, h, Lq, Lk, dk, spec, drop_attn, out=(dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]))
            dn, ((dWq, dbq), (dWk, dbk), (dWv, dbv)) = _proj_fused_bwd(dqkv, n, (pWq, pWk, pWv), (pbq, pbk, pbv))
            dx, da, db = layernorm_bwd(x2, ln_a, dn, add_to=dy2, eps=eps, a_param=pa, b_param=pb, up_drop=ctx.up_drop)
            return (dx.view(B, Lq, d), None, da, db, dWq, dbq, dWk, dbk, dWv, dbv, dWo, dbo,
                    None, None, None, None, None)
        if fuse == 2 and _adjacent(pWk, pWv) and _adjacent(pbk, pbv):
            dq = torch.empty((q.shape[0], d), device=q.device, dtype=torch.float32)
            dkv = torch.empty((k.shape[0], 2 * d), device=q.device, dtype=torch.float32)
            attn_bwd(q, k, v, o, do, lse, B, h, Lq, Lk, dk, spec, drop_attn, out=(dq, dkv[:, :d], dkv[:, d:]))
            dWq, dbq = linear_wgrad(dq, n, W=pWq, b=pbq)
            dn = linear_dgrad(dq, Wq)
            if cross:
                dmem, ((dWk, dbk), (dWv, dbv)) = _proj_fused_bwd(dkv, kv_in, (pWk, pWv), (pbk, pbv))
                dmem = dmem.view(B, Lk, d)
            else:
                _, ((dWk, dbk), (dWv, dbv)) = _proj_fused_bwd(dkv, kv_in, (pWk, pWv), (pbk, pbv), accumulate_into=dn)
                dmem = None
            dx, da, db = layernorm_bwd(x2, ln_a, dn, add_to=dy2, eps=eps, a_param=pa, b_param=pb, up_drop=ctx.up_drop)
            return (dx.view(B, Lq, d), dmem, da, db, dWq, dbq, dWk, dbk, dWv, dbv, dWo, dbo,
                    None, None, None, None, None)
        dq, dkk, dv = attn_bwd(q, k, v, o, do, lse, B, h, Lq, Lk, dk, spec, drop_attn)
        dWq, dbq = linear_wgrad(dq, n, W=pWq, b=pbq)
        dWk, dbk = linear_wgrad(dkk, kv_in, W=pWk, b=pbk)
        dWv, dbv = linear_wgrad(dv, kv_in, W=pWv, b=pbv)
        dn = linear_dgrad(dq, Wq)
        if cross:
            dmem = linear_dgrad(dkk, Wk)
            linear_dgrad(dv, Wv, out=dmem, accumulate=True)
            dmem = dmem.view(B, Lk, d)
        else:
            linear_dgrad(dkk, Wk, out=dn, accumulate=True)
            linear_dgrad(dv, Wv, out=dn, accumulate=True)
            dmem = None
        dx, da, db = layernorm_bwd(x2, ln_a, dn, add_to=dy2, eps=eps, a_param=pa, b_param=pb, up_drop=ctx.up_drop)
        return (dx.view(B, Lq, d), dmem, da, db, dWq, dbq, dWk, dbk, dWv, dbv, dWo, dbo,
                None, None, None, None, None)


class FFNSublayerFn(torch.autograd.Function):
    """y = x + dropout( W_2 dropout(relu(W_1 LN(x) + b_1)) + b_2 )  (trans_models.py:951-952, 1101)."""

    @staticmethod
    def forward(ctx, x, ln_a, ln_b, W1, b1, W2, b2, drop_ff, drop_res, eps):
        shp = x.shape
        x2 = _c(x).view(-1, shp[-1])
        _chk(x2, "x")
        n = layernorm_fwd(x2, ln_a, ln_b, eps)
        hmid = linear_fwd(n, W1, b1, relu=True, drop=drop_ff, out=rows_buffer(n.shape[0], W1.shape[0], n.device))
        y = linear_fwd(hmid, W2, b2, residual=x2, drop=drop_res)
        ctx.up_drop = _res_drop.get(x2.data_ptr())
        _note_res_drop(y, drop_res)
        ctx.save_for_backward(x2, n, hmid, ln_a, W1, W2)
        ctx.meta = (drop_ff, drop_res, eps, shp)
        ctx.P = (ln_a, ln_b, W1, b1, W2, b2)
        _note_use(*ctx.P)
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, n, hmid, ln_a, W1, W2 = ctx.saved_tensors
        drop_ff, drop_res, eps, shp = ctx.meta
        dy2 = _c(dy).view(x2.shape)
        pa, pb, pW1, pb1, pW2, pb2 = ctx.P
        W1, W2 = pW1, pW2                          # the parameter objects carry the pre-split planes of f32x3 mode
        dym, a_drop = masked_grad(dy2, drop_res)
        dW2, db2 = linear_wgrad(dym, hmid, a_drop=a_drop, W=pW2, b=pb2)
        # hmid = keepscale * relu(pre): hmid > 0 <=> kept and pre > 0, so one mask + one scale
        sc = 1.0 / (1.0 - drop_ff.p) if drop_ff is not None else 1.0
        dh = linear_dgrad(dym, W2, relumask=hmid, alpha=sc, a_drop=a_drop, out=rows_buffer(*hmid.shape, hmid.device))
        dW1, db1 = linear_wgrad(dh, n, W=pW1, b=pb1)
        dn = linear_dgrad(dh, W1)
        dx, da, db = layernorm_bwd(x2, ln_a, dn, add_to=dy2, eps=eps, a_param=pa, b_param=pb, up_drop=ctx.up_drop)
        return dx.view(shp), da, db, dW1, db1, dW2, db2, None, None, None


class ReparamFn(torch.autograd.Function):
    """z = mu + eps * eps_scale * exp(0.5 logvar)  (trans_models.py:771-775)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps, eps_scale):
        mu, logvar, eps = _c(mu), _c(logvar), _c(eps)
        z = torch.empty_like(mu)
        call("tvae_reparam_fwd", _p(mu), _p(logvar), _p(eps), _p(z), mu.numel(), float(eps_scale), _stream())
        ctx.save_for_backward(logvar, eps)
        ctx.eps_scale = float(eps_scale)
        return z

    @staticmethod
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dmu = torch.zeros_like(logvar)
        dlv = torch.zeros_like(logvar)
        call("tvae_reparam_bwd", _p(_c(dz)), _p(logvar), _p(eps), _p(dmu), _p(dlv), logvar.numel(),
             ctx.eps_scale, _stream())
        return dmu, dlv, None, None


# ----------------------------------------------------------------------------- conv / deconv
def _pack_w(w, rows, inner, k, tf):
    """(rows, inner*k)-shaped GEMM weight from a conv kernel tensor, see tvae_conv_w_pack."""
    wp = torch.empty((rows, k * inner), device=w.device, dtype=torch.float32)
    call("tvae_conv_w_pack", _p(w), _p(wp), rows, inner, k, int(tf), _stream())
    return wp


def _unpack_dw(dwp, shape, Cout, Cin, k, tf, transposed=False):
    dw = torch.empty(shape, device=dwp.device, dtype=torch.float32)
    call("tvae_conv_w_unpack_grad", _p(dwp), _p(dw), Cout, Cin, k, int(tf) | (2 if transposed else 0), _stream())
    return dw


def _conv_gemm(xin, rows_total, Cin, wp, bias, k, relu):
    """out[r, :] = act(sum_j x[r+j, :] . wp[:, j*Cin:(j+1)*Cin] + bias) for r < rows_total-(k-1);
    xin is the flat (rows_total, Cin) token-major buffer (im2col rows overlap: lda = Cin)."""
    Cout = wp.shape[0]
    out = torch.empty((rows_total, Cout), device=xin.device, dtype=torch.float32)
    gemm(xin, Cin, wp, k * Cin, out, Cout, rows_total - (k - 1), Cout, k * Cin, bias=bias,
         flags=GEMM_RELU if relu else 0)
    return out


def _conv_wgrad(dy_rows, xin, rows_total, Cin, Cout, k):
    """dwp[co, j*Cin+ci] = sum_r dy[r, co] x[r+j, ci] over r < rows_total-(k-1) (split-K atomics).  Returns
    (dwp, db, transposed): with 128 < Cout <= 160 the operands are swapped -- dwp comes out as (k*Cin, Cout), so that
    Cout lies along N on tvae_gemm's 160-wide tile instead of along M on two 128-row tiles, the second a quarter full."""
    rows = rows_total - (k - 1)
    dt = _role_dtype("wgrad")
    if 128 < Cout <= 160 and dt != BF16 and Cin % 4 == 0 and Cout % 4 == 0:
        dwt = torch.zeros((k * Cin, Cout), device=xin.device, dtype=torch.float32)
        gemm(xin, Cin, dy_rows, Cout, dwt, Cout, k * Cin, Cout, rows, flags=GEMM_A_KMAJOR | GEMM_B_KMAJOR | GEMM_SPLITK, dtype=dt)
        return dwt, colsum(dy_rows[:rows]), True
    dwp = torch.zeros((Cout, k * Cin), device=xin.device, dtype=torch.float32)
    db = torch.zeros((Cout,), device=xin.device, dtype=torch.float32)     # bias grad, fused row sums
    gemm(dy_rows, Cout, xin, Cin, dwp, k * Cin, Cout, k * Cin, rows,
         flags=GEMM_A_KMAJOR | GEMM_B_KMAJOR | GEMM_SPLITK, a_rowsum=db, dtype=dt)
    return dwp, db, False


def _conv_dgrad(dy_padded, rows_total, Cout, w_dgrad, Cin, k):
    """dx[u, ci] = sum_j dy[u-j, :] . w[:, ci, j]: a conv-as-GEMM over the gradient rows.
    dy_padded is ((k-1) + rows_total, Cout) with k-1 leading zero rows; w_dgrad is (Cin, k*Cout)."""
    dx = torch.empty((rows_total, Cin), device=dy_padded.device, dtype=torch.float32)
    gemm(dy_padded, Cout, w_dgrad, k * Cout, dx, Cin, rows_total, Cin, k * Cout, dtype=_role_dtype("dgrad"))
    return dx


class ConvBottleneckFn(torch.autograd.Function):
    """ConvBottleneck (trans_models.py:976-983): 3 x relu(MaxPool1d(2)(Conv1d)), token-major.
    Input (B, L, C0); output (B, 64*9) in the reference's channel-major flattening (:789)."""

    @staticmethod
    def forward(ctx, x, W0, b0, W1, b1, W2, b2):
        B, L, C = x.shape
        cur = _c(x).view(B * L, C)
        _chk(cur, "x")
        saved, metas = [], []
        for li, (W, b) in enumerate(((W0, b0), (W1, b1), (W2, b2))):
            Cout, Cin, k = W.shape
            wp = _pack_w(W, Cout, Cin, k, False)
            conv = _conv_gemm(cur, B * L, Cin, wp, b, k, relu=False)
            Lconv = L - k + 1
            Lp = Lconv // 2
            last = li == 2
            y = torch.empty((B, Cout, Lp) if last else (B * Lp, Cout), device=x.device, dtype=torch.float32)
            call("tvae_pool_relu_fwd", _p(conv), _p(y), B, L, Lconv, Cout, int(last), _stream())
            saved += [cur, conv]
            metas.append((L, Lconv, Lp, Cin, Cout, k))
            cur, L = y, Lp
        ctx.save_for_backward(W0, W1, W2, *saved)
        ctx.metas, ctx.B = metas, B
        ctx.P = (W0, b0, W1, b1, W2, b2)
        _note_use(*ctx.P)
        return cur.view(B, -1)

    @staticmethod
    def backward(ctx, dout):
        W = ctx.saved_tensors[:3]
        saved = ctx.saved_tensors[3:]
        B = ctx.B
        dy = _c(dout)
        grads = [None] * 6
        for li in (2, 1, 0):
            L, Lconv, Lp, Cin, Cout, k = ctx.metas[li]
            xin, conv = saved[2 * li], saved[2 * li + 1]
            # only the k-1 leading rows need zeroing: tvae_pool_relu_bwd writes every one of the B*L rows behind them
            buf = torch.empty(((k - 1) + B * L, Cout), device=dy.device, dtype=torch.float32)
            buf[:k - 1].zero_()
            dconv = buf[k - 1:]
            call("tvae_pool_relu_bwd", _p(conv), _p(dy), _p(dconv), B, L, Lconv, Cout, int(li == 2), _stream())
            pW, pb = ctx.P[2 * li], ctx.P[2 * li + 1]
            if _direct(pW, pb):          # nobody waits for these: weight gradient -> unpack -> add into param.grad beside the rest
                with _side_block(dconv, xin):
                    dwp, db_, tr = _conv_wgrad(dconv, xin, B * L, Cin, Cout, k)
                    _accumulate_direct(((pW, _unpack_dw(dwp, W[li].shape, Cout, Cin, k, False, tr)), (pb, db_)))
            else:
                dwp, grads[2 * li + 1], tr = _conv_wgrad(dconv, xin, B * L, Cin, Cout, k)
                grads[2 * li] = _unpack_dw(dwp, W[li].shape, Cout, Cin, k, False, tr)
            if li > 0 or ctx.needs_input_grad[0]:
                wd = _pack_w(W[li], Cin, Cout, k, True)       # (Cin, k*Cout): wd[ci][j*Cout+co] = W[co][ci][k-1-j]
                dy = _conv_dgrad(buf, B * L, Cout, wd, Cin, k)
        dx = dy.view(B, ctx.metas[0][0], ctx.metas[0][3]) if ctx.needs_input_grad[0] else None
        return (dx,) + tuple(grads)


def _poly_pack(W, b, s, mode):
    """GEMM weight (and bias) of the polyphase form of a strided ConvTranspose1d, see tvae_deconv_poly_pack."""
    Cin, Cout, k = W.shape
    Q = -(-k // s)
    shape = (s * Cout, Q * Cin) if mode == 0 else (Cin, Q * s * Cout)
    wp = torch.empty(shape, device=W.device, dtype=torch.float32)
    bp = torch.empty((s * Cout,), device=W.device, dtype=torch.float32) if (mode == 0 and b is not None) else None
    call("tvae_deconv_poly_pack", _p(W), _p(wp), _p(b if bp is not None else None), _p(bp), Cin, Cout, k, s, mode, _stream())
    return wp, bp


class DeconvBottleneckFn(torch.autograd.Function):
    """DeconvBottleneck (trans_models.py:1001-1009): 3 x relu(ConvTranspose1d(k=11, stride, pad 2)).
    Input (B, 576) = the (B,64,9) view of relu(decoder.linear(z)) (:852-853); output (B, 127, d).

    Strided layers (4, 3) run as POLYPHASE products over the un-upsampled input (include/tvae_kernels.h,
    tvae_deconv_poly_pack): one GEMM with K = ceil(k/s) Cin and N = s Cout whose output rows are the token-major output --
    the FLOPs of the transposed convolution itself, where the zero-stuffed form (still used for stride 1, where it stuffs
    nothing) multiplied s - 1 zero rows out of every s.  TVAE_DECONV_POLY=0 restores the zero-stuffed form everywhere."""
    STRIDES = (4, 3, 1)
    PAD = 2

    @staticmethod
    def forward(ctx, hlin, W0, b0, W1, b1, W2, b2):
        B = hlin.shape[0]
        src = _c(hlin)
        _chk(src, "hlin")
        Lin, src_rows, src_cm = hlin.shape[1] // W0.shape[0], 0, 1
        saved, metas = [], []
        poly_on = os.environ.get("TVAE_DECONV_POLY", "1") == "1"
        PAD = DeconvBottleneckFn.PAD
        for li, (W, b) in enumerate(((W0, b0), (W1, b1), (W2, b2))):
            Cin, Cout, k = W.shape
            s = DeconvBottleneckFn.STRIDES[li]
            Lout = (Lin - 1) * s - 2 * PAD + k
            if poly_on and s > 1 and PAD < s:
                # rows per molecule: Lm phase groups m (output position t = m s + r - PAD), Q - 1 zero rows in front
                Q = -(-k // s)
                Lm = ((Lin - 1) * s - PAD + k - 1) // s + 1
                Lp = Lm + Q - 1
                xp = torch.empty((B * Lp, Cin), device=src.device, dtype=torch.float32)
                call("tvae_upsample_fwd", _p(src), src_rows, src_cm, _p(xp), B, Lin, Cin, 1, Q - 1, Lp, _stream())
                wp, bp = _poly_pack(W, b, s, 0)
                y = torch.empty((B * Lp * s, Cout), device=src.device, dtype=torch.float32)      # = (B*Lp, s*Cout)
                gemm(xp, Cin, wp, Q * Cin, y, s * Cout, B * Lp - (Q - 1), s * Cout, Q * Cin, bias=bp, flags=GEMM_RELU)
                saved += [xp, y]
                metas.append(("poly", Lin, Lout, Lp, Cin, Cout, k, s, Q))
                src, src_rows, src_cm = y[PAD:], Lp * s, 0           # output position t of molecule b: row b Lp s + PAD + t
            else:
                lpad = k - 1 - PAD
                Lup = Lout + k - 1
                up = torch.empty((B * Lup, Cin), device=src.device, dtype=torch.float32)
                call("tvae_upsample_fwd", _p(src), src_rows, src_cm, _p(up), B, Lin, Cin, s, lpad, Lup, _stream())
                wp = _pack_w(W, Cout, Cin, k, True)                          # flipped taps, (Cout, k*Cin)
                y = _conv_gemm(up, B * Lup, Cin, wp, b, k, relu=True)        # (B*Lup, Cout), rows t < Lout valid
                saved += [up, y]
                metas.append(("up", Lin, Lout, Lup, Cin, Cout, k, s, lpad))
                src, src_rows, src_cm = y, Lup, 0
            Lin = Lout
        Cd = metas[2][5]
        out = torch.empty((B, Lin, Cd), device=src.device, dtype=torch.float32)
        call("tvae_upsample_fwd", _p(src), src_rows, 0, _p(out), B, Lin, Cd, 1, 0, Lin, _stream())
        ctx.save_for_backward(W0, W1, W2, *saved)
        ctx.metas, ctx.B = metas, B
        ctx.P = (W0, b0, W1, b1, W2, b2)
        _note_use(*ctx.P)
        return out

    @staticmethod
    def backward(ctx, dout):
        W = ctx.saved_tensors[:3]
        saved = ctx.saved_tensors[3:]
        B = ctx.B
        PAD = DeconvBottleneckFn.PAD
        grads = [None] * 6
        # (dup, d_stride, d_lpad, d_Lup): where the gradient of layer li's OUTPUT rows lives
        dup, d_stride, d_lpad, d_Lup = _c(dout), 1, 0, ctx.metas[2][2]
        for li in (2, 1, 0):
            kind, Lin, Lout = ctx.metas[li][:3]
            xin, y = saved[2 * li], saved[2 * li + 1]
            if kind == "poly":
                _, _, _, Lp, Cin, Cout, k, s, Q = ctx.metas[li]
                # gradient of the pre-activation in the product's own layout (B*Lp, s*Cout) = rows (b, m s + r) x co:
                # masked by the ReLU, zero on the rows outside [PAD, PAD + Lout) of every molecule
                buf = torch.empty((B * Lp * s + PAD, Cout), device=dup.device, dtype=torch.float32)
                buf[:PAD].zero_()                       # tvae_upsample_bwd writes all B*Lp*s rows behind the first PAD
                call("tvae_upsample_bwd", _p(dup), _p(y[PAD:]), _p(buf[PAD:]), Lp * s, 0, B, Lout, Cout, d_stride, d_lpad,
                     d_Lup, _stream())
                rows = B * Lp - (Q - 1)
                pW, pb = ctx.P[2 * li], ctx.P[2 * li + 1]
                direct = _direct(pW, pb)
                with _side_block(buf, xin) if direct else _side_block() as _on:      # (_side_block() without operands: plain)
                    dwp = torch.zeros((s * Cout, Q * Cin), device=dup.device, dtype=torch.float32)
                    dbp = torch.zeros((s * Cout,), device=dup.device, dtype=torch.float32)
                    gemm(buf, s * Cout, xin, Cin, dwp, Q * Cin, s * Cout, Q * Cin, rows,
                         flags=GEMM_A_KMAJOR | GEMM_B_KMAJOR | GEMM_SPLITK, a_rowsum=dbp, dtype=_role_dtype("wgrad"))
                    dW = torch.empty_like(W[li])
                    db = torch.empty((Cout,), device=dup.device, dtype=torch.float32)
                    call("tvae_deconv_poly_pack", _p(dwp), _p(dW), _p(dbp), _p(db), Cin, Cout, k, s, 2, _stream())
                    if direct:
                        _accumulate_direct(((pW, dW), (pb, db)))
                    else:
                        grads[2 * li], grads[2 * li + 1] = dW, db
                if li > 0 or ctx.needs_input_grad[0]:
                    wd, _ = _poly_pack(W[li], None, s, 1)                    # (Cin, Q*s*Cout)
                    dx = torch.empty((B * Lp, Cin), device=dup.device, dtype=torch.float32)
                    gemm(buf, s * Cout, wd, Q * s * Cout, dx, Cin, rows, Cin, Q * s * Cout, dtype=_role_dtype("dgrad"))
                    dup, d_stride, d_lpad, d_Lup = dx, 1, 0, Lp              # input i of molecule b: row b Lp + i
            else:
                _, _, _, Lup, Cin, Cout, k, s, lpad = ctx.metas[li]
                buf = torch.empty(((k - 1) + B * Lup, Cout), device=dup.device, dtype=torch.float32)
                buf[:k - 1].zero_()                     # tvae_upsample_bwd writes all B*Lup rows behind them
                dyr = buf[k - 1:]
                call("tvae_upsample_bwd", _p(dup), _p(y), _p(dyr), Lup, 0, B, Lout, Cout, d_stride, d_lpad, d_Lup,
                     _stream())
                pW, pb = ctx.P[2 * li], ctx.P[2 * li + 1]
                if _direct(pW, pb):
                    with _side_block(dyr, xin):
                        dwp, db_, tr = _conv_wgrad(dyr, xin, B * Lup, Cin, Cout, k)
                        _accumulate_direct(((pW, _unpack_dw(dwp, W[li].shape, Cout, Cin, k, True, tr)), (pb, db_)))
                else:
                    dwp, grads[2 * li + 1], tr = _conv_wgrad(dyr, xin, B * Lup, Cin, Cout, k)
                    grads[2 * li] = _unpack_dw(dwp, W[li].shape, Cout, Cin, k, True, tr)
                if li > 0 or ctx.needs_input_grad[0]:
                    wd = _pack_w(W[li], Cin, Cout, k, False)                 # wd[ci][j*Cout+co] = W[ci][co][j]
                    dup = _conv_dgrad(buf, B * Lup, Cout, wd, Cin, k)        # gradient of the zero-stuffed input
                    d_stride, d_lpad, d_Lup = s, lpad, Lup
        dh = None
        if ctx.needs_input_grad[0]:
            Lin, Cin = ctx.metas[0][1], ctx.metas[0][4]
            dh = torch.empty((B, Cin * Lin), device=dup.device, dtype=torch.float32)
            call("tvae_upsample_bwd", _p(dup), None, _p(dh), Lin, 1, B, Lin, Cin, d_stride, d_lpad, d_Lup, _stream())
        return (dh,) + tuple(grads)


# ----------------------------------------------------------------------------- loss
def loss_fwd_bwd(tokens, logits, weights, pred_len, mu, logvar, pred_prop, true_prop, beta, pad_idx,
                 want_grads=True, true_len=None, n_groups=1):
    """trans_vae_loss (reference loss.py:23-38) + its gradients in two launches.
    Returns (out5 device tensor [total,bce,bce_mask,kld,mse] -- (n_groups, 5) if n_groups > 1 --,
    grads dict).  n_groups > 1: the batch is that many consecutive chunks, each normalised on its
    own; gradients are those of the sum of the chunk losses (the reference's accumulation)."""
    B, Lp1 = tokens.shape
    L = Lp1 - 1
    n_cls = logits.shape[-1]
    lg = _c(logits).view(B * L, n_cls)
    dev = lg.device
    out5 = torch.empty((n_groups, 5), device=dev, dtype=torch.float32)
    scratch = torch.empty(8 * n_groups, device=dev, dtype=torch.float32)
    g = {}
    n_len = 0 if pred_len is None else pred_len.shape[1]      # None: vae_loss of the RNN models, no length head
    if want_grads:
        g["logits"] = torch.empty_like(lg)
        g["pred_len"] = None if pred_len is None else torch.empty_like(pred_len)
        g["mu"] = torch.empty_like(mu)
        g["logvar"] = torch.empty_like(logvar)
        g["pred_prop"] = torch.empty(B, device=dev, dtype=torch.float32) if pred_prop is not None else None
    pp = None if pred_prop is None else _c(pred_prop).view(-1)
    tp = None if pred_prop is None else _c(true_prop.to(torch.float32)).view(-1)
    tl = None if true_len is None else _c(true_len.long()).view(-1)
    call("tvae_loss_fwd_bwd", _p(_c(tokens)), _p(lg), lg.stride(0), _p(weights), _p(None if pred_len is None else _c(pred_len)), _p(_c(mu)),
         _p(_c(logvar)), _p(pp), _p(tp), _p(tl), float(beta), B, L, n_cls, n_len, mu.shape[1], int(pad_idx), int(n_groups),
         _p(out5), _p(g.get("logits")), _p(g.get("pred_len")), _p(g.get("mu")), _p(g.get("logvar")),
         _p(g.get("pred_prop")), _p(scratch), _stream())
    return (out5 if n_groups > 1 else out5[0]), g


class LossFn(torch.autograd.Function):
    """Autograd face of loss_fwd_bwd: returns the 5 scalars; backward scales the stored gradients
    by the incoming gradient of `total` (the other four outputs are monitors)."""

    @staticmethod
    def forward(ctx, tokens, logits, mu, logvar, pred_len, true_prop, pred_prop, weights, beta, pad_idx, true_len):
        out5, g = loss_fwd_bwd(tokens, logits, weights, pred_len, mu, logvar, pred_prop, true_prop, beta, pad_idx,
                               true_len=true_len)
        ctx.g = g
        ctx.shapes = (logits.shape, None if pred_prop is None else pred_prop.shape)
        total, bce, bmask, kld, mse = out5.unbind(0)
        ctx.mark_non_differentiable(bce, bmask, kld, mse)
        return total, bce, bmask, kld, mse

    @staticmethod
    def backward(ctx, gt, *_):
        g = ctx.g
        dpp = None
        if g["pred_prop"] is not None:
            dpp = (g["pred_prop"] * gt).view(ctx.shapes[1])
        dpl = None if g["pred_len"] is None else g["pred_len"] * gt
        return (None, (g["logits"] * gt).view(ctx.shapes[0]), g["mu"] * gt, g["logvar"] * gt, dpl,
                None, dpp, None, None, None, None)


# ----------------------------------------------------------------------------- Adam
def masked_grad(dy2, drop, force=False):
    """(gradient operand, a_drop) for the two GEMMs that consume the gradient of a dropped-out
    output.  fp32 / bf16 kernels regenerate the keep-mask while staging (a_drop); the f32x3 kernel
    stages a k-major operand one row at a time, across the mask's hash groups, so there the mask is
    applied once by an elementwise pass and both GEMMs read the product."""
    pre = _take_premasked(dy2, drop)
    if pre is not None:                      # the LayerNorm backward that produced dy2 wrote the masked copy too
        return pre, None
    streamed_bf16 = _state["dtype"] == BF16 and os.environ.get("TVAE_BF16_STREAM", "1") == "1"   # copies A as it is
    if drop is None or not (force or streamed_bf16 or F32X3 in (_state["dtype"], _state["wgrad"], _state["dgrad"])):
        return dy2, drop
    out = torch.empty_like(dy2)
    r, s, p = _drop_args(drop)
    call("tvae_residual_dropout", _p(dy2), None, _p(out), dy2.shape[0], dy2.shape[1], 0, 0, r, s, p, _stream())
    return out, None


class FusedAdam:
    """One-launch Adam over a list of (param, grad, exp_avg, exp_avg_sq) tensors."""
    CHUNK = 4096

    def __init__(self, params, betas=(0.9, 0.98), eps=1e-9):
        self.params = list(params)
        dev = self.params[0].device
        self.betas, self.eps = betas, eps
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self._hyper = (0.0, 1.0, 1.0, 1.0)           # lr, bias corrections, grad scale of the NEXT launch
        sizes = [p.numel() for p in self.params]
        ct, cs = [], []
        for i, n in enumerate(sizes):
            for st in range(0, n, self.CHUNK):
                ct.append(i)
                cs.append(st)
        self.n_chunks = len(ct)
        self.sizes = torch.tensor(sizes, dtype=torch.int64, device=dev)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=dev)
        self.chunk_start = torch.tensor(cs, dtype=torch.int64, device=dev)
        self._ptrs = None
        self._ptr_key = None

    def _ptr_table(self):
        grads = []
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            grads.append(p.grad)
        key = tuple(t.data_ptr() for t in self.params + grads)
        if key != self._ptr_key:
            allp = [t.data_ptr() for t in self.params + grads + self.exp_avg + self.exp_avg_sq]
            self._ptrs = torch.tensor(allp, dtype=torch.int64, device=self.params[0].device)
            self._ptr_key = key
        return self._ptrs

    def set_hyper(self, lr, step, grad_scale=1.0):
        """Numbers of the next launch.  They travel BY VALUE in the launch itself: the host queues steps ahead of
        the GPU (train() reads losses two steps late, bench.py never syncs between steps), so a device or pinned
        buffer rewritten per step would hand step t the learning rate of step t+1 or t+2."""
        b1, b2 = self.betas
        self._hyper = (float(lr), 1.0 - b1 ** step, 1.0 - b2 ** step, float(grad_scale))

    def launch(self, zero_grad=False):
        join_side()
        ptrs = self._ptr_table()
        call("tvae_adam_multi", _p(ptrs), _p(self.sizes), _p(self.chunk_tensor), _p(self.chunk_start),
             len(self.params), self.n_chunks, None, *self._hyper, self.betas[0], self.betas[1], self.eps,
             int(zero_grad), _stream())
        weights_changed()


class ResidualDropoutFn(torch.autograd.Function):
    """out = res + dropout(x) (res optional): the generic SublayerConnection / nn.Dropout glue."""

    @staticmethod
    def forward(ctx, x, res, drop):
        x2 = _c(x).view(-1, x.shape[-1])
        out = torch.empty_like(x2)
        r, s, p = _drop_args(drop)
        res2 = None if res is None else _c(res).view(x2.shape)
        call("tvae_residual_dropout", _p(x2), _p(res2), _p(out), x2.shape[0], x2.shape[1], 0, 0, r, s, p, _stream())
        ctx.drop, ctx.has_res = drop, res is not None
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        dy2 = _c(dy).view(-1, dy.shape[-1])
        if ctx.drop is None:
            dx = dy
        else:
            dx2 = torch.empty_like(dy2)
            r, s, p = _drop_args(ctx.drop)
            call("tvae_residual_dropout", _p(dy2), None, _p(dx2), dy2.shape[0], dy2.shape[1], 0, 0, r, s, p, _stream())
            dx = dx2.view(dy.shape)
        return dx, (dy if ctx.has_res else None), None


class PosEncFn(torch.autograd.Function):
    """out = dropout(x + pe[:L])  (PositionalEncoding.forward on its own, trans_models.py:1060-1063)."""

    @staticmethod
    def forward(ctx, x, pe, drop):
        B, L, d = x.shape
        x2 = _c(x).view(-1, d)
        out = torch.empty_like(x2)
        r, s, p = _drop_args(drop)
        call("tvae_residual_dropout", _p(x2), _p(pe.view(-1, d)), _p(out), B * L, d, L, 1, r, s, p, _stream())
        ctx.drop = drop
        return out.view(B, L, d)

    @staticmethod
    def backward(ctx, dy):
        if ctx.drop is None:
            return dy, None, None
        dy2 = _c(dy).view(-1, dy.shape[-1])
        dx2 = torch.empty_like(dy2)
        r, s, p = _drop_args(ctx.drop)
        call("tvae_residual_dropout", _p(dy2), None, _p(dx2), dy2.shape[0], dy2.shape[1], 0, 0, r, s, p, _stream())
        return dx2.view(dy.shape), None, None


# ----------------------------------------------------------------------------- GRU (RNN / RNNAttn models)
def _dropped(x2, drop):
    "x2 * keep-mask / (1 - p) as a new tensor (None drop: x2 itself)"
    if drop is None:
        return x2
    out = torch.empty_like(x2)
    r, s, p = _drop_args(drop)
    call("tvae_residual_dropout", _p(x2), None, _p(out), x2.shape[0], x2.shape[1], 0, 0, r, s, p, _stream())
    return out


_gru_side = {}


def _gru_streams(device, n):
    "side streams of the layer pipeline, one per layer, kept per device"
    key = (device.index if device.index is not None else torch.cuda.current_device())
    have = _gru_side.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have[:n]


def _chunk_drop(drop, c):
    "the dropout site of time chunk c of an inter-layer mask: chunks index their elements locally, so each gets its own stream"
    return None if drop is None else Drop(drop.p, drop.site ^ ((c + 1) << 10), drop.rng)


def _gru_chunks(T):
    ch = int(os.environ.get("TVAE_GRU_CHUNK", "16"))
    if ch <= 0 or ch >= T:
        return [(0, T)]
    return [(t0, min(T, t0 + ch)) for t0 in range(0, T, ch)]


class GRUStackFn(torch.autograd.Function):
    """torch.nn.GRU(num_layers=N, dropout=p) on a (B, T, I) input from h0 = 0 (reference transvae/rnn_models.py:241, 301,
    333, 378): returns (output of the top layer (B, T, H), final state of the top layer (B, H)).

    Per layer: the input projections are GEMMs over many time steps at once, the recurrence is tvae_gru_fwd (one fused
    launch per time step), the inter-layer dropout is a mask regenerated in the backward pass.  Inside, sequences are
    time-major (T, B, .) with a leading block of B zero rows (h_{-1}): the rows of h_{t-1} then sit exactly B rows before
    those of h_t, which makes the recurrent weight gradient ONE product dgh^T [0; h_0 .. h_{T-2}] over all steps.

    A step launch is latency-bound (7 us forward, 13 us backward whatever the arithmetic in it), so a stack of N layers is
    run as a PIPELINE over chunks of 16 time steps: layer l works on chunk c on its own HIP stream while layer l-1 is at
    chunk c+1 (events hand the chunks over), which shortens the chain of dependent launches from N*T to T + 16 (N-1); the
    backward pass does the same from the top layer and the last chunk down.  TVAE_GRU_CHUNK=0 runs layer after layer."""

    @staticmethod
    def forward(ctx, x, drops, *params):
        B, T, I = x.shape
        n_layers = len(params) // 4
        H = params[1].shape[1]
        _chk(x, "x")
        dev = x.device
        need_grad = any(ctx.needs_input_grad)          # False under torch.no_grad(): nothing is kept for a backward pass
        chunks = _gru_chunks(T) if n_layers > 1 else [(0, T)]
        main = torch.cuda.current_stream()
        x_tm = x.transpose(0, 1).contiguous().view(T * B, I)
        f32 = dict(device=dev, dtype=torch.float32)
        hbufs, gates, gis, xins = [], [], [], []
        for l in range(n_layers):                      # every buffer is allocated on the caller's stream, before the pipeline
            hb = torch.empty(((T + 1) * B, H), **f32)
            hb[:B].zero_()
            hbufs.append(hb)
            gates.append(torch.empty((T * B, 4 * H), **f32) if need_grad else None)
            gis.append(torch.empty((T * B, 3 * H), **f32))
            if l == 0:
                xins.append(x_tm)
            elif drops[l - 1] is not None:
                xins.append(torch.empty((T * B, H), **f32))
            else:
                xins.append(hbufs[l - 1][B:])
        linear_fwd(x_tm, params[0], params[2], out=gis[0])
        streams = _gru_streams(dev, n_layers) if len(chunks) > 1 else [main] * n_layers
        done = [[None] * len(chunks) for _ in range(n_layers)]
        for st in set(streams):
            if st is not main:
                st.wait_stream(main)
        # launches are ENQUEUED in wavefront order (chunk c of layer l next to chunk c+1 of layer l-1): the host issues a
        # launch about as fast as the GPU retires one, so work enqueued layer after layer would also run layer after layer
        order = sorted(((l, c) for l in range(n_layers) for c in range(len(chunks))), key=lambda lc: (lc[0] + lc[1], lc[0]))
        for l, c in order:
            w_ih, w_hh, b_ih, b_hh = params[4 * l:4 * l + 4]
            st = streams[l]
            t0, t1 = chunks[c]
            rows = slice(t0 * B, t1 * B)
            with torch.cuda.stream(st):
                if l > 0:
                    if st is not main:
                        st.wait_event(done[l - 1][c])
                    if drops[l - 1] is not None:
                        r, s_, p = _drop_args(_chunk_drop(drops[l - 1], c))
                        src = hbufs[l - 1][B:][rows]
                        call("tvae_residual_dropout", _p(src), None, _p(xins[l][rows]), src.shape[0], H, 0, 0, r, s_, p,
                             _stream())
                    linear_fwd(xins[l][rows], w_ih, b_ih, out=gis[l][rows])
                call("tvae_gru_fwd", _p(gis[l][rows]), _p(w_hh), _p(b_hh), _p(hbufs[l][t0 * B:]),
                     _p(hbufs[l][(t0 + 1) * B:]), _p(None if gates[l] is None else gates[l][rows]), t1 - t0, B, H, 1,
                     _stream())
                if st is not main:
                    done[l][c] = st.record_event()
        for st in streams:
            if st is not main:
                main.wait_stream(st)
        top = hbufs[-1][B:]
        out = top.view(T, B, H).transpose(0, 1).contiguous()
        h_last = top.view(T, B, H)[T - 1].clone()
        if need_grad:
            saved = []
            for l in range(n_layers):
                saved += [xins[l], hbufs[l], gates[l]]
            ctx.save_for_backward(*saved)
        ctx.P, ctx.drops, ctx.dims, ctx.chunks = params, drops, (B, T, I, H, n_layers), chunks
        _note_use(*params)
        return out, h_last

    @staticmethod
    def backward(ctx, dout, dh_last):
        B, T, I, H, n_layers = ctx.dims
        saved, params, chunks = ctx.saved_tensors, ctx.P, ctx.chunks
        dev = saved[0].device
        f32 = dict(device=dev, dtype=torch.float32)
        main = torch.cuda.current_stream()
        top = n_layers - 1
        d = [None] * n_layers                          # gradient arriving at every h_t of layer l
        d[top] = None if dout is None else dout.transpose(0, 1).contiguous().view(T * B, H)
        dl = None if dh_last is None else _c(dh_last)
        for l in range(top):
            d[l] = torch.empty((T * B, H), **f32)
        dgi = [torch.empty((T * B, 3 * H), **f32) for _ in range(n_layers)]
        dgh = [torch.empty((T * B, 3 * H), **f32) for _ in range(n_layers)]
        carry = [torch.empty((B, H), **f32) for _ in range(n_layers)]
        w_hh_t = [params[4 * l + 1].detach().t().contiguous() for l in range(n_layers)]
        streams = _gru_streams(dev, n_layers) if len(chunks) > 1 else [main] * n_layers
        ready = [[None] * len(chunks) for _ in range(n_layers)]
        for st in set(streams):
            if st is not main:
                st.wait_stream(main)
        nc = len(chunks)
        order = sorted(((l, c) for l in range(n_layers) for c in range(nc)), key=lambda lc: ((top - lc[0]) + (nc - 1 - lc[1]), -lc[0]))
        for l, c in order:                             # wavefront from the top layer's last chunk down
            w_ih = params[4 * l]
            xin, hbuf, gates = saved[3 * l:3 * l + 3]
            st = streams[l]
            t0, t1 = chunks[c]
            rows = slice(t0 * B, t1 * B)
            with torch.cuda.stream(st):
                if l < top and st is not main:
                    st.wait_event(ready[l + 1][c])
                final = t1 == T
                call("tvae_gru_bwd", _p(None if d[l] is None else d[l][rows]), _p(dl if (l == top and final) else None),
                     _p(hbuf[(t0 + 1) * B:]), _p(gates[rows]), _p(hbuf[t0 * B:]), _p(w_hh_t[l]), _p(dgi[l][rows]),
                     _p(dgh[l][rows]), _p(carry[l]), t1 - t0, B, H, 1, 0 if final else 1, _stream())
                if l > 0:
                    linear_dgrad(dgi[l][rows], w_ih, out=d[l - 1][rows])
                    if ctx.drops[l - 1] is not None:
                        r, s_, p = _drop_args(_chunk_drop(ctx.drops[l - 1], c))
                        tgt = d[l - 1][rows]
                        call("tvae_residual_dropout", _p(tgt), None, _p(tgt), tgt.shape[0], H, 0, 0, r, s_, p, _stream())
                    if st is not main:
                        ready[l][c] = st.record_event()
        for st in streams:
            if st is not main:
                main.wait_stream(st)
        grads = [None] * (4 * n_layers)
        for l in range(n_layers):
            w_ih, w_hh, b_ih, b_hh = params[4 * l:4 * l + 4]
            xin, hbuf, _ = saved[3 * l:3 * l + 3]
            grads[4 * l], grads[4 * l + 2] = linear_wgrad(dgi[l], xin, W=w_ih, b=b_ih, side=False)
            grads[4 * l + 1], grads[4 * l + 3] = linear_wgrad(dgh[l], hbuf[:T * B], W=w_hh, b=b_hh, side=False)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(dgi[0], params[0]).view(T, B, I).transpose(0, 1).contiguous()
        return (dx, None) + tuple(grads)


class SoftmaxFn(torch.autograd.Function):
    "softmax over the last dimension (F.softmax(.., dim=2) of RNNAttnEncoder, rnn_models.py:246)"

    @staticmethod
    def forward(ctx, x):
        x2 = _c(x).view(-1, x.shape[-1])
        _chk(x2, "x")
        y = torch.empty_like(x2)
        call("tvae_softmax_rows_fwd", _p(x2), x2.stride(0), _p(y), y.stride(0), x2.shape[0], x2.shape[1], _stream())
        ctx.save_for_backward(y)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy2 = _c(dy).view(y.shape)
        dx = torch.empty_like(y)
        call("tvae_softmax_rows_bwd", _p(y), y.stride(0), _p(dy2), dy2.stride(0), _p(dx), dx.stride(0), y.shape[0], y.shape[1],
             _stream())
        return dx.view(dy.shape)


def gemm_batched(A, lda, sA, Bm, ldb, sB, Cm, ldc, sC, M, N, K, batch, residual=None, ldr=0, sR=0, flags=0):
    call("tvae_gemm_batched", _p(A), lda, sA, _p(Bm), ldb, sB, _p(Cm), ldc, sC, M, N, K, batch, None, _p(residual), ldr, sR,
         1.0, flags, _stream())


class BmmReluFn(torch.autograd.Function):
    "relu(torch.bmm(p, v)): p (B, L, S) attention weights, v (B, S, d)  (rnn_models.py:247-248), one launch for all B"

    @staticmethod
    def forward(ctx, p, v):
        p, v = _c(p), _c(v)
        _chk(p, "p")
        _chk(v, "v")
        B, L, S = p.shape
        d = v.shape[2]
        out = torch.empty((B, L, d), device=p.device, dtype=torch.float32)
        gemm_batched(p, S, L * S, v, d, S * d, out, d, L * d, L, d, S, B, flags=GEMM_B_KMAJOR | GEMM_RELU)
        ctx.save_for_backward(p, v, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        p, v, out = ctx.saved_tensors
        B, L, S = p.shape
        d = v.shape[2]
        dpre = relu_mask(_c(dout).view(B * L, d), out.view(B * L, d))
        dp = torch.empty_like(p)
        dv = torch.empty_like(v)
        gemm_batched(dpre, d, L * d, v, d, S * d, dp, S, L * S, L, S, d, B)                       # dpre_b v_b^T
        gemm_batched(p, S, L * S, dpre, d, L * d, dv, d, S * d, S, d, L, B,
                     flags=GEMM_A_KMAJOR | GEMM_B_KMAJOR)                                          # p_b^T dpre_b
        return dp, dv
