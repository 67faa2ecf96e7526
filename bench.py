#!/usr/bin/env python3
"""Benchmark of the TransVAE hot path on MI355X: trans4x-256 training step (forward + loss +
backward + Noam/Adam, dropout ON as the reference trains) and KV-cached greedy decode.

    python bench.py --gpus 1 --steps 10 --warmup 3            # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `value` = whole-job train tokens/s (B*126 target positions per
step and rank, pads included: the algorithm is dense), inputs resident in HBM.
Extra keys: decode molecules/s, `roofline` for the dominant kernel (the MFMA GEMM, timed live
with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle -- a port of the
reference's path pinned by golden vectors -- timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md dense MFMA peaks; f32x3 = fp32-accurate products from six bf16 MFMAs: bf16 peak / 6
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "f32x3": 2500.0 / 6}
FLOP_PER_MOL_FWD_BWD = 5.940e9                      # trans4x-256, SURVEY.md 8(d)
FLOP_PER_MOL_DECODE = 1.196e9


def build_vae(arch="trans4x-256", batch=500, chunks=5):
    from transvae_amd import synth
    from transvae_amd.trans_models import TransVAE
    cd, od, w = synth.load_vocab("zinc")
    params = {"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w, "BATCH_SIZE": batch, "BATCH_CHUNKS": chunks}
    vae = TransVAE(params=params, name="bench", **synth.ARCHS[arch])
    vae.model.load_state_dict(synth.synth_state_dict(vae.model.state_dict(), 0))
    return vae


def time_launches(fn, iters, warm_seconds=0.25):
    """Average duration (s) of `fn`'s launch at the clock a busy GPU runs at.  An MI355X that has idled (allocation, a host
    pause) climbs to its working clock over roughly the first 50 ms of load: the same fp32 GEMM launch goes 360 -> 290 us
    over that stretch (tools/debug/warm_probe.py).  The train step keeps the GPU busy for whole seconds, so the roofline
    launch is timed the same way: first ~warm_seconds of back-to-back launches, then `iters` timed ones with HIP events on
    the launch stream (torch's current stream)."""
    import time
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < warm_seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def time_gemm(M, N, K, dtype_name, iters=50):
    """Average duration (s) of the dominant kernel on one representative launch shape."""
    from transvae_amd import ops
    x = torch.randn(M, K, device="cuda")
    w = torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.05, requires_grad=False)   # a weight, as in the model
    b = torch.zeros(N, device="cuda")
    out = ops.rows_buffer(M, N, "cuda")              # the padded-row buffer the model itself allocates
    return time_launches(lambda: ops.linear_fwd(x, w, b, relu=True, out=out), iters)


def roofline_of(mode, M, N, K):
    """Roofline object of the dominant kernel (the MFMA GEMM) on one launch shape, timed live with events."""
    from transvae_amd import ops
    ops.set_compute_dtype(mode)
    tg = time_gemm(M, N, K, mode)
    flops = 2.0 * M * N * K
    ach = flops / tg / 1e12
    # HBM bytes per launch of this kernel and shape, from separate rocprofv3 --pmc passes
    # (tools/pmc_traffic.sh -> profiles/r<round>_gemm_pmc_<mode>.json); None when no such measurement exists
    traffic, traffic_source = None, None
    for tag in ("r3", "r2", "r1"):        # the newest measurement of this kernel and shape
        pmc = os.path.join(ROOT, "profiles", "%s_gemm_pmc_%s.json" % (tag, mode))
        if os.path.exists(pmc):
            rec = json.load(open(pmc))
            if rec.get("shape") == [M, N, K]:
                traffic = rec["traffic_bytes_per_launch"]
                traffic_source = ("profiles/%s_gemm_pmc_%s.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this "
                                  "kernel and shape (tools/pmc_traffic.sh), not measured in this run" % (tag, mode))
                break
    kern = {"fp32": "gemm_kernel (v_mfma_f32_32x32x2_f32)", "bf16": "gemm_dma_kernel (v_mfma_f32_32x32x16_bf16; fp32 activations and bf16 weights streamed to LDS by LDS-DMA)",
            "f32x3": "gemm3_dma_kernel (6 x v_mfma_f32_32x32x16_bf16 per fp32 product block, fp32 activations and pre-split weight planes streamed to LDS by LDS-DMA; peak = bf16 peak / 6; a bare loop of its fragment reads and MFMAs reaches 0.70 of that peak: profiles/r3_mfma_x3_bare_loop.txt)"}[mode]
    return {"bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS[mode], "unit": "TFLOP/s",
            "frac": ach / PEAK_TFLOPS[mode], "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes": 4.0 * (M * K + N * K + N + M * N),
            "kernel": "%s: x[%d,%d] @ W[%d,%d]^T + bias, ReLU" % (kern, M, K, N, K),
            "launch_us": tg * 1e6, "flop_per_launch": flops}


def decode_attn_launch(n, n_keys=63, d=256, h=4, T=126):
    """(callable that launches the dominant decode kernel once, algorithmic bytes of that launch): the one-query attention
    of a decode step over the self-attention K/V cache, at the mean key count of a 125-step decode.  Algorithmic bytes:
    the n_keys cached K rows and V rows of every molecule once, plus the query and the output row."""
    from transvae_amd import ops
    from transvae_amd._lib import call
    dev = torch.device("cuda", torch.cuda.current_device())
    ks = torch.randn(n, T, d, device=dev)
    vs = torch.randn(n, T, d, device=dev)
    q = torch.randn(n, 3 * d, device=dev)
    o = torch.empty(n, d, device=dev)
    dk = d // h

    def fn():
        call("tvae_decode_attn", ops._p(q), 3 * d, ops._p(ks), ops._p(vs), d, T, ops._p(o), d, None, 0, n_keys, n, h, dk,
             1.0 / dk ** 0.5, ops._stream())
    return fn, 4.0 * n * d * (2 * n_keys + 2)


def roofline_decode_of(n):
    "roofline object of decode_attn_kernel (HBM-bound), timed live with events on the launch stream"
    fn, nbytes = decode_attn_launch(n)
    t = time_launches(fn, 20)
    traffic, traffic_source = None, None
    for tag in ("r3", "r2"):
        pmc = os.path.join(ROOT, "profiles", "%s_decode_attn_pmc.json" % tag)
        if os.path.exists(pmc):
            rec = json.load(open(pmc))
            if rec.get("shape", {}).get("n") == n:
                traffic = rec["traffic_bytes_per_launch"]
                traffic_source = "profiles/%s_decode_attn_pmc.json (rocprofv3 --pmc passes, tools/pmc_decode.sh), not measured in this run" % tag
                break
    return {"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": nbytes / t / 1e9 / 8000.0,
            "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes": nbytes, "launch_us": t * 1e6,
            "kernel": "decode_attn_kernel<64>: one query per (molecule, head) over 63 cached fp32 K and V rows, %d molecules "
                      "(half of the decode's device time; the other half is the fp32 MFMA GEMMs of `roofline`)" % n}


def _live_tile_fraction(key_mask, Lq, causal):
    """Fraction of the (32-query block, 32-key tile) pairs of an attention call that the kernels execute: a pair whose mask
    words are all zero is skipped (it only produces exact zeros), csrc/attention.hip.  key_mask (B, Lk) bool on the host."""
    B, Lk = key_mask.shape
    nq, nk = (Lq + 31) // 32, (Lk + 31) // 32
    live = 0
    vis = key_mask.bool()
    for kt in range(nk):
        any_key = vis[:, 32 * kt:32 * kt + 32].any(1)                      # (B,)
        for qb in range(nq):
            if causal and 32 * kt > min(32 * qb + 31, Lq - 1):             # the whole tile lies above the diagonal
                continue
            live += int(any_key.sum())
    return live / float(B * nq * nk)


def roofline_attention_of(batch, h=4, dk=64, iters=20):
    """Roofline object of the fused attention kernels (csrc/attention.hip) on the encoder self-attention shape of the
    train step (127 x 127 per molecule and head, 4 of the 10 attention blocks; the other 6 are 126 x 126 / 126 x 127), on
    the SMILES-shaped batch the step runs on and on a batch without padding.  FLOPs: dense count 4 Lq Lk d per molecule
    forward (QK^T and PV), 2.5 x that backward (S recomputed, dP, dV, dQ, dK) -- SURVEY.md 8(d) -- and the LIVE-tile count
    (the tiles the kernels execute; all-masked tiles are skipped because they are exact zeros).  Bytes: Q, K, V in and O
    out (forward), Q, K, V, O, dO in and dQ, dK, dV out (backward), fp32 rows of the surrounding GEMMs."""
    from transvae_amd import ops, synth
    B, d, Lq, Lk = batch, h * dk, 127, 127
    dev = torch.device("cuda", torch.cuda.current_device())
    toks = synth.synth_tokens(B, seed=100)
    drop = ops.Drop(0.1, 3, ops.rng_state(dev))
    dense_f = 4.0 * Lq * Lk * d * B
    out = {"bound": "mfma", "peak": PEAK_TFLOPS["fp32"], "unit": "TFLOP/s", "hbm_peak_GBps": 8000.0,
           "shape": "encoder self-attention of the train step: B %d, h %d, Lq = Lk = 127, dk %d, dropout 0.1 on P" % (B, h, dk),
           "kernels": {}}
    for pad in ("smiles", "unpadded"):
        km = (toks[:, :Lk] != 24) if pad == "smiles" else torch.ones(B, Lk, dtype=torch.bool)
        spec = ops.MaskSpec(key_mask=km.to(torch.uint8).to(dev), causal=False)
        q, k, v, do = (torch.randn(B * L, d, device=dev) for L in (Lq, Lk, Lk, Lq))
        o, lse, _ = ops.attn_fwd(q, k, v, B, h, Lq, Lk, dk, spec, drop)
        tf = time_launches(lambda: ops.attn_fwd(q, k, v, B, h, Lq, Lk, dk, spec, drop), iters, 0.1)
        tb = time_launches(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, h, Lq, Lk, dk, spec, drop), iters, 0.1)
        live = _live_tile_fraction(km, Lq, False)
        row_bytes = 4.0 * B * Lq * d
        for name, t, fl, nbytes in (("attn_fwd_kernel<64>", tf, dense_f, 4 * row_bytes),
                                    ("attn_bwd_dq_kernel<64> + attn_bwd_dkv_kernel<64>", tb, 2.5 * dense_f, 8 * row_bytes)):
            out["kernels"].setdefault(name, {})[pad] = {
                "launch_us": t * 1e6, "mean_visible_keys": float(km.float().sum(1).mean()),
                "dense_flop": fl, "live_tile_fraction": live, "achieved_dense_tflops": fl / t / 1e12,
                "achieved_live_tflops": fl * live / t / 1e12, "frac_of_mfma_peak_live": fl * live / t / 1e12 / PEAK_TFLOPS["fp32"],
                "algorithmic_bytes": nbytes, "achieved_GBps": nbytes / t / 1e9, "frac_of_hbm_peak": nbytes / t / 1e9 / 8000.0}
    bw = out["kernels"]["attn_bwd_dq_kernel<64> + attn_bwd_dkv_kernel<64>"]
    out["achieved"] = bw["unpadded"]["achieved_dense_tflops"]        # headline: the backward pair with every tile live
    out["frac"] = out["achieved"] / PEAK_TFLOPS["fp32"]
    out["traffic"] = None
    out["counters"] = "profiles/r3_attn_sq_counters.txt (rocprofv3 --pmc over tools/attn_bench.py: MFMA-busy, LDS bank conflicts, wave stalls)"
    return out


def config1_of(steps, cpu_threads=None):
    """BASELINE.json configs[0]: trans1x-128, one fp32 train step (forward + loss + backward + Noam/Adam) on a synthetic
    batch of 500 -- on the GPU (the same one-pass step as the headline model) and, if cpu_threads, the CPU oracle beside it."""
    from transvae_amd import ops, synth
    ops.set_compute_dtype("fp32")
    vae = build_vae("trans1x-128")
    vae.model.train()
    dev = torch.device("cuda", torch.cuda.current_device())
    toks = synth.synth_tokens(500, seed=100)
    data = torch.cat([toks.float(), torch.zeros(500, 1)], 1).to(dev)
    for _ in range(3):
        vae.train_step(data, 0.01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        vae.train_step(data, 0.01)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"workload": "trans1x-128 ZINC-vocab train step (fwd+loss+bwd+Noam/Adam), batch 500 in one pass (5 x 100 chunk-wise loss "
                       "normalisation), dropout 0.1, fp32, synthetic SMILES-shaped tokens, synthetic checkpoint",
           "value": 500 * 126 / dt, "unit": "tokens/s", "ms_per_step": 1e3 * dt, "steps": steps, "dtype": "f32",
           "train_algorithmic_tflops": 500 * 1.141e9 / dt / 1e12}
    del vae
    if cpu_threads:
        out["cpu_baseline"] = cpu_baseline(cpu_threads, arch="trans1x-128", cfg="1x128_zinc", d_model=128)
    return out


def rnn_secondary(kind, d_model, batch=500, steps=5, decode_n=16384):
    """Secondary figures for SURVEY.md 8(f) row 4: train step and greedy decode of a recurrent model (same synthetic
    batch shape, dropout on, exact-fp32 recurrence; the input-projection GEMMs follow the selected arithmetic)."""
    from transvae_amd import ops, rnn_models, synth
    cd, od, w = synth.load_vocab("zinc")
    cls = {"rnnattn": rnn_models.RNNAttn, "rnn": rnn_models.RNN}[kind]
    vae = cls(params={"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w, "BATCH_SIZE": batch, "BATCH_CHUNKS": 5},
              name="bench_" + kind, d_model=d_model)
    vae.model.load_state_dict(synth.synth_state_dict(vae.model.state_dict(), 0))
    vae.model.train()
    dev = torch.device("cuda", torch.cuda.current_device())
    data = torch.cat([synth.synth_tokens(batch, seed=1).float(), torch.zeros(batch, 1)], 1).to(dev)
    for _ in range(3):
        vae.train_step(data, 0.01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        vae.train_step(data, 0.01)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    z = torch.randn(decode_n, 128, generator=torch.Generator().manual_seed(0)).to(dev)
    vae.greedy_decode(z[:64])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vae.greedy_decode(z)
    torch.cuda.synchronize()
    td = time.perf_counter() - t0
    return {"model": "%s-%d" % (kind, d_model), "train_tokens_per_s": batch * 126 / dt, "ms_per_step": 1e3 * dt,
            "batch": batch, "decode_mols_per_s": decode_n / td, "decode_batch": decode_n}


def host_threads():
    """Threads for the CPU baseline: this process's CPU share.  On the GPU box os.cpu_count() is the whole 8-GPU host; the
    share of one GPU's job is what OMP_NUM_THREADS says there (16), so that is used when set, else every core."""
    env = os.environ.get("OMP_NUM_THREADS", "")
    return int(env) if env.isdigit() and int(env) > 0 else (os.cpu_count() or 1)


def cpu_baseline(threads, batch=500, chunks=5, arch="trans4x-256", cfg="4x256_zinc", d_model=256, n_timed=3):
    """The oracle's train step on host cores, as SURVEY.md 8(d) asks: batch 500 = 5 chunks of 100 molecules, each
    chunk forward + loss + backward with the gradients accumulating, then one Noam/Adam step.  Two figures: dropout
    on (what scripts/train.py of the reference really runs; `value`) and dropout off.  Bounded sample: one untimed
    chunk to warm the allocator and the thread pool, then n_timed timed steps per figure (median reported, spread kept)."""
    from oracle import tvae_oracle as O
    from transvae_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_params
    torch.set_num_threads(threads)
    P = oracle_params(cfg, torch.float32, requires_grad=True)
    keys = [k for k in P if P[k].requires_grad]        # the two positional tables are buffers
    w = torch.tensor(synth.load_vocab("zinc")[2], dtype=torch.float32)
    toks = synth.synth_tokens(batch, seed=1)
    props = torch.zeros(batch)
    csize = batch // chunks
    m = [torch.zeros_like(P[k]) for k in keys]
    v = [torch.zeros_like(P[k]) for k in keys]
    count = [0]

    def step(p_drop, n_chunks=chunks):
        count[0] += 1
        for k in keys:
            P[k].grad = None
        for c in range(n_chunks):
            sl = slice(c * csize, (c + 1) * csize)
            losses, _ = O.chunk_loss(P, toks[sl], props[sl], w, 0.01, p_drop=p_drop, attn_drop=p_drop)
            losses[0].backward()                      # accumulates, no 1/chunks (trans_models.py:236)
        with torch.no_grad():
            O.adam_step([P[k] for k in keys], [P[k].grad for k in keys], m, v, count[0], O.noam_rate(count[0], d_model))

    def timed(p_drop):
        ts = []
        for _ in range(n_timed):
            t0 = time.perf_counter()
            step(p_drop)
            ts.append(time.perf_counter() - t0)
        return sorted(ts)

    step(0.1, n_chunks=1)                             # warm-up: one chunk
    on, off = timed(0.1), timed(0.0)
    t_on, t_off = on[len(on) // 2], off[len(off) // 2]
    return {"value": batch * 126 / t_on, "unit": "tokens/s", "cores": threads, "kind": "port",
            "value_dropout_off": batch * 126 / t_off,
            "step_seconds_dropout_on": on, "step_seconds_dropout_off": off,
            "host_cpu_count": os.cpu_count(),
            "sample": "train step of %s on %d molecules = %d chunks of %d (forward + loss + backward per chunk, gradients "
                      "accumulated, then Noam/Adam), fp32 torch CPU oracle on %d threads (this job's CPU share; the host shows "
                      "%s); 1 warm-up chunk, then %d timed steps with dropout 0.1 as the reference trains (median %.1f s, range "
                      "%.1f-%.1f; `value`) and %d with dropout off (median %.1f s; `value_dropout_off`)"
                      % (arch, batch, chunks, csize, threads, os.cpu_count(), n_timed, t_on, on[0], on[-1], n_timed, t_off)}


def cpu_decode_baseline(threads, n=24):
    from oracle import tvae_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_params
    torch.set_num_threads(threads)
    P = oracle_params("4x256_zinc", torch.float32)
    z = torch.randn(n, 128, generator=torch.Generator().manual_seed(0))
    t0 = time.perf_counter()
    O.greedy_decode(P, z)
    t = time.perf_counter() - t0
    return {"value": n / t, "unit": "molecules/s", "cores": threads, "kind": "port",
            "sample": "%d latents, reference-style full-recompute greedy loop (trans_models.py:383-436), %.1f s" % (n, t)}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks of this same command line (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) BEFORE anything in this process has touched the
    GPU, pass rank 0's JSON line through, and fail if any rank fails.  Counting devices does not initialise the GPU."""
    import socket
    import subprocess
    rehearse = os.environ.get("TVAE_BENCH_REHEARSE") == "1"
    have = torch.cuda.device_count()
    if have < n and not rehearse:
        sys.stderr.write("bench.py: --gpus %d but this machine shows %d GPU(s); refusing to report a world size it did "
                         "not run (TVAE_BENCH_REHEARSE=1 rehearses N ranks on one GPU over gloo, numbers meaningless)\n"
                         % (n, have))
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    line, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if any(codes):
        sys.stderr.write("bench.py: rank exit codes %s\n" % codes)
        return 1
    sys.stdout.write(line.decode())
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", choices=["fp32", "f32x3", "bf16"], default="fp32")
    ap.add_argument("--batch", type=int, default=500)
    ap.add_argument("--chunks", type=int, default=5)
    ap.add_argument("--decode-n", type=int, default=1000000,
                    help="latents to decode IN TOTAL (BASELINE.json config 4: 1 M), sharded over the ranks and decoded in slabs of "
                         "65,536 per GPU; token ids stay uint8 on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-extra-modes", action="store_true", help="skip the secondary timings in the other GEMM arithmetics")
    ap.add_argument("--no-rnn", action="store_true", help="skip the secondary figures of the recurrent models")
    ap.add_argument("--no-roofline", action="store_true", help="skip the roofline launches (profiles of the train step alone)")
    ap.add_argument("--no-config1", action="store_true", help="skip the trans1x-128 step (BASELINE.json configs[0])")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run nothing but the launches of the roofline kernel (warmed by time, then 50 timed) and print its roofline "
                         "object: under rocprofv3 --kernel-trace --stats the profile then holds exactly these launches")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    # stdout carries ONE JSON line and nothing else: libraries that print there from native code (RCCL's version banner
    # on communicator creation, gloo's connection notes) are sent to stderr for the whole run, and the line is written
    # to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # TVAE_BENCH_REHEARSE=1: every rank on cuda:0 over gloo -- a functional rehearsal of the N > 1 path on a
    # one-GPU box (RCCL refuses two ranks on one device); the numbers of such a run mean nothing
    rehearse = os.environ.get("TVAE_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    # TVAE_BENCH_FORCE_DP=1: run the data-parallel code path (RCCL process group, gradient arena, bucketed
    # all-reduce, final gather) even with ONE rank -- the only way to exercise the real RCCL calls on a one-GPU box
    force_dp = os.environ.get("TVAE_BENCH_FORCE_DP") == "1"
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher and the flag disagree\n" % (args.gpus, world))
        sys.exit(2)
    torch.cuda.set_device(local)
    use_dist = world > 1 or force_dp
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_dp and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    dev = torch.device("cuda", local)

    from transvae_amd import ops, synth
    ops.set_compute_dtype(args.dtype)
    if args.roofline_only:
        if rank == 0:
            emit({"roofline": roofline_of(args.dtype, args.batch * 127, 1024, 256)})
        return
    vae = build_vae(batch=args.batch, chunks=args.chunks)
    vae.model.train()                                   # dropout on: what scripts/train.py runs
    ops.seed_rng(dev, 1234 + rank)
    dp = None
    if use_dist:
        from transvae_amd.parallel import DataParallel
        dp = DataParallel(vae)
        dp.timing = True                 # GPU time the compute stream spends waiting for the bucket all-reduces
    toks = synth.synth_tokens(args.batch, seed=100 + rank).float()
    data = torch.cat([toks, torch.zeros(args.batch, 1)], 1).to(dev)     # resident in HBM
    beta = 0.01

    def step():
        return vae.train_step(data, beta, dp=dp)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(t):
        tmax = torch.tensor([t], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    def timed(n_warm, n_steps):
        for _ in range(n_warm):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            tab = step()
        fence()
        dt = time.perf_counter() - t0
        mine = dt
        if use_dist:
            dt = max_over_ranks(dt)
        return dt, tab, mine

    if dp is not None:
        dp.wait_events = []
    dt, table, dt_mine = timed(args.warmup, args.steps)
    dp_report = None
    if use_dist:
        # per-rank step time and the time each rank's compute stream stood waiting in dp.finish() for its buckets
        waits = [a.elapsed_time(b) for a, b in dp.wait_events[-args.steps:]]
        mine = torch.tensor([1e3 * dt_mine / args.steps, sum(waits) / max(len(waits), 1)], dtype=torch.float64,
                            device="cpu" if rehearse else dev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dp_report = {"ms_per_step_per_rank": [float(t[0]) for t in allr],
                     "allreduce_wait_ms_per_step_per_rank": [float(t[1]) for t in allr],
                     "gradient_bytes_per_step": int(dp.arena.flat.numel() * 4), "buckets": len(dp.arena.buckets),
                     "backend": dist.get_backend()}
        dp.timing = False
    last_loss = float(table[:, 0].mean().item())
    tokens = world * args.batch * 126 * args.steps
    value = tokens / dt
    nonpad = int((toks[:, 1:] != 24).sum())          # target positions that are not padding (rank 0's batch; every rank's has the same length law)

    out = {
        "metric": "train tokens/s (fwd+bwd) + decoded mols/s, trans4x-256, 1/2/4/8 MI355X",
        "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"fp32": "f32", "f32x3": "f32 (GEMM products as 3xbf16 splits, fp32 accumulate)", "bf16": "bf16"}[args.dtype], "data": "synthetic",
        "config": {"workload": "trans4x-256 ZINC-vocab train step (fwd+loss+bwd+Noam/Adam), batch %d per GPU in one pass with the "
                               "reference's %d x %d chunk-wise loss normalisation, max_len 126, dropout 0.1, synthetic "
                               "SMILES-shaped tokens, synthetic checkpoint"
                               % (args.batch, args.chunks, args.batch // args.chunks),
                   "decode_workload": "KV-cached greedy decode of %d latents z ~ N(0, I) in total (BASELINE.json config 4), %d per GPU "
                                      "in slabs of up to 65,536, token ids kept as uint8 on the device"
                                      % (args.decode_n, args.decode_n // world),
                   "global_batch": world * args.batch, "seq_len": 126,
                   "parallelism": "dp%d" % world if world > 1 else "single"},
        "value_nonpad": world * nonpad * args.steps / dt, "nonpad_unit": "non-pad target tokens/s (secondary, SURVEY.md 8d: %d of "
                        "the %d target positions of a batch carry a token or <end>)" % (nonpad, args.batch * 126),
        "final_loss": last_loss,
        "train_streams": 1 if os.environ.get("TVAE_WGRAD_STREAM", "1") == "0" else 2,   # weight-gradient GEMMs run beside the backward pass
        "data_parallel": dp_report,
        "train_algorithmic_tflops": world * args.batch * FLOP_PER_MOL_FWD_BWD * args.steps / dt / 1e12,
        "train_frac_of_mfma_peak": (args.batch * FLOP_PER_MOL_FWD_BWD * args.steps / dt / 1e12) / PEAK_TFLOPS[args.dtype],
        "train_flop_count": "algorithmic-dense: 5.940 GFLOP per molecule (SURVEY.md 8d) counts every attention tile; the kernels "
                            "skip tiles that are provably all zero on padded molecules (attention is 8 % of the count)",
    }

    # ---- the same step in the other GEMM arithmetics (secondary numbers; `value` is args.dtype) -----
    if not args.no_extra_modes:
        others = {}
        for mode in ("fp32", "f32x3", "bf16"):
            if mode == args.dtype:
                continue
            ops.set_compute_dtype(mode)
            dto, _, _ = timed(2, args.steps)
            others[mode] = {"value": tokens / dto, "unit": "tokens/s", "ms_per_step": 1e3 * dto / args.steps,
                            "parity": "fp32 tolerances (tests/test_model_gpu.py runs in this mode too)" if mode != "bf16"
                                      else "1e-2 (operands rounded to bf16)"}
        ops.set_compute_dtype(args.dtype)
        out["other_gemm_modes"] = others

    # ---- greedy decode (KV-cached), sharded by latent batch: the --decode-n latents are dealt to the ranks, every rank
    # decodes its share in slabs of 65,536 (VAEShell.decode_latents), ids stay uint8 on the device; one final gather ------
    if not args.no_decode:
        vae.model.eval()
        share = args.decode_n // world + (1 if rank < args.decode_n % world else 0)
        z = torch.randn(share, 128, generator=torch.Generator().manual_seed(rank)).to(dev)
        vae.greedy_decode(z[:64])
        fence()
        t0 = time.perf_counter()
        ids = vae.decode_latents(z)
        fence()
        td = time.perf_counter() - t0
        if use_dist:
            td = max_over_ranks(td)
            shares = [args.decode_n // world + (1 if r < args.decode_n % world else 0) for r in range(world)]
            small = torch.zeros((max(shares), ids.shape[1]), dtype=torch.uint8, device=dev)   # vocabulary of 26: one byte per token
            small[:share] = ids
            if rehearse:
                small = small.cpu()
            gathered = [torch.empty_like(small) for _ in range(world)] if rank == 0 else None
            dist.gather(small, gathered, dst=0)                        # the one exchange: final token ids
        out["decode_mols_per_s"] = args.decode_n / td
        out["decode_latents_total"] = args.decode_n
        out["decode_latents_per_gpu"] = share
        out["decode_slab"] = min(share, vae.DECODE_SLAB)
        out["decode_streams"] = int(os.environ.get("TVAE_DECODE_STREAMS", "4"))      # pieces of a slab decoded side by side
        out["decode_seconds"] = td
        out["decode_scaling"] = "strong (the total is fixed: BASELINE.json config 4 decodes 1 M latents, sharded by rank)"
        out["decode_ids_bytes"] = int(ids.numel())                      # uint8 on the device
        out["decode_algorithmic_tflops"] = args.decode_n * FLOP_PER_MOL_DECODE / td / 1e12
        if not args.no_extra_modes and args.dtype == "fp32":
            # one slab with the fp32-accurate 3xbf16 GEMMs beside the same slab in exact fp32 (two fp32-accurate arithmetics
            # round differently: among millions of arg-max decisions a few near-ties -- top-2 logit margin below fp32
            # resolution -- fall the other way and the row diverges from there)
            n3 = min(share, vae.DECODE_SLAB)
            ops.set_compute_dtype("f32x3")
            vae.greedy_decode(z[:64])
            fence()
            t0 = time.perf_counter()
            ids3 = vae.decode_latents(z[:n3])
            fence()
            t3 = time.perf_counter() - t0
            if use_dist:
                t3 = max_over_ranks(t3)
            ops.set_compute_dtype(args.dtype)
            out["decode_other_gemm_modes"] = {"f32x3": {"value": world * n3 / t3, "unit": "molecules/s", "latents_per_gpu": n3,
                                                        "rows_identical_to_fp32": float((ids3 == ids[:n3]).all(dim=1).float().mean().item())}}
        del z, ids
        torch.cuda.empty_cache()                                        # the slab's K/V caches (~110 GB) go back before the rest

    if rank == 0:
        # ---- roofline of the dominant kernel: the MFMA GEMM on the FFN w_1 launch shape -----------
        M, N, K = args.batch * 127, 1024, 256        # the whole batch goes through in one pass

        def roofline(mode):
            return roofline_of(mode, M, N, K)

        if not args.no_roofline:
            out["roofline"] = roofline(args.dtype)
            out["roofline_attention"] = roofline_attention_of(args.batch)
            if not args.no_decode:
                out["roofline_decode"] = roofline_decode_of(65536)
            if not args.no_extra_modes and args.dtype != "f32x3":
                out["roofline_f32x3"] = roofline("f32x3")
        ops.set_compute_dtype(args.dtype)
        if not args.no_rnn and not args.no_extra_modes and world == 1:
            del vae
            torch.cuda.empty_cache()
            out["recurrent_models"] = [rnn_secondary("rnnattn", 256), rnn_secondary("rnn", 128)]
        cpu = not args.no_cpu_baseline and world == 1
        threads = host_threads()
        if not args.no_config1 and world == 1:
            out["config1"] = config1_of(args.steps, threads if cpu else None)
        if cpu:
            out["cpu_baseline"] = cpu_baseline(threads)
            out["cpu_baseline_decode"] = cpu_decode_baseline(threads)
        emit(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
