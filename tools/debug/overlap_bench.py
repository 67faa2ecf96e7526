#!/usr/bin/env python3
"""Does a weight-gradient GEMM (split-K, tiny epilogue) overlap a data-gradient GEMM (heavy epilogue) when the two are
launched on two streams?  Times N back-to-back (dgrad, wgrad) pairs of a Linear layer serially on one stream and
concurrently on two, with events around the whole batch.  python tools/debug/overlap_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from transvae_amd import ops

dev = "cuda"
KM = ops.GEMM_A_KMAJOR | ops.GEMM_B_KMAJOR


def pair(M, N, K):
    "Linear (K -> N) on M rows: dgrad dx[M,K] = dy[M,N] W[N,K]; wgrad dW[N,K] = dy^T x"
    dy = torch.randn(M, N, device=dev)
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    dx = torch.empty(M, K, device=dev)
    dW = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)

    def dgrad():
        ops.gemm(dy, N, W, K, dx, K, M, K, N, flags=ops.GEMM_B_KMAJOR)

    def wgrad():
        ops.gemm(dy, N, x, K, dW, K, N, K, M, flags=KM | ops.GEMM_SPLITK, a_rowsum=db)
    return dgrad, wgrad


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


side = torch.cuda.Stream()
for M, N, K in [(63000, 256, 256), (63500, 768, 256), (63500, 1024, 256), (63500, 256, 1024)]:
    dgrad, wgrad = pair(M, N, K)
    t_d = timed(dgrad, 20)
    t_w = timed(wgrad, 20)

    def serial():
        dgrad()
        wgrad()

    def overlapped():
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            wgrad()
        dgrad()
        ev2 = torch.cuda.Event()
        ev2.record(side)
        main.wait_event(ev2)

    def overlapped_lag():
        # no join inside the pair: the side stream runs one wgrad behind, as it would in a backward pass
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            wgrad()
        dgrad()

    t_s = timed(serial, 20)
    t_o = timed(overlapped, 20)
    t_l = timed(overlapped_lag, 20)
    torch.cuda.synchronize()
    print("Linear %4d -> %4d on %d rows: dgrad %6.1f  wgrad %6.1f  serial pair %6.1f  two streams (joined) %6.1f  two streams (lagging) %6.1f us"
          % (K, N, M, t_d, t_w, t_s, t_o, t_l))
