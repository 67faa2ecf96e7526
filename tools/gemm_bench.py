#!/usr/bin/env python3
"""Micro-benchmark of tvae_gemm on the launch shapes of the trans4x-256 train step (HIP events)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transvae_amd import ops, _lib

def bench(name, fn, flops, iters=20):
    # an idle MI355X takes ~50 ms of load to reach its working clock: warm up by time, not by launch count
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < 0.15:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters
    print("%-46s %9.1f us  %7.1f TFLOP/s" % (name, t * 1e6, flops / t / 1e12), flush=True)

def main():
    dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    ops.set_compute_dtype(dt)
    dev = "cuda"
    M = 63500
    rng = ops.rng_state(torch.device("cuda", 0))
    shapes = [(1024, 256), (256, 1024), (256, 256), (768, 256), (512, 256)]
    if len(sys.argv) > 3:
        shapes = [tuple(int(v) for v in sys.argv[3].split("x"))]
    for (N, K) in shapes:
        x = torch.randn(M, K, device=dev); w = torch.nn.Parameter(torch.randn(N, K, device=dev) * 0.05); b = torch.zeros(N, device=dev)
        y = ops.rows_buffer(M, N, dev); res = torch.randn(M, N, device=dev); dy = torch.randn(M, N, device=dev)   # y: as the model allocates it
        fl = 2.0 * M * N * K
        tag = "M%d N%d K%d" % (M, N, K)
        drop = ops.Drop(0.1, 5, rng)
        if only in ("", "nt"):
            bench("NT  fwd bias            " + tag, lambda: ops.linear_fwd(x, w, b, out=y), fl)
            bench("NT  fwd bias relu drop  " + tag, lambda: ops.linear_fwd(x, w, b, relu=True, drop=drop, out=y), fl)
            bench("NT  fwd bias drop resid " + tag, lambda: ops.linear_fwd(x, w, b, residual=res, drop=drop, out=y), fl)
        if only in ("", "nn"):
            dx = ops.rows_buffer(M, K, dev)
            bench("NN  dgrad               " + tag, lambda: ops.linear_dgrad(dy, w, out=dx), fl)
            bench("NN  dgrad a_drop        " + tag, lambda: ops.linear_dgrad(dy, w, out=dx, a_drop=drop), fl)
        if only in ("", "tn"):
            bench("TN  wgrad (+bias grad)  " + tag, lambda: ops.linear_wgrad(dy, x), fl)
            bench("TN  wgrad a_drop        " + tag, lambda: ops.linear_wgrad(dy, x, a_drop=drop), fl)

if __name__ == "__main__":
    main()
