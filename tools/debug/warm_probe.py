#!/usr/bin/env python3
"""Is the fp32 GEMM's rate a function of how long the GPU has been busy?  Same launch, timed again and again."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from transvae_amd import ops, _lib

def t_of(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters

def main():
    ops.set_compute_dtype("fp32")
    M, K, N = 63500, 256, 1024
    x = torch.randn(M, K, device="cuda")
    w = torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.05); b = torch.zeros(N, device="cuda")
    y = torch.empty((M + 128) * N, device="cuda")
    y2 = ops.rows_buffer(M, N, "cuda")
    direct = lambda: ops.gemm(x, K, w, K, y, N, M, N, K, bias=b, flags=_lib.GEMM_RELU)
    viaop = lambda: ops.linear_fwd(x, w, b, out=y2)
    t0 = time.time()
    for rep in range(12):
        td = t_of(direct); tv = t_of(viaop)
        print("t=%6.3f s  direct %7.1f us  linear_fwd %7.1f us" % (time.time() - t0, td * 1e6, tv * 1e6), flush=True)
    torch.cuda.synchronize(); time.sleep(2.0)
    print("after 2 s idle")
    for rep in range(4):
        td = t_of(direct); tv = t_of(viaop)
        print("t=%6.3f s  direct %7.1f us  linear_fwd %7.1f us" % (time.time() - t0, td * 1e6, tv * 1e6), flush=True)
    # different data: zeros, small, large
    for name, xx in (("zeros", torch.zeros_like(x)), ("randn*1e-3", x * 1e-3), ("randn", x), ("randn*100", x * 100), ("relu-like (half zeros)", torch.relu(x))):
        f = lambda: ops.gemm(xx, K, w, K, y, N, M, N, K, bias=b, flags=_lib.GEMM_RELU)
        t_of(f, 20)
        print("%-24s %7.1f us" % (name, t_of(f, 20) * 1e6), flush=True)

if __name__ == "__main__":
    main()
