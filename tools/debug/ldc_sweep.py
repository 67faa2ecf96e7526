#!/usr/bin/env python3
"""Does the output row stride matter?  x[63500,256] @ W[N,256]^T + bias for several N and ldc (fp32 kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from transvae_amd import ops, _lib

def t_of(fn, iters=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters

def main():
    ops.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "fp32")
    M, K = 63500, 256
    x = torch.randn(M, K, device="cuda")
    for N in (512, 768, 896, 1024, 1152, 1280, 2048):
        w = torch.randn(N, K, device="cuda") * 0.05; b = torch.zeros(N, device="cuda")
        for pad in (0, 32, 64, 256):
            ldc = N + pad
            y = torch.empty((M + 128) * ldc, device="cuda")
            t = t_of(lambda: ops.gemm(x, K, w, K, y, ldc, M, N, K, bias=b, flags=_lib.GEMM_RELU))
            print("N %5d ldc %5d  %8.1f us  %6.1f TFLOP/s" % (N, ldc, t * 1e6, 2.0 * M * N * K / t / 1e12), flush=True)

if __name__ == "__main__":
    main()
