#!/usr/bin/env python3
"""One GEMM shape, a few launches, for rocprofv3 --pmc passes: python pmc_shape.py <dtype> <N> <K> [nt|nn|tn]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transvae_amd import ops
dt, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "nt"
ops.set_compute_dtype(dt)
M = 63500
x = torch.randn(M, K, device="cuda"); w = torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.05, requires_grad=False)   # a weight, as in the model
b = torch.zeros(N, device="cuda")
y = ops.rows_buffer(M, N, "cuda"); dy = torch.randn(M, N, device="cuda")
n = int(os.environ.get("PMC_LAUNCHES", "6"))     # PMC_LAUNCHES=300: long enough (under counter collection every launch is
for _ in range(n):                               # serialised) for the chip to reach the clock it holds under sustained load
    if kind == "nt": ops.linear_fwd(x, w, b, out=y)
    elif kind == "nn": ops.linear_dgrad(dy, w)
    else: ops.linear_wgrad(dy, x)
torch.cuda.synchronize()
