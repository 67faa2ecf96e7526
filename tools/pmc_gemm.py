#!/usr/bin/env python3
"""Launch the roofline GEMM of bench.py (x[63500,256] @ W[1024,256]^T + bias, ReLU; argv[1] = fp32 | f32x3 | bf16) a few times;
run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) to count its HBM traffic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transvae_amd import ops

M, N, K = 63500, 1024, 256
ops.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "fp32")
x = torch.randn(M, K, device="cuda"); w = torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.05, requires_grad=False)
b = torch.zeros(N, device="cuda")
out = ops.rows_buffer(M, N, "cuda")
for _ in range(8):
    ops.linear_fwd(x, w, b, relu=True, out=out)
torch.cuda.synchronize()
