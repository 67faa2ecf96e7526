#!/usr/bin/env python3
"""Does a power-of-two output row stride cost anything?  NT GEMM M63500 N1024 K256 with ldc = 1024 + pad."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transvae_amd import ops
from gemm_bench import bench

M, N, K = 63500, 1024, 256
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.zeros(N, device="cuda")
for pad in (0, 16, 32, 64, 128):
    y = torch.empty(M, N + pad, device="cuda")[:, :N]
    bench("NT fwd bias ldc=%d" % (N + pad), lambda: ops.linear_fwd(x, w, b, out=y), 2.0 * M * N * K)
for pad in (0, 32):
    xp = torch.randn(M, K + pad, device="cuda")[:, :K]
    y = torch.empty(M, N + 32, device="cuda")[:, :N]
    bench("NT fwd bias lda=%d ldc=%d" % (K + pad, N + 32), lambda: ops.linear_fwd(xp, w, b, out=y), 2.0 * M * N * K)
