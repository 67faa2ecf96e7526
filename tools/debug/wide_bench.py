#!/usr/bin/env python3
"""The conv-bottleneck products with N = 148 / 160 on the 128 x 160 tile kernel (TVAE_GEMM_WIDE=1, default) or on the
128 x 128 kernel (TVAE_GEMM_WIDE=0): warm launches timed with events.  python tools/debug/wide_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from transvae_amd import ops

KM = ops.GEMM_A_KMAJOR | ops.GEMM_B_KMAJOR
shapes = [("NT conv0 fwd", 63492, 160, 2304, 256, 0), ("NT deconv2 dgrad", 68500, 148, 2816, 256, 0),
          ("NT conv1 dgrad", 29500, 160, 896, 112, 0), ("TN conv0 wgrad (swapped)", 2304, 160, 63492, 256, KM | ops.GEMM_SPLITK),
          ("TN conv0 wgrad (as before)", 160, 2304, 63492, 160, KM | ops.GEMM_SPLITK),
          ("NT deconv1 poly fwd", 21997, 444, 448, 112, 0)]
print("TVAE_GEMM_WIDE =", os.environ.get("TVAE_GEMM_WIDE", "1"))
for name, M, N, K, lda, fl in shapes:
    if fl & ops.GEMM_A_KMAJOR:
        a = torch.randn(K * lda + M + 64, device="cuda")
        b = torch.randn(K, N, device="cuda")
        ldb = N if name.startswith("TN conv0 wgrad (sw") else 256
        if ldb != N:
            b = torch.randn(K * ldb + N + 64, device="cuda")
    else:
        a = torch.randn(M * lda + K + 64, device="cuda")
        b = torch.randn(N, K, device="cuda") * 0.05
        ldb = K
    c = torch.zeros(M, N, device="cuda")
    bias = None if fl else torch.zeros(N, device="cuda")
    t = bench.time_launches(lambda: ops.gemm(a, lda, b, ldb, c, N, M, N, K, bias=bias, flags=fl | (0 if fl else ops.GEMM_RELU)), 20)
    print("%-28s %6d x %4d x %6d  %8.1f us  %6.1f TFLOP/s" % (name, M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12))
