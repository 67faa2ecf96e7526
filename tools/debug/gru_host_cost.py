#!/usr/bin/env python3
"""Host cost of the per-time-step launches: time for tvae_gru_fwd / tvae_gru_bwd to RETURN (127 launches each) vs GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from transvae_amd import ops
T, B, H = 127, 500, 256
dev = "cuda"
gi = torch.randn(T * B, 3 * H, device=dev); w = torch.randn(3 * H, H, device=dev) * 0.05; b = torch.zeros(3 * H, device=dev)
hb = torch.zeros((T + 1) * B, H, device=dev); gates = torch.empty(T * B, 4 * H, device=dev)
dgi = torch.empty(T * B, 3 * H, device=dev); dgh = torch.empty(T * B, 3 * H, device=dev); carry = torch.empty(B, H, device=dev)
d = torch.randn(T * B, H, device=dev); wt = w.t().contiguous()
def fwd():
    ops.call("tvae_gru_fwd", ops._p(gi), ops._p(w), ops._p(b), ops._p(hb), ops._p(hb[B:]), ops._p(gates), T, B, H, 1, ops._stream())
def bwd():
    ops.call("tvae_gru_bwd", ops._p(d), None, ops._p(hb[B:]), ops._p(gates), ops._p(hb), ops._p(wt), ops._p(dgi), ops._p(dgh), ops._p(carry), T, B, H, 1, 0, ops._stream())
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host returns after %.1f us per launch, GPU done after %.1f us per launch" % (name, (t1 - t0) / 1270 * 1e6, (t2 - t0) / 1270 * 1e6))
