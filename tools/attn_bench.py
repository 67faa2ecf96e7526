#!/usr/bin/env python3
"""Micro-benchmark of the fused attention kernels on the trans4x-256 train shapes (B = 500, h = 4, dk = 64),
with the key masks of the SMILES-shaped synthetic batch (bench.py's tokens) and with no padding at all."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transvae_amd import ops, synth


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    if len(sys.argv) > 1:                      # python tools/attn_bench.py bf16: the bf16-MFMA instantiations
        ops.set_compute_dtype(sys.argv[1])
    B, h, dk = 500, 4, 64
    d = h * dk
    toks = synth.synth_tokens(B, seed=100)
    rng = ops.rng_state(torch.device("cuda", 0))
    drop = ops.Drop(0.1, 3, rng)
    for name, Lq, Lk, causal in [("enc self", 127, 127, False), ("dec self (causal)", 126, 126, True), ("dec src", 126, 127, False)]:
        for pad in ("smiles", "none"):
            km = (toks[:, :Lk] != 24) if pad == "smiles" else torch.ones(B, Lk, dtype=torch.bool)
            spec = ops.MaskSpec(key_mask=km.to(torch.uint8).cuda(), causal=causal)
            q, k, v, do = (torch.randn(B * L, d, device="cuda") for L in (Lq, Lk, Lk, Lq))
            o, lse, _ = ops.attn_fwd(q, k, v, B, h, Lq, Lk, dk, spec, drop)
            tf = timeit(lambda: ops.attn_fwd(q, k, v, B, h, Lq, Lk, dk, spec, drop))
            tb = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, h, Lq, Lk, dk, spec, drop))
            print("%-18s keys %-6s (mean visible %5.1f)  fwd %6.1f us   bwd (dQ + dK/dV) %6.1f us" %
                  (name, pad, km.float().sum(1).mean().item(), tf, tb), flush=True)


if __name__ == "__main__":
    main()
