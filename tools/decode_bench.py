#!/usr/bin/env python3
"""Greedy-decode throughput of trans4x-256 at several latent batch sizes (wall clock, one warm-up)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_vae, FLOP_PER_MOL_DECODE

def main():
    sizes = [int(a) for a in sys.argv[1:]] or [4096, 16384]
    vae = build_vae()
    vae.model.eval()
    vae.greedy_decode(torch.randn(64, 128))
    for n in sizes:
        z = torch.randn(n, 128, generator=torch.Generator().manual_seed(0)).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ids = vae.greedy_decode(z)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("n=%6d  %.3f s  %9.0f mol/s  %6.1f algorithmic TFLOP/s  mem %.1f GB" %
              (n, dt, n / dt, n * FLOP_PER_MOL_DECODE / dt / 1e12, torch.cuda.max_memory_allocated() / 2**30), flush=True)

if __name__ == "__main__":
    main()
