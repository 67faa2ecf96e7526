#!/usr/bin/env python3
"""Greedy-decode throughput at small latent batches (what scripts/sample.py's default of 100 per batch runs):
plain launches beside the replayed HIP graph of the same step loop."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_vae  # noqa: E402
from transvae_amd import decode  # noqa: E402


def rate(vae, z, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        vae.greedy_decode(z)
    torch.cuda.synchronize()
    return reps * z.shape[0] / (time.perf_counter() - t0)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [100, 500, 2000]
    vae = build_vae()
    vae.model.eval()
    vae.greedy_decode(torch.randn(64, 128))
    for n in sizes:
        z = torch.randn(n, 128, generator=torch.Generator().manual_seed(0)).cuda()
        limit, decode.GRAPH_MAX_N = decode.GRAPH_MAX_N, 0
        vae.greedy_decode(z)
        plain = rate(vae, z, 3)
        decode.GRAPH_MAX_N = limit
        vae.greedy_decode(z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        vae.greedy_decode(z)                                   # second call: capture + first replay
        torch.cuda.synchronize()
        t_cap = time.perf_counter() - t0
        graphed = rate(vae, z, 5)
        print("n=%5d  plain %8.0f mol/s   graph %8.0f mol/s   (capture call %.2f s)" % (n, plain, graphed, t_cap), flush=True)


if __name__ == "__main__":
    main()
