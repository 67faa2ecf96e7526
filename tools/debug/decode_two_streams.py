#!/usr/bin/env python3
"""Experiment: greedy decode of 131072 latents as slabs of 65536 one after the other (decode_latents as it is) against
slabs of 32768 driven by TWO host threads on two streams: the one-query attention over the K/V caches is HBM-bound, the
projections MFMA-bound, so two independent slabs should overlap.  python tools/debug/decode_two_streams.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda:0")
vae = bench.build_vae()
vae.model.eval()
N = 131072
g = torch.Generator(device="cpu").manual_seed(5)
mem = torch.randn(N, 128, generator=g).to(dev)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


def two_threads(slab):
    out = torch.empty((N, vae.tgt_len), dtype=torch.uint8, device=dev)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    main = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(main)

    def work(k):
        with torch.cuda.stream(streams[k]), torch.no_grad():
            lo, hi = k * (N // 2), (k + 1) * (N // 2)
            for s0 in range(lo, hi, slab):
                out[s0:s0 + slab] = vae.greedy_decode(mem[s0:s0 + slab]).to(torch.uint8)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for s in streams:
        main.wait_stream(s)
    return out


vae.decode_latents(mem[:4096])            # warm
t1, ref = timed(lambda: vae.decode_latents(mem, slab=65536))
print("one stream, slabs of 65536:            %.3f s  %.1f k molecules/s" % (t1, N / t1 / 1e3), flush=True)
t2, got = timed(lambda: two_threads(32768))
print("two threads / streams, slabs of 32768: %.3f s  %.1f k molecules/s   identical tokens: %s" % (t2, N / t2 / 1e3, bool((got == ref).all())), flush=True)
t3, ref3 = timed(lambda: vae.decode_latents(mem, slab=32768))
print("one stream, slabs of 32768:            %.3f s  %.1f k molecules/s" % (t3, N / t3 / 1e3), flush=True)
