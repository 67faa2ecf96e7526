#!/usr/bin/env python3
"""Where the HOST time of a trans1x-128 train step goes (that step is bound by the launch rate): cProfile over 30 steps,
backward run on the calling thread (set_multithreading_enabled(False)) so that one profiler sees both passes, top functions by own time.
python tools/debug/host_profile_config1.py [model]"""
import os, sys, cProfile, pstats, threading, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, torch
import bench
from transvae_amd import synth
vae = bench.build_vae(sys.argv[1] if len(sys.argv) > 1 else "trans1x-128")
vae.model.train()
data = torch.cat([synth.synth_tokens(500, seed=100).float(), torch.zeros(500, 1)], 1).cuda()
for _ in range(5):
    vae.train_step(data, 0.02)
torch.cuda.synchronize()
profs = {}
t0 = time.perf_counter()
for _ in range(30):
    vae.train_step(data, 0.02)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print("%.3f ms/step queued by the host, %.3f ms/step with the GPU drained (no profiler)" % (th / 30 * 1e3, (time.perf_counter() - t0) / 30 * 1e3))
torch.autograd.set_multithreading_enabled(False)
main = cProfile.Profile()
t0 = time.perf_counter()
main.enable()
for _ in range(30):
    vae.train_step(data, 0.02)
main.disable()
th = time.perf_counter() - t0
torch.cuda.synchronize()
print("%.3f ms/step queued by the host, %.3f ms/step with the GPU drained (under the profiler)" % (th / 30 * 1e3, (time.perf_counter() - t0) / 30 * 1e3))
for name, p in [("main thread", main)] + [("thread %d" % t, p) for t, p in profs.items()]:
    try:
        p.disable()
    except Exception:
        pass
    out = io.StringIO()
    try:
        pstats.Stats(p, stream=out).sort_stats("tottime").print_stats(45)
    except Exception as e:
        print(name, "no stats:", e)
        continue
    print("=====", name)
    print("\n".join(l[:170] for l in out.getvalue().splitlines()[4:60]))
