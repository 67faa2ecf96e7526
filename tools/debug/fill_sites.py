#!/usr/bin/env python3
"""Which call sites launch the fill / copy / add kernels of one fp32 train step (trans4x-256, batch 500): torch.profiler
with python stacks, aten::fill_ / zero_ / copy_ / add_ / cat events grouped by their innermost frame inside this repo.
python tools/debug/fill_sites.py"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from transvae_amd import synth

vae = bench.build_vae()
vae.model.train()
toks = synth.synth_tokens(500, seed=100).float()
data = torch.cat([toks, torch.zeros(500, 1)], 1).cuda()
for _ in range(3):
    vae.train_step(data, 0.03)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    vae.train_step(data, 0.03)
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add_", "aten::add", "aten::cat", "aten::zeros", "aten::zeros_like",
        "aten::contiguous", "aten::clone", "aten::mul", "aten::sum")
sites = collections.Counter()
for ev in prof.events():
    if ev.name not in want:
        continue
    where = "(no python frame: autograd engine)"
    for fr in ev.stack:
        if "/repo/" in fr and "torch/" not in fr:
            where = fr.split("/repo/")[-1]
            break
    sites[(ev.name, where)] += 1
for (name, where), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print("%4d  %-18s %s" % (n, name, where))
