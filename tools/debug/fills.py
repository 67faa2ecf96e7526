"""Where do the elementwise fill / copy / add launches of a train step come from?  (torch profiler, python stacks)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import build_vae
from transvae_amd import synth, ops
vae = build_vae(); vae.model.train()
toks = synth.synth_tokens(500, seed=100).float()
data = torch.cat([toks, torch.zeros(500, 1)], 1).cuda()
for _ in range(2): vae.train_step(data, 0.01)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    vae.train_step(data, 0.01)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::zeros", "aten::zeros_like", "aten::contiguous", "aten::clone", "aten::cat", "aten::to", "aten::_to_copy"):
        st = [s for s in (ev.stack or []) if "transvae_amd" in s or "torch/autograd" in s][:2]
        cnt[(ev.name, " <- ".join(s.split("/")[-1][:60] for s in st))] += 1
for (name, where), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print("%4d %-18s %s" % (c, name, where))
