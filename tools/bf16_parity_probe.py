#!/usr/bin/env python3
"""bf16 mode: how far are the gradients from the fp64 truth when the weight-gradient and/or data-gradient products run
in another arithmetic?  Golden step inputs, every tensor against the fp64 oracle.  -> stdout table."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import CONFIGS, load, oracle_params, rel_l2  # noqa: E402
from test_model_gpu import build, run_chunk  # noqa: E402
from oracle import tvae_oracle as O  # noqa: E402
from transvae_amd import ops, synth  # noqa: E402

for cfg_name in sys.argv[1:] or ["1x128_zinc", "4x256_zinc", "4x256_pubchem_pp"]:
    g = load("step_" + cfg_name)
    toks = torch.from_numpy(g["tokens"].astype(np.int64))
    props, eps = torch.from_numpy(g["props"]), torch.from_numpy(g["eps"])
    P = oracle_params(cfg_name, torch.float64, requires_grad=True)
    w = torch.tensor(synth.load_vocab(CONFIGS[cfg_name]["vocab"])[2], dtype=torch.float64)
    losses, _ = O.chunk_loss(P, toks, props.double(), w, float(g["beta"]), eps=eps.double())
    losses[0].backward()
    for wg, dg in ((None, None), ("f32x3", None), (None, "f32x3"), ("f32x3", "f32x3")):
        ops.set_compute_dtype("bf16", wgrad=wg, dgrad=dg)
        vae = build(cfg_name)
        vae.model.eval()
        out, _ = run_chunk(vae, toks, props, eps, float(g["beta"]))
        out[0].backward()
        errs = {}
        for k, p in vae.model.named_parameters():
            ref = P[k].grad.numpy()
            if np.linalg.norm(ref) < 1e-10:
                continue
            errs[k] = rel_l2(p.grad.cpu(), ref)
        v = np.array(sorted(errs.values()))
        worst = max(errs, key=errs.get)
        print("%-18s wgrad=%-6s dgrad=%-6s rel-L2 all %d tensors: median %.2e  90%% %.2e  worst %.2e (%s)  within 1e-2: %d"
              % (cfg_name, wg, dg, len(v), np.median(v), v[int(0.9 * len(v))], v[-1], worst, int((v < 1e-2).sum())), flush=True)
