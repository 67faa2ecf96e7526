#!/usr/bin/env python3
"""Is the rnnattn-256 train step bound by the host's launch rate?  Time to ENQUEUE n steps vs time until the GPU is done."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from transvae_amd import ops, rnn_models, synth
cd, od, w = synth.load_vocab("zinc")
vae = rnn_models.RNNAttn(params={"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w, "BATCH_SIZE": 500, "BATCH_CHUNKS": 5}, d_model=256)
vae.model.train()
dev = torch.device("cuda", 0)
ops.seed_rng(dev, 1)
data = torch.cat([synth.synth_tokens(500, seed=1).float(), torch.zeros(500, 1)], 1).to(dev)
for _ in range(3):
    vae.train_step(data, 0.01)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    vae.train_step(data, 0.01)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("chunk=%s: host enqueue %.2f ms per step, GPU done after %.2f ms per step" % (os.environ.get("TVAE_GRU_CHUNK", "16"), (t1 - t0) * 100, (t2 - t0) * 100))
