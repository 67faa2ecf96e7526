#!/usr/bin/env python3
"""Train step and greedy decode of the recurrent models on one MI355X:  python tools/rnn_bench.py [rnnattn|rnn] [d_model] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transvae_amd import ops, rnn_models, synth

def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "rnnattn"
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    cd, od, w = synth.load_vocab("zinc")
    cls = {"rnnattn": rnn_models.RNNAttn, "rnn": rnn_models.RNN}[kind]
    vae = cls(params={"CHAR_DICT": cd, "ORG_DICT": od, "CHAR_WEIGHTS": w, "BATCH_SIZE": batch, "BATCH_CHUNKS": 5}, d_model=d)
    vae.model.load_state_dict(synth.synth_state_dict(vae.model.state_dict(), 0))
    vae.model.train()
    dev = torch.device("cuda", 0)
    ops.seed_rng(dev, 1)
    data = torch.cat([synth.synth_tokens(batch, seed=1).float(), torch.zeros(batch, 1)], 1).to(dev)
    for _ in range(3):
        vae.train_step(data, 0.01)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        vae.train_step(data, 0.01)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%s-%d train step, batch %d: %.2f ms  %.0f tokens/s" % (kind, d, batch, dt * 1e3, batch * 126 / dt), flush=True)
    z = torch.randn(int(os.environ.get("RNN_DECODE_N", "16384")), 128).to(dev)
    vae.greedy_decode(z[:64])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vae.greedy_decode(z)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s-%d greedy decode of %d latents: %.3f s  %.0f mol/s" % (kind, d, z.shape[0], dt, z.shape[0] / dt), flush=True)

if __name__ == "__main__":
    main()
