#!/usr/bin/env python3
"""scripts/sample.py at BASELINE.json config 4's size on one GPU: a synthetic trans4x-256 checkpoint is written, then the
CLI entry point samples --n_samples molecules the way a user would run it (latents drawn per batch of 100 in the
reference's order, decoded in slabs of 65,536, CSV written).  Prints one JSON line: wall time of the whole call and
molecules/s.  python tools/sample_bench.py [n_samples] [n_samples_per_batch]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from scripts import parsers, sample

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
per = int(sys.argv[2]) if len(sys.argv) > 2 else 100
tmp = tempfile.mkdtemp()
vae = bench.build_vae()
vae.current_state["model_state_dict"] = vae.model.state_dict()
ckpt = os.path.join(tmp, "bench4x256.ckpt")
vae.save(vae.current_state, ckpt, use_name=False)
del vae
torch.cuda.empty_cache()
out = os.path.join(tmp, "gen.csv")
args = parsers.sample_parser().parse_args(["--model", "transvae", "--model_ckpt", ckpt, "--sample_mode", "rand", "--n_samples", str(n),
                                           "--n_samples_per_batch", str(per), "--save_path", out])
torch.manual_seed(0)
t0 = time.perf_counter()
sample.sample(args)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
rows = sum(1 for _ in open(out)) - 1
print(json.dumps({"command": "scripts/sample.py --model transvae --sample_mode rand --n_samples %d --n_samples_per_batch %d" % (n, per),
                  "molecules": rows, "seconds_whole_call": dt, "molecules_per_s_whole_call": rows / dt,
                  "includes": "checkpoint load, %d sample_from_memory draws, greedy decode in slabs of 65,536, decode_mols, CSV write" % (-(-n // per)),
                  "csv_bytes": os.path.getsize(out)}))
