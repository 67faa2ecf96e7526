"""Measure the data pipeline in front of the train step on the GPU box.

  encode   SMILES/s of tvae_encode_smiles (threaded C++) beside the oracle's Python restatement of the
           reference's tokenizer + encode_smiles + vae_data_gen loop
  batches  time per (500, 128) float batch: one gather launch from the uint8 dataset in HBM beside the
           reference's DataLoader (index + stack 500 rows on the host) followed by the H2D copy
  epoch    VAEShell.train on trans4x-256 (its log's run_time column: GPU time from step end to step end)
           beside the reference-shaped loop (DataLoader, loss table read back every step)

    python tools/data_bench.py [--mols 20000]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import tvae_oracle as O  # noqa: E402  (the CPU baseline of the encode leg, as in bench.py)
from transvae_amd import ops, synth  # noqa: E402
from transvae_amd.data import DeviceDataset, DeviceLoader, encode_smiles_batch, vae_data_gen  # noqa: E402
from transvae_amd.trans_models import TransVAE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mols", type=int, default=20000)
    args = ap.parse_args()
    out = {}
    g = np.load(os.path.join(ROOT, "tests", "golden", "data.npz"))
    base = [str(s) for s in g["zinc_smiles"]]
    cdict, org, w = synth.load_vocab("zinc")

    # ---- encode -------------------------------------------------------------------------------
    big = base * 1000                                           # 1 M real ZINC strings (1000 distinct)
    threads = min(16, os.cpu_count() or 1)
    encode_smiles_batch(big[:50000], cdict, n_threads=threads)
    t0 = time.perf_counter()
    ids = encode_smiles_batch(big, cdict, n_threads=threads)
    t_native = time.perf_counter() - t0
    t0 = time.perf_counter()
    rows = O.data_rows(big[:20000], None, cdict)
    t_py = time.perf_counter() - t0
    assert np.array_equal(ids[:20000].astype(np.float32), rows[:, :127])
    out["encode"] = {"native_smiles_per_s": len(big) / t_native, "threads": threads,
                     "python_restatement_smiles_per_s": 20000 / t_py,
                     "note": "native time includes building the text blob from the Python strings"}

    # ---- batches ------------------------------------------------------------------------------
    dev = torch.device("cuda:0")
    mols = np.array(big[:args.mols], dtype=object)[:, None]
    data = vae_data_gen(mols, None, cdict)
    ds = DeviceDataset.from_tensor(data, dev)
    ld = DeviceLoader(ds, 500)
    for _ in ld:
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for _ in range(5):
        for b in ld:
            n += 1
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / n
    dl = torch.utils.data.DataLoader(data, batch_size=500, shuffle=True, drop_last=True)
    t0 = time.perf_counter()
    n = 0
    for b in dl:
        b = b.to(dev)
        n += 1
    torch.cuda.synchronize()
    t_host = (time.perf_counter() - t0) / n
    out["batches"] = {"device_loader_us_per_batch": t_dev * 1e6, "dataloader_plus_h2d_us_per_batch": t_host * 1e6,
                      "batch": [500, 128]}

    # ---- epoch --------------------------------------------------------------------------------
    def build():
        torch.manual_seed(0)
        vae = TransVAE(params={"CHAR_DICT": cdict, "ORG_DICT": org, "CHAR_WEIGHTS": w, "BATCH_SIZE": 500,
                               "BATCH_CHUNKS": 5}, name="bench", **synth.ARCHS["trans4x-256"])
        vae.model.load_state_dict(synth.synth_state_dict(vae.model.state_dict(), 0))
        return vae

    with tempfile.TemporaryDirectory() as tmp:
        vae = build()
        ops.seed_rng(dev, 1)
        t0 = time.perf_counter()
        vae.train(mols, mols[:500], epochs=2, save=False, log=True, log_dir=tmp)
        torch.cuda.synchronize()
        t_total = time.perf_counter() - t0
        log = pd.read_csv(os.path.join(tmp, "log_bench.txt"))
        rt = log[(log["data_type"] == "train") & (log["epoch"] == 1)]["run_time"].to_numpy()[2:]
    steps = args.mols // 500
    vae = build()
    vae.model.train()
    for data_b in list(dl)[:3]:
        vae.train_step(data_b, 0.01).cpu()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for data_b in dl:
        vae.train_step(data_b, 0.01).cpu().numpy()
    torch.cuda.synchronize()
    t_ref_loop = (time.perf_counter() - t0) / steps
    out["epoch"] = {"train_ms_per_step": float(np.mean(rt) * 1e3), "train_tokens_per_s": 500 * 126 / float(np.mean(rt)),
                    "reference_shaped_loop_ms_per_step": t_ref_loop * 1e3,
                    "reference_shaped_loop_tokens_per_s": 500 * 126 / t_ref_loop,
                    "train_call_wall_s": t_total, "steps_per_epoch": steps,
                    "note": "train = VAEShell.train (device loader, losses read back two steps late); the other loop "
                            "is the same train_step fed by torch's DataLoader with the loss table read every step"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
