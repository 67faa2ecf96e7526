#!/usr/bin/env python3
"""HBM-bound kernels of the trans4x-256 train step: achieved bytes/s against the 8 TB/s HBM3E peak.
Durations: the rocprofv3 kernel summary of the fp32 train profile (profiles/<tag>_train_fp32_kernel_stats.csv);
bytes: ALGORITHMIC bytes per launch (each tensor the kernel must read or write, once), batch 500.

    python tools/hbm_table.py r2 > profiles/r2_hbm_kernels.txt
"""
import csv, os, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r2"
SRC = {"r1": "r1_final_train_fp32_kernel_stats.csv"}.get(TAG, "%s_train_fp32_kernel_stats.csv" % TAG)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, Ls, Lt, d, V, P = 500, 127, 126, 256, 25, 7876205
rows_avg = B * (Ls * 11 + Lt * 9) / 20.0          # 20 LayerNorms per forward: 11 over source rows, 9 over target rows
KERNELS = [
    # name fragment, what, algorithmic bytes per launch
    ("ln_fwd_kernel<4>", "LayerNorm forward (x -> y; unbiased std, eps on std)", rows_avg * d * 8),
    ("ln_bwd_kernel<4>", "LayerNorm backward (x, dy, residual-stream gradient -> dx; da, db)", rows_avg * d * 16),
    ("adam_kernel", "multi-tensor Adam + zero_grad (p, g, m, v -> p, m, v, g)", P * 32),
    ("embed_fwd_kernel", "embedding * sqrt(d) + positional encoding + dropout", B * (Ls + Lt) / 2 * d * 4),
    ("embed_bwd_kernel", "embedding gradient (atomics into 26 table rows)", B * (Ls + Lt) / 2 * d * 4),
    ("loss_reduce_kernel", "weighted CE over 25 classes + length CE + KLD (+MSE): sums", B * Lt * V * 4 + B * 128 * 12),
    ("loss_grad_kernel", "the same terms: gradients (logits -> dlogits, ...)", 2 * (B * Lt * V * 4 + B * 128 * 12)),
    ("upsample_fwd_kernel", "zero-upsampling in front of a transposed convolution", None),
    ("pool_relu_fwd_kernel", "MaxPool1d(2) + ReLU", None),
]
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", SRC)))}
print("HBM-bound kernels, trans4x-256 train step, batch 500 (durations: profiles/%s)" % SRC)
print("%-22s %7s %10s %12s %10s %8s  %s" % ("kernel", "calls", "avg us", "alg. MB", "TB/s", "of 8 TB/s", "what"))
for frag, what, nbytes in KERNELS:
    hit = [r for n, r in stats.items() if frag in n]
    if not hit:
        continue
    r = hit[0]
    us = float(r["AverageNs"]) / 1e3
    if nbytes is None:
        print("%-22s %7s %10.1f %12s %10s %8s  %s" % (frag, r["Calls"], us, "-", "-", "-", what + " (several shapes per step: not priced)"))
        continue
    tbs = nbytes / (us * 1e-6) / 1e12
    print("%-22s %7s %10.1f %12.1f %10.2f %7.0f%%  %s" % (frag, r["Calls"], us, nbytes / 1e6, tbs, 100 * tbs / 8.0, what))
print()
print("Softmax has no HBM traffic of its own: the score tile of an attention call lives in MFMA accumulators (csrc/attention.hip).")
print("The two loss kernels and the embedding kernels move 6-65 MB per launch: they are launch-latency-sized (< 0.3 % of the step).")
print("Loss kernels: 250 workgroups (50 per chunk of 100 molecules), one per CU, each a chain of dependent phases (stage the 256")
print("logit rows in LDS -> 25-class log-softmax per thread -> block sums -> one wave per molecule for the length term -> KL -> MSE")
print("-> 5 atomics): their 14-18 us are load and barrier latencies in sequence, not bytes -- staging the rows through LDS (round 2)")
print("took the 64-cache-lines-per-load walks out (19.8 -> 17.9 us, 17.3 -> 13.7 us) and that is what there was to take.")
