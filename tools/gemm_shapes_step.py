#!/usr/bin/env python3
"""Every GEMM launch of one fp32 train step of trans4x-256 (batch 500) with its shape: M, N, K and flags logged by a wrapper
around ops.gemm, each launch timed with events (synchronised: durations are those of isolated warm launches), aggregated
by shape with the achieved TFLOP/s.  python tools/gemm_shapes_step.py [dtype [model]]"""
import sys, os, collections
os.environ["TVAE_WGRAD_STREAM"] = "0"      # isolated launches: the weight gradients stay on the compute stream, where the events are
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from transvae_amd import ops, synth

if len(sys.argv) > 1:
    ops.set_compute_dtype(sys.argv[1])
vae = bench.build_vae(*sys.argv[2:3])
vae.model.train()
toks = synth.synth_tokens(500, seed=100).float()
data = torch.cat([toks, torch.zeros(500, 1)], 1).cuda()
for _ in range(3):
    vae.train_step(data, 0.03)
torch.cuda.synchronize()
log = []
orig = ops.gemm


def timed(A, lda, Bm, ldb, Cm, ldc, M, N, K, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(A, lda, Bm, ldb, Cm, ldc, M, N, K, **kw)
    e1.record()
    e1.synchronize()
    f = kw.get("flags", 0)
    kind = ("TN" if f & ops.GEMM_A_KMAJOR else "NN" if f & ops.GEMM_B_KMAJOR else "NT") + (" drop" if kw.get("drop") is not None else "") + \
           (" res" if kw.get("residual") is not None else "")
    log.append((kind, M, N, K, e0.elapsed_time(e1) * 1e3))


ops.gemm = timed
for _ in range(2):
    vae.train_step(data, 0.03)
ops.gemm = orig
agg = collections.defaultdict(lambda: [0, 0.0])
for kind, M, N, K, us in log:
    agg[(kind, M, N, K)][0] += 1
    agg[(kind, M, N, K)][1] += us
tot = sum(v[1] for v in agg.values()) / 2
print("%-12s %8s %6s %6s %5s %9s %9s %8s" % ("form", "M", "N", "K", "n", "avg us", "ms/step", "TFLOP/s"))
for (kind, M, N, K), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-12s %8d %6d %6d %5.1f %9.1f %9.2f %8.1f" % (kind, M, N, K, n / 2, us / n, us / 2e3, 2.0 * M * N * K * n / us / 1e6))
print("all GEMMs (isolated launches): %.2f ms/step" % (tot / 1e3))
