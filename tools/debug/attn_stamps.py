#!/usr/bin/env python3
"""Where a workgroup of the attention kernels spends its life: shader-clock stamps (-DTVAE_ATTN_STAMPS build of the library,
made in the container by `python tools/debug/attn_stamps.py build`, loaded through TVAE_LIB_PATH) at the phase boundaries of
attn_fwd_kernel and attn_bwd_dkv_kernel on the SMILES-shaped train batch.  Prints the mean cycles per phase."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "tools", "debug", "libtvae_stamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    src = os.path.join(ROOT, "transvae_amd", "csrc")
    objs = [os.path.join(ROOT, "transvae_amd", "lib", f) for f in os.listdir(os.path.join(ROOT, "transvae_amd", "lib")) if f.endswith(".o") and f != "attention.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-c", os.path.join(src, "attention.hip"), "-o", "/tmp/attention_stamps.o", "--offload-arch=gfx950", "-O3",
                           "-fPIC", "-std=c++17", "-DTVAE_ATTN_STAMPS"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-o", LIB, "/tmp/attention_stamps.o"] + objs + ["--offload-arch=gfx950", "-pthread"])
    print(LIB)
    sys.exit(0)
os.environ["TVAE_LIB_PATH"] = LIB
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from transvae_amd import ops, synth, _lib

B, h, dk, L = 500, 4, 64, 127
d = h * dk
toks = synth.synth_tokens(B, seed=100)
km = (toks[:, :L] != 24)
spec = ops.MaskSpec(key_mask=km.to(torch.uint8).cuda(), causal=False)
drop = ops.Drop(0.1, 3, ops.rng_state(torch.device("cuda", 0)))
q, k, v, do = (torch.randn(B * L, d, device="cuda") for _ in range(4))
o, lse, _ = ops.attn_fwd(q, k, v, B, h, L, L, dk, spec, drop)


def stamps():
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 4 * 16, dtype=np.uint64)
    fn = _lib.lib().tvae_debug_stamps
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf.ctypes.data, buf.size) == 0
    return buf.reshape(1024, 4, 16).astype(np.int64)


def report(name, s, labels):
    print(name)
    t0 = s[:, :, 0]
    prev = t0
    for slot, lab in labels:
        cur = s[:, :, slot]
        ok = cur > 0
        dt = np.where(ok, cur - prev, 0)
        print("  %-58s %8.0f cycles (mean over waves that got there: %d)" % (lab, dt[ok].mean() if ok.any() else 0, ok.sum()))
        prev = np.where(ok, cur, prev)
    life = s[:, :, 14] - t0
    print("  wave lifetime %.0f cycles; workgroup start spread within the first 512: %.0f cycles" % (life[life > 0].mean(), np.ptp(t0[:512, 0])))


for _ in range(5):
    ops.attn_fwd(q, k, v, B, h, L, L, dk, spec, drop)
report("attn_fwd_kernel<64> (SMILES-shaped keys)", stamps(),
       [(1, "loads issued"), (2, "K / V tiles landed and written to LDS"), (3, "barrier"), (4, "S = K Q^T, masks, max, exp, sum"), (13, "dropout, P V"),
        (14, "O stored")])
for _ in range(5):
    ops.attn_bwd(q, k, v, o, do, lse, B, h, L, L, dk, spec, drop)
report("attn_bwd_dkv_kernel<64> (SMILES-shaped keys; the dQ kernel's stamps are not taken)", stamps(),
       [(1, "info word read, loads issued"), (2, "Q / dO tiles landed and written to LDS"), (3, "statistics to LDS, barrier"),
        (4, "query tile 1: S, dP, softmax arithmetic"), (5, "query tile 1: dV, dK MFMAs issued"), (6, "query tile 2: S, dP, softmax arithmetic"),
        (7, "query tile 2: dV, dK MFMAs issued"), (12, "loop exit"), (13, "partial sums through LDS (2 barriers)"), (14, "dK, dV stored")])
