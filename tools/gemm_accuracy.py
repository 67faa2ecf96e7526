#!/usr/bin/env python3
"""Error of tvae_gemm against an fp64 product for the three arithmetic modes (relative L2 and max)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transvae_amd import ops

def err(y, ref):
    d = (y.double() - ref)
    return (d.norm() / ref.norm()).item(), (d.abs().max() / ref.abs().max()).item()

torch.manual_seed(0)
for (M, N, K) in [(4096, 256, 256), (4096, 1024, 256), (2048, 256, 1024), (1000, 260, 2304)]:
    x = torch.randn(M, K, device="cuda"); w = torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.05); b = torch.randn(N, device="cuda")
    dy = torch.randn(M, N, device="cuda")
    torch.set_grad_enabled(False)
    ref_f = x.double() @ w.double().t() + b.double()
    ref_d = dy.double() @ w.double()
    ref_w = dy.double().t() @ x.double()
    for mode in ("fp32", "f32x3", "bf16"):
        ops.set_compute_dtype(mode)
        yf = ops.linear_fwd(x, w, b)
        yd = ops.linear_dgrad(dy, w)
        yw, _ = ops.linear_wgrad(dy, x)
        print("M%5d N%5d K%5d %-6s fwd %.2e/%.2e  dgrad %.2e/%.2e  wgrad %.2e/%.2e" %
              ((M, N, K, mode) + err(yf, ref_f) + err(yd, ref_d) + err(yw, ref_w)), flush=True)
    t = x.double() @ w.double().t() + b.double()
    print("           torch fp32 matmul   fwd %.2e/%.2e" % err(x @ w.t() + b, t))
