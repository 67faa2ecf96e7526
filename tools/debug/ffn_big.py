"""dh = (dy W2) * [hmid > 0] at M = 12700 against fp64, element by element: where do the errors sit?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from transvae_amd import ops
torch.manual_seed(0)
for M in (508, 1524, 12700, 63500):
    d, dff = 256, 1024
    n = torch.randn(M, d); W1 = torch.randn(dff, d) * 0.06; b1 = torch.randn(dff) * 0.05
    W2 = torch.randn(d, dff) * 0.03; dy = torch.randn(M, d)
    pre64 = n.double() @ W1.double().t() + b1.double()
    h64 = torch.relu(pre64)
    dh64 = (dy.double() @ W2.double()) * (pre64 > 0)
    g = lambda t: t.cuda()
    ng, W1g, b1g, W2g, dyg = g(n), g(W1), g(b1), g(W2), g(dy)
    hmid = ops.linear_fwd(ng, W1g, b1g, relu=True, out=ops.rows_buffer(M, dff, ng.device))
    dh = ops.linear_dgrad(dyg, W2g, relumask=hmid, out=ops.rows_buffer(M, dff, ng.device))
    torch.cuda.synchronize()
    hm, dhc = hmid.cpu().double(), dh.cpu().double()
    gate_mis = ((hm > 0) != (pre64 > 0))
    err = (dhc - dh64).abs()
    bad = err > 1e-4 * dh64.abs().max()
    print("M=%d: h rel-L2 %.2e; gate mismatches %d (min |pre| there %.2e); dh rel-L2 %.2e; elements off by >1e-4 of max: %d; of those at gate mismatches: %d"
          % (M, float((hm - h64).norm() / h64.norm()), int(gate_mis.sum()),
             float(pre64.abs()[gate_mis].min()) if gate_mis.any() else -1, float((dhc - dh64).norm() / dh64.norm()),
             int(bad.sum()), int((bad & gate_mis).sum())), flush=True)
    if bad.any():
        idx = torch.nonzero(bad)[:8]
        print("   first bad (row, col, got, want, pre):", [(int(r), int(c), float(dhc[r, c]), float(dh64[r, c]), float(pre64[r, c])) for r, c in idx])
