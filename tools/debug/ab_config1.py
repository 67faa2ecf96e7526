#!/usr/bin/env python3
"trans1x-128 (BASELINE config 1) step time under the stream switches: python tools/debug/ab_config1.py"
import os, sys, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import time, torch
    import bench
    from transvae_amd import synth
    vae = bench.build_vae("trans1x-128")
    vae.model.train()
    data = torch.cat([synth.synth_tokens(500, seed=100).float(), torch.zeros(500, 1)], 1).cuda()
    for _ in range(5):
        vae.train_step(data, 0.02)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        vae.train_step(data, 0.02)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("  WGRAD_STREAM=%s CONV_WGRAD_STREAM=%s  %.3f ms/step (host queued in %.3f)" % (
        os.environ.get("TVAE_WGRAD_STREAM", "1"), os.environ.get("TVAE_CONV_WGRAD_STREAM", "1"), (time.perf_counter() - t0) / 30 * 1e3, th / 30 * 1e3), flush=True)
    sys.exit(0)
for w, c in (("1", "1"), ("1", "0"), ("0", "0"), ("1", "1"), ("1", "0"), ("0", "0")):
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, TVAE_WGRAD_STREAM=w, TVAE_CONV_WGRAD_STREAM=c))
