#!/usr/bin/env python3
"secondary recurrent-model bench lines with the weight-gradient stream off / on (TVAE_WGRAD_STREAM)"
import os, sys, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    for kind, d in (("rnnattn", 256), ("rnn", 128)):
        r = bench.rnn_secondary(kind, d, steps=10, decode_n=2048)
        print("  WGRAD_STREAM=%s %s-%d  %.3f ms/step" % (os.environ.get("TVAE_WGRAD_STREAM"), kind, d, r["ms_per_step"]), flush=True)
    sys.exit(0)
for v in ("0", "1", "0", "1"):
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, TVAE_WGRAD_STREAM=v))
