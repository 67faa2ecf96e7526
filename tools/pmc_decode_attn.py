#!/usr/bin/env python3
"""Launch the dominant decode kernel (decode_attn_kernel<64>: one query per (molecule, head) over the self-attention K/V
cache at the mean key count of a 125-step decode) a few times; under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
(separate passes) this counts its HBM traffic.  Same launch as bench.py's roofline_decode."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import decode_attn_launch  # noqa: E402

fn, _ = decode_attn_launch(int(sys.argv[1]) if len(sys.argv) > 1 else 65536)
for _ in range(8):
    fn()
torch.cuda.synchronize()
