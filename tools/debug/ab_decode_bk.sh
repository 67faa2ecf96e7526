#!/bin/bash
# four-stream decode of 262144 latents with the GEMMs' k-depth left to the dispatcher against TVAE_GEMM_BK=32 (two workgroups of
# 72 KB per CU instead of four of 40 KB: room for the HBM-bound attention waves of another piece beside them); rows are
# bit-identical either way (both depths sum in the same order)
for v in 0 32 0 32; do
  export TVAE_GEMM_BK=$v
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-modes --no-roofline --no-config1 --decode-n 262144 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('TVAE_GEMM_BK=$v  %.0f molecules/s (%.2f s)' % (d['decode_mols_per_s'], d['decode_seconds']))" || exit 1
done
