#!/bin/bash
# decode_latents on one stream (TVAE_DECODE_STREAMS=1) against two half slabs on two streams (=2), 262144 latents
for v in 1 4 1 4; do
  export TVAE_DECODE_STREAMS=$v
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-modes --no-roofline --no-config1 --decode-n 262144 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('TVAE_DECODE_STREAMS=$v  %.0f molecules/s (%.2f s)' % (d['decode_mols_per_s'], d['decode_seconds']))" || exit 1
done
