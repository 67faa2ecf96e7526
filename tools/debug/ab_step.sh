#!/bin/bash
# A/B of one environment switch on the train step, same box: bash tools/debug/ab_step.sh VAR [modes]
VAR=${1:-TVAE_WGRAD_STREAM}
MODES=${2:-"fp32 f32x3 bf16"}
F="--steps 20 --warmup 3 --no-cpu-baseline --no-decode --no-extra-modes --no-roofline --no-config1"
for m in $MODES; do
  for v in 0 1 0 1; do
    export $VAR=$v
    timeout -k 10 120 python bench.py $F --dtype $m 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$m $VAR=$v  %.3f ms/step' % d['ms_per_step'])" || exit 1
  done
done
