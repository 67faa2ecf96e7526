#!/bin/bash
# L2 memory-side read traffic of the weight-gradient GEMM shapes (FETCH_SIZE pass only): bash tools/pmc_wgrad.sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp || exit 1
for S in "1024 256" "256 256" "256 1024"; do
  set -- $S
  D="$R/gpurun_out/pmc_wgrad_$1_$2"
  rm -rf "$D"
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D" -- python3 "$R/tools/pmc_shape.py" fp32 $1 $2 tn > "$D.log" 2>&1 || exit 1
  python3 - "$D" $1 $2 <<'PY'
import csv, glob, sys
d, n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
M = 63500
alg = 4.0 * M * (n + k)
print("wgrad dW[%d,%d]: FETCH (x2, gfx950) %.1f MB per launch, operands %.1f MB, ratio %.2f" % (n, k, 2 * 1024 * sum(v[2:]) / len(v[2:]) / 1e6, alg / 1e6, 2 * 1024 * sum(v[2:]) / len(v[2:]) / alg))
PY
done
