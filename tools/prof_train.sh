#!/bin/bash
# rocprofv3 kernel stats of the train step in one GEMM arithmetic:  bash tools/prof_train.sh <fp32|f32x3|bf16> <tag>
# -> gpurun_out/prof_<tag>/ (+ the bench line of the profiled run in gpurun_out/prof_<tag>_bench.json)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
DT=$1; TAG=$2
cd /tmp || exit 1
rm -rf "$R/gpurun_out/prof_$TAG"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$TAG" -- python3 "$R/bench.py" --dtype $DT --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-extra-modes --no-roofline --no-config1 > "$R/gpurun_out/prof_${TAG}_bench.json" 2> "$R/gpurun_out/prof_${TAG}.err" || { tail -5 "$R/gpurun_out/prof_${TAG}.err"; exit 1; }
cd "$R" && python3 - "$TAG" <<'P'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob("gpurun_out/prof_%s/**/*kernel_stats.csv" % tag, recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms (4 train steps)" % (tot / 1e6))
for r in rows[:24]:
    print("%-100s %5s %8.2f ms %9.1f us %5.1f%%" % (r["Name"].replace("(anonymous namespace)::", "")[:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
P
