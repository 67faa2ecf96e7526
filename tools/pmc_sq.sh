#!/bin/bash
# SQ-side counters of one GEMM shape (wave cycles, parked / issue-stalled / active, MFMA-busy) plus the
# clock the chip held (GRBM_GUI_ACTIVE / 8 / duration).  One rocprofv3 --pmc pass per call:
#   bash tools/pmc_sq.sh <dtype> <N> <K> <nt|nn|tn> <tag>    -> gpurun_out/pmc_sq_<tag>/ + a digest on stdout
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
DT=$1; N=$2; K=$3; KIND=$4; TAG=$5
cd /tmp || exit 1
rm -rf "$R/gpurun_out/pmc_sq_$TAG"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d "$R/gpurun_out/pmc_sq_$TAG" -- python3 "$R/tools/pmc_shape.py" $DT $N $K $KIND > "$R/gpurun_out/pmc_sq_$TAG.log" 2>&1 || { tail -5 "$R/gpurun_out/pmc_sq_$TAG.log"; exit 1; }
cd "$R" && python3 - "$TAG" <<'P'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob("gpurun_out/pmc_sq_%s/**/*counter_collection.csv" % tag, recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"] and "split3" not in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
kt = glob.glob("gpurun_out/pmc_sq_%s/**/*kernel_trace.csv" % tag, recursive=True)
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
ids = sorted(by, key=int)[2:]
if len(ids) > 12:
    ids = ids[:3] + ids[-6:]          # a long run: the first launches (cold clock) and the last ones
for d in ids:
    c = by[d]
    t = dur.get(d)
    line = "disp %s" % d
    if t:
        line += " %.1f us clock %.2f GHz" % (t * 1e6, c.get("GRBM_GUI_ACTIVE", 0) / 8 / t / 1e9)
    wc = c.get("SQ_WAVE_CYCLES", 1)
    line += " | of wave cycles: parked %.2f issue-stall %.2f (lds %.2f) active %.2f" % (
        c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_WAIT_INST_LDS", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc)
    line += " | MFMA busy cycles %.3g, SQ busy cycles %.3g, GUI %.3g" % (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), c.get("SQ_BUSY_CYCLES", 0), c.get("GRBM_GUI_ACTIVE", 0))
    print(line)
P
