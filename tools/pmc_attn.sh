#!/bin/bash
# SQ-side counters of the three fused attention kernels on the train-step shapes (tools/attn_bench.py: SMILES-shaped and
# unpadded key masks): wave cycles (parked / issue-stalled / active), MFMA-busy, LDS waits and bank conflicts, the clock the
# chip held.  Two rocprofv3 --pmc passes (8 SQ counters per pass), each with --kernel-trace only.
#   bash tools/pmc_attn.sh <tag>    -> gpurun_out/pmc_attn_<tag>_{a,b}/ + a digest on stdout
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r3}
cd /tmp || exit 1
rm -rf "$R/gpurun_out/pmc_attn_${TAG}_a" "$R/gpurun_out/pmc_attn_${TAG}_b"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d "$R/gpurun_out/pmc_attn_${TAG}_a" -- python3 "$R/tools/attn_bench.py" > "$R/gpurun_out/pmc_attn_${TAG}_a.log" 2>&1 || { tail -5 "$R/gpurun_out/pmc_attn_${TAG}_a.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES \
  --output-format csv -d "$R/gpurun_out/pmc_attn_${TAG}_b" -- python3 "$R/tools/attn_bench.py" > "$R/gpurun_out/pmc_attn_${TAG}_b.log" 2>&1 || { tail -5 "$R/gpurun_out/pmc_attn_${TAG}_b.log"; echo "(second pass failed: first pass digest only)"; }
cd "$R" && python3 - "$TAG" <<'P'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> values
dur = collections.defaultdict(list)
for ps in ("a", "b"):
    fs = glob.glob("gpurun_out/pmc_attn_%s_%s/**/*counter_collection.csv" % (tag, ps), recursive=True)
    if not fs:
        continue
    by = collections.defaultdict(dict)
    name = {}
    for r in csv.DictReader(open(fs[0])):
        if "attn_" not in r["Kernel_Name"]:
            continue
        by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    kt = glob.glob("gpurun_out/pmc_attn_%s_%s/**/*kernel_trace.csv" % (tag, ps), recursive=True)
    d = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            d[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
    # attn_bench.py runs 6 cases (3 shapes x smiles / unpadded) of 1 + 3 + 20 forward and 3 + 20 backward launches each
    ids = sorted(by, key=int)
    per_kernel = collections.defaultdict(list)
    for i in ids:
        per_kernel[name[i]].append(i)
    for k, lst in per_kernel.items():
        n = len(lst) // 6
        for case in range(6):
            chunk = lst[case * n:(case + 1) * n][-10:]         # the last launches of the case: warm
            key = "%s | case %d" % (k, case)
            for i in chunk:
                for c, v in by[i].items():
                    agg[key][c].append(v)
                if ps == "a" and i in d:
                    dur[key].append(d[i])
cases = ["enc self smiles", "enc self unpadded", "dec self causal smiles", "dec self causal unpadded", "dec src smiles", "dec src unpadded"]
mean = lambda v: sum(v) / max(len(v), 1)
print("# attention kernels, trans4x-256 train shapes (B 500, h 4, dk 64: 2000 workgroups of 4 waves), rocprofv3 --pmc, mean of the last 10 launches per case")
print("# wave-cycle shares: parked = SQ_WAIT_ANY, issue-stall = SQ_WAIT_INST_ANY, active = SQ_ACTIVE_INST_ANY (of SQ_WAVE_CYCLES, quad-cycles)")
for key in sorted(agg):
    c = {k: mean(v) for k, v in agg[key].items()}
    k, case = key.split(" | case ")
    t = mean(dur[key]) if dur[key] else 0.0
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1.0
    line = "%-34s %-26s" % (k[:34], cases[int(case)])
    if t:
        line += " %6.1f us  clock %.2f GHz" % (t * 1e6, c.get("GRBM_GUI_ACTIVE", 0) / 8 / t / 1e9 if c.get("GRBM_GUI_ACTIVE") else 0)
    # MFMA-busy share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles of the launch), cycles = GRBM_GUI_ACTIVE / 8 (sum over 8 XCDs)
    line += " | wave cycles: parked %.2f issue-stall %.2f (lds %.2f) active %.2f | MFMA-busy %.2f | wave lifetime %.1fk cycles" % (
        c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_WAIT_INST_LDS", 0) / wc,
        c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * (c.get("GRBM_GUI_ACTIVE", 0) / 8.0 or 1.0)),
        4.0 * wc / 8000.0 / 1e3)
    if "SQ_LDS_BANK_CONFLICT" in c:
        line += " | LDS bank-conflict cycles / LDS active %.3f, LDS insts %.3g, VALU insts %.3g" % (
            c["SQ_LDS_BANK_CONFLICT"] / (c.get("SQ_LDS_IDX_ACTIVE", 0) or 1.0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VALU", 0))
    print(line)
P
