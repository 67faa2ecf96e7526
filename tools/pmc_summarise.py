#!/usr/bin/env python3
"""gpurun_out/pmc_{fetch,write}/**/counter_collection.csv -> gpurun_out/pmc_out/<tag>_gemm_pmc_<dtype>.json (only gpurun_out/ comes back from the GPU box: copy into profiles/ afterwards) (per-launch HBM bytes
of the roofline GEMM).  Units and corrections per MI355X_MICROARCH.md (HBM section): the counters are in
KiB; on gfx950 FETCH_SIZE reports half of the bytes of 16-B-per-lane streaming reads, so it is doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import csv, glob, json, os, statistics, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M, N, K = 63500, 1024, 256
TAG = sys.argv[1] if len(sys.argv) > 1 else "r2"
OUT = os.path.join(root, "gpurun_out", "pmc_out")
os.makedirs(OUT, exist_ok=True)
for dt in (sys.argv[2:] or ["fp32", "f32x3", "bf16"]):
    out = {}
    for kind in ("fetch", "write"):
        f = glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_%s" % (kind, dt), "**", "*counter_collection.csv"), recursive=True)[0]
        rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"] and "split3" not in r["Kernel_Name"]]
        vals = [float(r["Counter_Value"]) for r in rows]
        out[kind + "_size_kib_raw_mean"] = statistics.mean(vals[2:])          # first two launches: cold caches
        out[kind + "_launches"] = len(vals)
        out["kernel"] = rows[0]["Kernel_Name"].replace("(anonymous namespace)::", "").split("((")[0][:80]
        with open(os.path.join(OUT, "%s_gemm_pmc_%s_%s.csv" % (TAG, dt, kind)), "w") as g:
            g.write("Dispatch_Id,Grid_Size,Kernel_Name,Counter_Name,Counter_Value\n")
            for r in rows:
                g.write("%s,%s,\"%s\",%s,%s\n" % (r["Dispatch_Id"], r["Grid_Size"], r["Kernel_Name"], r["Counter_Name"], r["Counter_Value"]))
    out["shape"] = [M, N, K]
    out["dtype"] = dt
    out["read_bytes"] = 2.0 * out["fetch_size_kib_raw_mean"] * 1024
    out["write_bytes"] = out["write_size_kib_raw_mean"] * 1024
    out["traffic_bytes_per_launch"] = out["read_bytes"] + out["write_bytes"]
    out["algorithmic_bytes_per_launch"] = 4.0 * (M * K + N * K + N + M * N)
    out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
    out["method"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/pmc_gemm.py %s; "
                     "FETCH_SIZE doubled (gfx950)" % dt)
    json.dump(out, open(os.path.join(OUT, "%s_gemm_pmc_%s.json" % (TAG, dt)), "w"), indent=1)
    print(json.dumps(out))
