#!/usr/bin/env python3
"""gpurun_out/pmc_{fetch,write}/**/counter_collection.csv -> profiles/r1_gemm_pmc.json (per-launch HBM bytes
of the roofline GEMM).  Units and corrections per MI355X_MICROARCH.md (HBM section): the counters are in
KiB; on gfx950 FETCH_SIZE reports half of the bytes of 16-B-per-lane streaming reads, so it is doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import csv, glob, json, os, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for kind in ("fetch", "write"):
    f = glob.glob(os.path.join(root, "gpurun_out", "pmc_%s" % kind, "**", "*counter_collection.csv"), recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_kernel" in r["Kernel_Name"]]
    out[kind + "_size_kib_raw_mean"] = statistics.mean(vals[2:])          # first two launches: cold caches
    out[kind + "_launches"] = len(vals)
M, N, K = 63500, 1024, 256
out["shape"] = [M, N, K]
out["read_bytes"] = 2.0 * out["fetch_size_kib_raw_mean"] * 1024
out["write_bytes"] = out["write_size_kib_raw_mean"] * 1024
out["traffic_bytes_per_launch"] = out["read_bytes"] + out["write_bytes"]
out["algorithmic_bytes_per_launch"] = 4.0 * (M * K + N * K + N + M * N)
out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
out["method"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/pmc_gemm.py; FETCH_SIZE doubled (gfx950)"
json.dump(out, open(os.path.join(root, "profiles", "r1_gemm_pmc.json"), "w"), indent=1)
print(json.dumps(out))
