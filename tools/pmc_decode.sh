#!/bin/bash
# HBM traffic of the decode attention launch of bench.py's roofline_decode: two rocprofv3 --pmc passes -> gpurun_out/pmc_out/<tag>_decode_attn_pmc.json (copy into profiles/ afterwards)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r2}
cd /tmp || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$R/gpurun_out/pmc_dec_$C"
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$R/gpurun_out/pmc_dec_$C" -- python3 "$R/tools/pmc_decode_attn.py" > "$R/gpurun_out/pmc_dec_$C.log" 2>&1 || { tail -5 "$R/gpurun_out/pmc_dec_$C.log"; exit 1; }
done
cd "$R" && python3 - "$TAG" <<'P'
import csv, glob, json, os, statistics, sys
sys.path.insert(0, ".")
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_dec_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "decode_attn_kernel" in r["Kernel_Name"]]
    out[c.lower() + "_kib_raw_mean"] = statistics.mean(vals[2:])
    out[c.lower() + "_launches"] = len(vals)
n, n_keys, d, h = 65536, 63, 256, 4
out["shape"] = {"n": n, "n_keys": n_keys, "d_model": d, "heads": h}
out["read_bytes"] = 2.0 * out["fetch_size_kib_raw_mean"] * 1024      # gfx950: FETCH_SIZE counts half of 16-B-per-lane streaming reads
out["write_bytes"] = out["write_size_kib_raw_mean"] * 1024
out["traffic_bytes_per_launch"] = out["read_bytes"] + out["write_bytes"]
out["algorithmic_bytes_per_launch"] = 4.0 * n * d * (2 * n_keys + 2)
out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
out["method"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/pmc_decode_attn.py; FETCH_SIZE doubled (gfx950)"
os.makedirs("gpurun_out/pmc_out", exist_ok=True)
json.dump(out, open("gpurun_out/pmc_out/%s_decode_attn_pmc.json" % tag, "w"), indent=1)
print(json.dumps(out))
P
