#!/usr/bin/env python3
"""GEMM time of a profiled train run by kernel instantiation and grid (= by shape): python tools/gemm_by_grid.py <prof dir> [steps]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if "gemm" not in n or "_kernel<" not in n: continue
    key = (n.split("(")[0][5:], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Z"])
    agg[key][0] += 1; agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-58s tiles %5d slabs %4s  n=%5.1f  avg %7.1f us  %6.2f ms/step" % (k[0], k[1], k[2], v[0] / steps, v[1] / v[0], v[1] / steps / 1e3))
print("all GEMMs: %.2f ms/step" % (sum(v[1] for v in agg.values()) / steps / 1e3))
