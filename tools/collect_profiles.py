#!/usr/bin/env python3
"""Copy what `bash tools/round_profiles.sh <tag>` left under gpurun_out/ into profiles/ under the names profiles/README.md
lists (gpurun_out/ is scratch; profiles/ is what is committed).  python tools/collect_profiles.py <tag>"""
import glob, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")


def cp(src, dst):
    if src and os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, dst))
        print("%-60s <- %s" % (dst, os.path.relpath(src, R)))
    else:
        print("MISSING: %s (for %s)" % (src, dst))


def stats(d):
    f = glob.glob(os.path.join(G, d, "**", "*kernel_stats.csv"), recursive=True)
    return max(f, key=os.path.getmtime) if f else None          # merged runs pile up under gpurun_out/: the newest


for m in ("fp32", "f32x3", "bf16"):
    cp(stats("prof_%s_train_%s" % (tag, m)), "%s_train_%s_kernel_stats.csv" % (tag, m))
    cp(os.path.join(G, "prof_%s_train_%s_bench.json" % (tag, m)), "%s_train_%s_bench_line.json" % (tag, m))
    cp(stats("prof_%s_roofline_%s" % (tag, m)), "%s_roofline_only_%s_kernel_stats.csv" % (tag, m))
    cp(os.path.join(G, "%s_roofline_only_%s_line.json" % (tag, m)), "%s_roofline_only_%s_line.json" % (tag, m))
cp(stats("prof_%s_decode" % tag), "%s_decode_65536_kernel_stats.csv" % tag)
cp(os.path.join(G, "%s_decode_result.txt" % tag), "%s_decode_65536_result.txt" % tag)
for f in sorted(glob.glob(os.path.join(G, "pmc_out", "%s_*" % tag))):
    cp(f, os.path.basename(f))
for name, dst in (("%s_gemm_shapes_step_fp32.txt", None), ("%s_attn_sq_counters.txt", None), ("%s_wide_tile_gemm.txt", None),
                  ("%s_bench_default.json", "%s_bench_default_line.json")):
    cp(os.path.join(G, name % tag), (dst or name) % tag)
