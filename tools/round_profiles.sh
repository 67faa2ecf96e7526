#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call:  bash tools/round_profiles.sh <tag>
#   kernel stats of the train step in the three GEMM arithmetics, of the 65536-latent decode, the
#   HBM traffic of the roofline GEMM (PMC passes), every GEMM of a step by shape, the SQ counters of the attention kernels,
#   the wide-tile GEMM beside the 128 x 128 kernel and the default bench line.  Each step under its own timeout.
set -o pipefail
TAG=${1:-rX}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
for DT in fp32 f32x3 bf16; do
  bash tools/prof_train.sh $DT ${TAG}_train_$DT > gpurun_out/${TAG}_train_$DT.digest 2>&1 || { echo "prof_train $DT failed"; tail -5 gpurun_out/${TAG}_train_$DT.digest; exit 1; }
  echo "== train $DT"; head -8 gpurun_out/${TAG}_train_$DT.digest
done
for DT in fp32 f32x3 bf16; do
  ( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_${TAG}_roofline_$DT" -- python3 "$R/bench.py" --roofline-only --dtype $DT > "$R/gpurun_out/${TAG}_roofline_only_${DT}_line.json" 2> "$R/gpurun_out/${TAG}_roofline_only_$DT.err" ) || { echo "roofline-only $DT failed"; tail -5 gpurun_out/${TAG}_roofline_only_$DT.err; exit 1; }
  echo "== roofline-only $DT"; cat gpurun_out/${TAG}_roofline_only_${DT}_line.json
done
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_${TAG}_decode" -- python3 "$R/tools/decode_bench.py" 65536 > "$R/gpurun_out/${TAG}_decode_result.txt" 2>&1 ) || { echo "decode profile failed"; tail -5 gpurun_out/${TAG}_decode_result.txt; exit 1; }
echo "== decode"; grep "n=" gpurun_out/${TAG}_decode_result.txt
PMC_TAG=${TAG} bash tools/pmc_traffic.sh > gpurun_out/${TAG}_pmc.log 2>&1 || { echo "pmc failed"; tail -5 gpurun_out/${TAG}_pmc.log; exit 1; }
echo "== pmc ok"
bash tools/pmc_decode.sh ${TAG} > gpurun_out/${TAG}_pmc_decode.log 2>&1 || { echo "pmc decode failed"; tail -5 gpurun_out/${TAG}_pmc_decode.log; exit 1; }
echo "== pmc decode ok"
timeout -k 10 200 python tools/gemm_shapes_step.py > gpurun_out/${TAG}_gemm_shapes_step_fp32.txt 2>&1 || { echo "gemm shapes failed"; tail -5 gpurun_out/${TAG}_gemm_shapes_step_fp32.txt; exit 1; }
echo "== gemm shapes"; tail -1 gpurun_out/${TAG}_gemm_shapes_step_fp32.txt
bash tools/pmc_attn.sh ${TAG} > gpurun_out/${TAG}_attn_sq_counters.txt 2>&1 || { echo "attention counters failed"; tail -5 gpurun_out/${TAG}_attn_sq_counters.txt; exit 1; }
echo "== attention counters ok"
timeout -k 10 200 python tools/debug/wide_bench.py > gpurun_out/${TAG}_wide_tile_gemm.txt 2>&1 && TVAE_GEMM_WIDE=0 timeout -k 10 200 python tools/debug/wide_bench.py >> gpurun_out/${TAG}_wide_tile_gemm.txt 2>&1
timeout -k 10 500 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_bench_default.err; exit 1; }
echo "== bench"; cut -c1-330 gpurun_out/${TAG}_bench_default.json
