#!/bin/bash
# HBM traffic of the roofline GEMM from the L2 memory-side counters: two separate rocprofv3 --pmc
# passes (FETCH_SIZE and WRITE_SIZE cannot share a pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Run on the GPU box from the repo root:  bash tools/pmc_traffic.sh  -> gpurun_out/pmc_{fetch,write}/
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp || exit 1
for DT in ${PMC_DTYPES:-fp32 f32x3 bf16}; do
  rm -rf "$R/gpurun_out/pmc_fetch_$DT" "$R/gpurun_out/pmc_write_$DT"
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc_fetch_$DT" -- python3 "$R/tools/pmc_gemm.py" $DT > "$R/gpurun_out/pmc_fetch_$DT.log" 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc_write_$DT" -- python3 "$R/tools/pmc_gemm.py" $DT > "$R/gpurun_out/pmc_write_$DT.log" 2>&1 || exit 1
done
cd "$R" && python3 tools/pmc_summarise.py ${PMC_TAG:-r2} ${PMC_DTYPES:-fp32 f32x3 bf16}
