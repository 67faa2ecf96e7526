#!/bin/bash
# Run the GPU test-suite on the box, one pytest process per file; stop at the first abnormal exit
# (anything but pytest's 0 = passed / 1 = test failures), print a digest to stdout.
mkdir -p gpurun_out
for f in "$@"; do
  name=$(basename "$f" .py)
  timeout -k 10 600 python -u -m pytest "$f" -q --timeout 300 --timeout-method thread -m gpu -s > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "== $f rc=$rc"
  grep -E "^(FAILED|ERROR)|passed|failed|worst per-tensor" "gpurun_out/$name.log" | cut -c1-220
  grep -E "^E  " "gpurun_out/$name.log" | cut -c1-260 | head -60
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal exit, stopping"; exit $rc; fi
done
exit 0
