#!/bin/bash
# Runs the GPU test files one process each; stops at the first process that dies abnormally
# (anything other than "all passed" (0) or "some assertions failed" (1)).
mkdir -p gpurun_out
for f in "$@"; do
  name=$(basename "$f" .py)
  timeout -k 10 420 python -u -m pytest "$f" -v --timeout 120 --timeout-method thread -s > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "== $f rc=$rc"
  grep -E "PASSED|FAILED|ERROR|rel-L2|passed|failed" "gpurun_out/$name.log" | tail -60
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal exit, stopping"; tail -30 "gpurun_out/$name.log"; exit $rc; fi
done
exit 0
