#!/bin/bash
# round 4, session S: the GPU suite twice more (flakiness check: the driver runs it with -x) + smoke
set -u
OUT=gpurun_out/r04s; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
for i in 1 2; do
  timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $OUT/pytest_$i.log 2>&1; echo "suite $i rc=$?"; grep -E "passed|failed|^FAILED" $OUT/pytest_$i.log | head -5
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
