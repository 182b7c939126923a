#!/bin/bash
# round 4, GPU session H: NHWC-order LSTM input; suite; bench
set -u
OUT=gpurun_out/r04h; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_all.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest_all.log | head -30
for v in 1 0; do
  MIRL_LSTM_NHWC_INPUT=$v timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > $OUT/bench_nhwc$v.json 2> $OUT/bench_nhwc$v.err; echo "bench nhwc=$v rc=$?"
  python - $OUT/bench_nhwc$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("  ms/step", round(d["ms_per_step"], 3), "median", round(d["step_ms"]["median"], 3))
PY
done
