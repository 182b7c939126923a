#!/bin/bash
# round 4, session I: conv layers 2-3 backward over the window matrix (conv_col.hip): tests, kernel probe, same-box A/B bench
set -u
OUT=gpurun_out/r04i; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 600 python -m pytest tests/test_conv_col_gpu.py tests/test_abi.py tests/test_conv3_gpu.py -m gpu -q --timeout 600 -x > $OUT/pytest_col.log 2>&1; echo "col rc=$?"; grep -E "passed|failed|^E  " $OUT/pytest_col.log | head -20
timeout 300 python tools/conv_col_probe.py 5120 640 > $OUT/conv_col_probe.jsonl 2> $OUT/conv_col_probe.err; echo "probe rc=$?"; cat $OUT/conv_col_probe.jsonl; tail -3 $OUT/conv_col_probe.err
for v in 1 0; do
  MIRL_CONV_COL=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_col$v.json 2> $OUT/bench_col$v.err; echo "bench col=$v rc=$?"
  python - $OUT/bench_col$v.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms/step", d["ms_per_step"], d.get("step_ms"))
except Exception as e:
    print("no line", e)
PY
done
