#!/bin/bash
# round 4, session N: full GPU suite + smoke on the current tree, A/B bench of the input layer's weight-gradient pipe
set -u
OUT=gpurun_out/r04n; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_full.log 2>&1; echo "full suite rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest_full.log | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
for v in 1 0; do
  MIRL_CONV1_WRW_BF16=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_wrw$v.json 2> $OUT/bench_wrw$v.err; echo "bench wrw bf16=$v rc=$?"
  python - $OUT/bench_wrw$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], d.get("step_ms"))
for r in d["roofline_all"]["kernels"]:
    if r["kernel"].startswith("k_conv1_u8_wrw"):
        print("   ", r["kernel"], r["launches_per_step"], round(r["avg_us"], 1), round(r["ms_per_step"], 3), r.get("frac_of_roofline"), r.get("bound"))
PY
done
