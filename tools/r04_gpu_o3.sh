#!/bin/bash
set -u
OUT=gpurun_out/r04o; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 900 python -m pytest tests/test_network_ab_gpu.py tests/test_e2e_gpu.py tests/test_conv_wrw_gpu.py -m gpu -q --timeout 600 > $OUT/pytest4.log 2>&1; echo "rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest4.log | head -30
for v in 1 0; do
  MIRL_CONV_WRW=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench2_wrw$v.json 2> $OUT/bench2_wrw$v.err; echo "bench conv wrw=$v rc=$?"
  python - $OUT/bench2_wrw$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], d.get("step_ms"))
for r in d["roofline_all"]["kernels"]:
    if r["kernel"].startswith("k_conv_wrw"):
        print("   ", r["kernel"], r["launches_per_step"], round(r["avg_us"], 1), round(r["ms_per_step"], 3), r.get("frac_of_roofline"), r.get("bound"))
PY
done
