#!/bin/bash
# round 4, session L: layer-2 data gradient on the bf16 pipe (conv_mid.hip b3) + acting call reuse: tests, A/B bench
set -u
OUT=gpurun_out/r04l; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 900 python -m pytest tests/test_conv_mid_gpu.py tests/test_abi.py tests/test_gemm3_gpu.py tests/test_fast_acting_gpu.py tests/test_e2e_gpu.py -m gpu -q --timeout 600 > $OUT/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest.log | head -30
for v in bf16 f32; do
  MIRL_CONV2_BWD_PIPE=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c2$v.json 2> $OUT/bench_c2$v.err; echo "bench conv2 bwd pipe=$v rc=$?"
  python - $OUT/bench_c2$v.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms/step", d["ms_per_step"], d.get("step_ms"))
    for r in d["roofline_all"]["kernels"]:
        if r["kernel"].startswith("k_conv2") or r["kernel"] in ("k_gemm3_nn_qp",):
            print("   ", r["kernel"], r["launches_per_step"], round(r["avg_us"], 1), round(r["ms_per_step"], 3), r.get("frac_of_roofline"))
except Exception as e:
    print("no line", e)
PY
done
