#!/bin/bash
# round 4, session J: the feature product's backward in the data-gradient GEMM's epilogue (gemm3 EP 3): tests, A/B bench
set -u
OUT=gpurun_out/r04j; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 900 python -m pytest tests/test_gemm3_gpu.py tests/test_conv_col_gpu.py tests/test_abi.py tests/test_fused_gpu.py tests/test_network_ab_gpu.py -m gpu -q --timeout 600 > $OUT/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest.log | head -30
for v in 1 0; do
  MIRL_GEMM3_QP_BWD=$v MIRL_CONV_COL=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_qp$v.json 2> $OUT/bench_qp$v.err; echo "bench qp=$v rc=$?"
  python - $OUT/bench_qp$v.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms/step", d["ms_per_step"], d.get("step_ms"))
    for r in d["roofline_all"]["kernels"]:
        if r["kernel"] in ("k_gemm3_nn", "k_gemm3_nn_qp", "k_iqn_mul_bwd", "k_gemm3_tn"):
            print("   ", r["kernel"], r["launches_per_step"], round(r["avg_us"], 1), round(r["ms_per_step"], 3))
except Exception as e:
    print("no line", e)
PY
done
