#!/bin/bash
# round 4, GPU session C: pre-split weights + quantile-product epilogue probes, A/B + wide e2e diagnostics, default bench
set -u
OUT=gpurun_out/r04c; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 600 python tools/gemm3_probe.py nt:1310720x1024x512 nn:1310720x512x1024 nt:1310720x512x512 nt:40960x2048x3136 qp:1310720x512x64:32 > $OUT/gemm3_probe.jsonl 2> $OUT/gemm3_probe.err; echo "probe rc=$?"; cat $OUT/gemm3_probe.jsonl; tail -3 $OUT/gemm3_probe.err
timeout 900 python -m pytest tests/test_gemm3_gpu.py tests/test_e2e_gpu.py tests/test_network_ab_gpu.py tests/test_fast_acting_gpu.py tests/test_qmath_gpu.py tests/test_fused_gpu.py -q -s --timeout 600 > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?"; grep -E "passed|failed|^E  " $OUT/pytest_new.log | head -40; cat $OUT/wide_e2e_deviation.txt; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r04c/network_ab.json"))
    print(json.dumps({k: v for k, v in d.items() if k not in ("grad_dev",)}, indent=0)[:3000])
except Exception as e:
    print("no network_ab.json", e)
PY
for ps in 1 0; do
  MIRL_GEMM3_PRESPLIT=$ps timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > $OUT/bench_presplit$ps.json 2> $OUT/bench_presplit$ps.err; echo "bench presplit=$ps rc=$?"
  python - $OUT/bench_presplit$ps.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("  ms/step", round(d["ms_per_step"], 3), d["step_ms"])
PY
done
