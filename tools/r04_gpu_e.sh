#!/bin/bash
# round 4, GPU session E: fused output layers (EP 2), whole GPU suite, bench
set -u
OUT=gpurun_out/r04e; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 600 python -m pytest tests/test_gemm3_gpu.py tests/test_fused_gpu.py -q --timeout 600 > $OUT/pytest_gemm3.log 2>&1; echo "gemm3 tests rc=$?"; grep -E "passed|failed|^E  " $OUT/pytest_gemm3.log | head -30
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gemm3_gpu.py --deselect tests/test_fused_gpu.py > $OUT/pytest_all.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest_all.log | head -40; cat $OUT/wide_e2e_deviation.txt
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r04e/network_ab.json"))
    print(json.dumps({k: v for k, v in d.items() if k not in ("grad_dev", "vs_float64_b16")}))
    print({m: {k: v for k, v in d["vs_float64_b16"][m].items() if k != "grad_dev"} for m in ("hip", "lib")})
except Exception as e:
    print("no network_ab.json", e)
PY
for hd in 1 0; do
  MIRL_GEMM3_HEAD=$hd timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_head$hd.json 2> $OUT/bench_head$hd.err; echo "bench head=$hd rc=$?"; tail -c 200 $OUT/bench_head$hd.err
  python - $OUT/bench_head$hd.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  ms/step", round(d["ms_per_step"], 3), d["step_ms"]["median"], "roofline_step", d.get("roofline_step", {}).get("frac"))
for k in d["roofline_all"]["kernels"][:14]:
    print("   ", k["kernel"], k["launches_per_step"], k["avg_us"], k["ms_per_step"], k.get("bound"), k.get("frac_of_roofline"))
PY
done
