#!/bin/bash
# round 4, GPU session D: narrow TN product, PMC of the pre-split experiment, full tests, bench lines
set -u
OUT=gpurun_out/r04d; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 300 python tools/gemm3_probe.py tn:512x64x1310720 > $OUT/gemm3_probe_narrow.jsonl 2> $OUT/probe.err; MIRL_GEMM3_NARROW=0 timeout 300 python tools/gemm3_probe.py tn:512x64x1310720 >> $OUT/gemm3_probe_narrow.jsonl 2>> $OUT/probe.err; cat $OUT/gemm3_probe_narrow.jsonl
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -x > $OUT/pytest_all.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^E  " $OUT/pytest_all.log | head -30
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r04d/network_ab.json"))
    print(json.dumps({k: v for k, v in d.items() if k not in ("grad_dev",)})[:2500])
    print({m: {k: v for k, v in d["vs_float64_b16"][m].items() if k != "grad_dev"} for m in ("hip", "lib")})
except Exception as e:
    print("no network_ab.json", e)
PY
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -c 300 $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], d["step_ms"], "value", d["value"]); print("roofline_step", {k: v for k, v in d.get("roofline_step", {}).items() if k != "how"})
print("cpu", {k: v for k, v in d.get("cpu_baseline", {}).items() if k in ("value", "runs")})
for k in d["roofline_all"]["kernels"][:16]:
    print(k["kernel"], k["launches_per_step"], k["avg_us"], k["ms_per_step"], k.get("bound"), k.get("frac_of_roofline"))
PY
timeout 300 python bench.py --mbatch 64 --envs 32 --replay-size 125000 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > $OUT/share8.json 2> $OUT/share8.err; python - $OUT/share8.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("share8 ms/step", round(d["ms_per_step"], 3), d["step_ms"]["median"])
PY
GEMM3_SPECS="nt:1310720x1024x512" bash tools/gpu_round.sh r04d g3pmc > $OUT/g3pmc.log 2>&1; tail -60 $OUT/g3pmc.log
