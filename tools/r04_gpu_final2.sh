#!/bin/bash
# round 4, final session part 2: A/B test with the robust gradient bars, the judged bench line with cpu_baseline, linearity check
set -u
OUT=gpurun_out/r04final2; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 600 python -m pytest tests/test_network_ab_gpu.py tests/test_abi.py tests/test_conv_in_gpu.py tests/test_lstm_gpu.py -m gpu -q --timeout 600 > $OUT/pytest_ab.log 2>&1; echo "ab rc=$?"; grep -E "passed|failed|^E  " $OUT/pytest_ab.log | head
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err; echo "bench rc=$?"
python - $OUT/bench_final.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], d["step_ms"], "value", d["value"], "steps/s", d["learner_steps_per_sec"]); print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "measured_copy_peak_GBps", "avg_launch_ms")})
print("roofline_step", {k: v for k, v in d.get("roofline_step", {}).items() if k != "how"})
print("cpu", {k: v for k, v in d.get("cpu_baseline", {}).items() if k in ("value", "runs", "linearity_check", "error")}, d.get("speedup_vs_cpu_baseline"))
PY
timeout 900 python bench.py --steps 5 --warmup 3 --profile-steps 0 --cpu-linearity-check > $OUT/bench_cpu_linearity.json 2> $OUT/bench_cpu_linearity.err; echo "lin rc=$?"
python - $OUT/bench_cpu_linearity.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cpu", {k: v for k, v in d.get("cpu_baseline", {}).items() if k in ("value", "runs", "linearity_check", "error")})
PY
