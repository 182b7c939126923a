#!/bin/bash
# round 4, GPU session F: A/B facts, copy probe, rocprofv3 family split of the current step, acting cost, rank shares, cpu linearity
set -u
OUT=gpurun_out/r04f; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 600 python -m pytest tests/test_network_ab_gpu.py -q -s --timeout 600 > $OUT/pytest_ab.log 2>&1; echo "ab rc=$?"; grep -E "passed|failed|^E  |^\{" $OUT/pytest_ab.log | cut -c1-1800 | head -12
timeout 300 python tools/copy_probe.py > $OUT/copy_probe.jsonl 2> $OUT/copy_probe.err; cat $OUT/copy_probe.jsonl; tail -2 $OUT/copy_probe.err
R="$(pwd)"; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats" -o bench -- python "$R/bench.py" --steps 5 --warmup 3 --no-cpu-baseline --profile-steps 0 > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/bench_under_rocprof.err"); echo "prof rc=$?"
python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1; head -75 "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete
run() { tag=$1; shift; timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; echo "$tag rc=$?"; python - $OUT/$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("  ms/step", round(d["ms_per_step"], 3), "median", round(d["step_ms"]["median"], 3))
except Exception as e:
    print("  no line", e)
PY
}
run noacting --no-acting
run share2 --mbatch 256 --envs 128 --replay-size 500000
run share4 --mbatch 128 --envs 64 --replay-size 250000
run share8 --mbatch 64 --envs 32 --replay-size 125000
run share8_overlap --mbatch 64 --envs 32 --replay-size 125000 --overlap-acting on
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-linearity-check --cpu-seconds 30 > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench full rc=$?"
python - $OUT/bench_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], d["step_ms"], "value", d["value"]); print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "measured_copy_peak_GBps", "avg_launch_ms")})
print("roofline_step", {k: v for k, v in d.get("roofline_step", {}).items() if k != "how"})
print("cpu", {k: v for k, v in d.get("cpu_baseline", {}).items() if k in ("value", "runs", "linearity_check")})
PY
