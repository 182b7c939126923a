#!/bin/bash
# round 4, GPU session B: rollout graph + pinned tests, then one rank's share of the 8-GPU job with / without the rollout graph
set -u
OUT=gpurun_out/r04b; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 900 python -m pytest tests/test_fast_acting_gpu.py tests/test_e2e_gpu.py tests/test_network_ab_gpu.py -q -s --timeout 600 > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?"; grep -E "passed|failed|Error|error|assert|^E " $OUT/pytest_new.log | head -40; grep -E "wide IQN|IQN-LSTM e2e|max rel dev|^\{" $OUT/pytest_new.log | cut -c1-1500
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 --deselect tests/test_e2e_gpu.py --deselect tests/test_network_ab_gpu.py --deselect tests/test_fast_acting_gpu.py > $OUT/pytest_all.log 2>&1; echo "suite rc=$?"; tail -6 $OUT/pytest_all.log
S8="--mbatch 64 --envs 32 --replay-size 125000 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0"
run() { tag=$1; shift; timeout 300 env "$@" python bench.py $S8 $EXTRA > $OUT/$tag.json 2> $OUT/$tag.err; echo "$tag rc=$?"; python - $OUT/$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("  ms/step", round(d["ms_per_step"], 3), d["step_ms"]["median"])
except Exception as e:
    print("  no line", e)
PY
}
EXTRA="" run share8_rollout_graph MIRL_ROLLOUT_GRAPH=1
EXTRA="" run share8_per_step MIRL_ROLLOUT_GRAPH=0
EXTRA="--overlap-acting on" run share8_overlap_rollout MIRL_ROLLOUT_GRAPH=1
EXTRA="--no-acting" run share8_noacting MIRL_ROLLOUT_GRAPH=1
R="$(pwd)"; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats" -o bench -- python "$R/bench.py" --mbatch 64 --envs 32 --replay-size 125000 --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 > "$R/$OUT/share8_under_rocprof.json" 2> "$R/$OUT/share8_under_rocprof.err"); echo "prof rc=$?"
python tools/summarize_prof.py "$OUT" > "$OUT/share8_summary.txt" 2>&1; head -70 "$OUT/share8_summary.txt"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -c 300 $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "value", d["value"]); print("roofline_step", {k: v for k, v in d.get("roofline_step", {}).items() if k != "how"})
PY
