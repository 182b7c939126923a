#!/bin/bash
# round 4, final evidence session: suite, smoke, the judged bench line, rocprofv3 split, rank shares, other configs
set -u
OUT=gpurun_out/r04final4; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_all.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest_all.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-linearity-check > $OUT/bench_final.json 2> $OUT/bench_final.err; echo "bench rc=$?"; tail -c 300 $OUT/bench_final.err
python - $OUT/bench_final.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], d["step_ms"], "value", d["value"]); print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "measured_copy_peak_GBps", "avg_launch_ms")})
print("roofline_step", {k: v for k, v in d.get("roofline_step", {}).items() if k != "how"})
print("cpu", {k: v for k, v in d.get("cpu_baseline", {}).items() if k in ("value", "runs", "linearity_check")}, d.get("speedup_vs_cpu_baseline"))
for k in d["roofline_all"]["kernels"][:40]:
    print(k["kernel"], k["launches_per_step"], k["avg_us"], k["ms_per_step"], k.get("bound"), k.get("frac_of_roofline"))
PY
R="$(pwd)"; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats" -o bench -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/bench_under_rocprof.err"); echo "prof rc=$?"
python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1; head -48 "$OUT/summary.txt"; cp "$OUT/stats/bench_kernel_stats.csv" "$OUT/kernel_stats_full.csv" 2>/dev/null
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete
run() { tag=$1; shift; timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err; echo "$tag rc=$?"; python - $OUT/$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("  ms/step", round(d["ms_per_step"], 3), "median", round(d["step_ms"]["median"], 3), "value", round(d["value"]))
except Exception as e:
    print("  no line", e)
PY
}
run share2 --mbatch 256 --envs 128 --replay-size 500000
run share4 --mbatch 128 --envs 64 --replay-size 250000
run share8 --mbatch 64 --envs 32 --replay-size 125000
run share8_overlap --mbatch 64 --envs 32 --replay-size 125000 --overlap-acting on --profile-steps 0
run noacting --no-acting --profile-steps 0
run dedup --frame-dedup --profile-steps 0
timeout 500 python bench.py --config rainbow_iqn --steps 50 --warmup 10 --no-cpu-baseline > $OUT/rainbow_iqn.json 2> $OUT/rainbow_iqn.err; echo "rainbow rc=$?"; python -c "
import json; d=json.loads(open('$OUT/rainbow_iqn.json').read().strip().splitlines()[-1]); print('  ms/step', d['ms_per_step'], d['value'])"
timeout 500 python bench.py --config dqn_uniform --steps 50 --warmup 10 --no-cpu-baseline > $OUT/dqn_uniform.json 2> $OUT/dqn_uniform.err; echo "dqn rc=$?"; python -c "
import json; d=json.loads(open('$OUT/dqn_uniform.json').read().strip().splitlines()[-1]); print('  ms/step', d['ms_per_step'], d['value'])"
BENCH_FORCE_DIST=1 timeout 400 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > $OUT/selflaunch_rccl_world1.json 2> $OUT/selflaunch_rccl_world1.err; echo "selflaunch rc=$?"; wc -l $OUT/selflaunch_rccl_world1.json; python -c "
import json; d=json.loads(open('$OUT/selflaunch_rccl_world1.json').read().strip().splitlines()[-1]); print('  ms/step', d['ms_per_step'], d.get('rccl'))"
timeout 900 python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 > $OUT/gpus2_shared_gloo.json 2> $OUT/gpus2_shared_gloo.err; echo "gpus2 rc=$?"; wc -l $OUT/gpus2_shared_gloo.json; python -c "
import json; d=json.loads(open('$OUT/gpus2_shared_gloo.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('value','n_gpus','ms_per_step','scaling')}, 'weak' in d, 'overlapped_acting' in d)"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats8" -o bench -- python "$R/bench.py" --mbatch 64 --envs 32 --replay-size 125000 --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 > "$R/$OUT/share8_under_rocprof.json" 2> "$R/$OUT/share8_under_rocprof.err"); echo "prof8 rc=$?"
mkdir -p $OUT/s8 && cp -r $OUT/stats8 $OUT/s8/stats && python tools/summarize_prof.py "$OUT/s8" > "$OUT/share8_summary.txt" 2>&1; head -30 "$OUT/share8_summary.txt"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete
