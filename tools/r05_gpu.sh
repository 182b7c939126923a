#!/bin/bash
# Round-5 GPU session helper: tools/r05_gpu.sh <tag> <what...>; everything lands under gpurun_out/<tag>/.
set -u
TAG="${1:-r05a}"; shift || true
OUT="gpurun_out/$TAG"; mkdir -p "$OUT"
R="$(pwd)"; export TMPDIR=/tmp
stats() {  # <dir> <steps>: per-kernel table of a rocprofv3 --stats run, per vector step
  python - "$1" "$2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
steps = float(sys.argv[2])
print("%8s %9s %9s  %s" % ("calls/st", "us/step", "avg_us", "kernel"))
for r in rows[:70]:
    print("%8.2f %9.2f %9.2f  %s" % (int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e3 / steps, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
print("total us/step", sum(float(r["TotalDurationNs"]) for r in rows) / 1e3 / steps)
PY
}
for w in "$@"; do
  case $w in
    actnet)  timeout 600 python -m pytest tests/test_actnet_gpu.py -q --timeout 300 > "$OUT/pytest_actnet.log" 2>&1; echo "actnet rc=$?"; tail -25 "$OUT/pytest_actnet.log";;
    facting) timeout 900 python -m pytest tests/test_fast_acting_gpu.py -q --timeout 600 > "$OUT/pytest_fast_acting.log" 2>&1; echo "facting rc=$?"; tail -25 "$OUT/pytest_fast_acting.log";;
    tests)   timeout 1200 python -m pytest tests -m gpu -x -q --timeout 600 -o faulthandler_timeout=500 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -15 "$OUT/pytest.log";;
    act32|act256|act32old|act256old)
      E=${w#act}; E=${E%old}; X=""; case $w in *old) X="MIRL_ACT_FUSED=0";; esac
      (cd /tmp && env $X timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/${w}_stats" -o act -- python "$R/tools/acting_probe.py" 200 --envs=$E > "$R/$OUT/$w.json" 2> "$R/$OUT/$w.err"); echo "$w rc=$?"; cat "$OUT/$w.json"; tail -3 "$OUT/$w.err"
      stats "$OUT/${w}_stats" 230 | tee "$OUT/${w}_kernels.txt" | head -34
      find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete;;
    probe32|probe256|probe32old|probe256old)
      E=${w#probe}; E=${E%old}; X=""; case $w in *old) X="MIRL_ACT_FUSED=0";; esac
      env $X timeout 300 python tools/acting_probe.py 400 --envs=$E > "$OUT/$w.json" 2> "$OUT/$w.err"; echo "$w rc=$?"; cat "$OUT/$w.json"; tail -2 "$OUT/$w.err";;
    share8|share8old|share4|share2|share8serial|share4serial|share2serial)
      N=${w#share}; N=${N%old}; N=${N%serial}; X=""; F=""; case $w in *old) X="MIRL_ACT_FUSED=0";; *serial) F="--train-arg overlap_passes=false";; esac
      env $X timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline $F --mbatch $((512/N)) --envs $((256/N)) --replay-size $((1000000/N)) > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; echo "$w rc=$?"; tail -c 400 "$OUT/bench_$w.err"
      python - "$OUT/bench_$w.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 3), d["step_ms"], "roofline_step", (d.get("roofline_step") or {}).get("frac"))
PY
      ;;
    benchq|benchqold|benchq_nopol|benchq_noact)
      X=""; F=""; case $w in *old) X="MIRL_ACT_FUSED=0";; *_nopol) F="--no-policy-outputs";; *_noact) F="--no-acting";; esac
      env $X timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $F > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; echo "$w rc=$?"; tail -c 400 "$OUT/bench_$w.err"
      python - "$OUT/bench_$w.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 3), d["step_ms"], "roofline_step", (d.get("roofline_step") or {}).get("frac"))
for k in d["roofline_all"]["kernels"][:14]:
    print(k["kernel"], k["launches_per_step"], k["avg_us"], k["ms_per_step"], k.get("frac_of_roofline"))
PY
      ;;
    bench)   timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_iqn_lstm.json" 2> "$OUT/bench_iqn_lstm.err"; echo "bench rc=$?"; tail -c 800 "$OUT/bench_iqn_lstm.err"; head -c 3000 "$OUT/bench_iqn_lstm.json"; echo;;
    bench23) for c in rainbow_iqn dqn_uniform; do timeout 600 python bench.py --config $c --steps 200 --warmup 20 > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err"; echo "bench $c rc=$?"; tail -c 400 "$OUT/bench_$c.err"; python - "$OUT/bench_$c.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 3), d["step_ms"], "roofline", d["roofline"]["frac"], "roofline_step", (d.get("roofline_step") or {}).get("frac"))
PY
      done;;
    prof)    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats" -o bench -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0 > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/bench_under_rocprof.err"); echo "prof rc=$?"; python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1; head -70 "$OUT/summary.txt"; python tools/trace_window.py "$OUT/stats" > "$OUT/headline_one_step_in_order.txt" 2>&1; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete;;
    profrainbow|profdqn)
      c=rainbow_iqn; case $w in profdqn) c=dqn_uniform;; esac
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/${w}_stats" -o bench -- python "$R/bench.py" --config $c --steps 100 --warmup 20 --no-cpu-baseline --profile-steps 0 > "$R/$OUT/$w.json" 2> "$R/$OUT/$w.err"); echo "$w rc=$?"; tail -c 300 "$OUT/$w.err"
      stats "$OUT/${w}_stats" 120 | tee "$OUT/${w}_kernels.txt" | head -20
      python tools/trace_window.py "$OUT/${w}_stats" > "$OUT/${w}_one_step_in_order.txt" 2>&1; head -150 "$OUT/${w}_one_step_in_order.txt"
      find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete;;
    profshare8)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/${w}_stats" -o bench -- python "$R/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --profile-steps 0 --mbatch 64 --envs 32 --replay-size 125000 > "$R/$OUT/$w.json" 2> "$R/$OUT/$w.err"); echo "$w rc=$?"; tail -c 300 "$OUT/$w.err"
      python tools/trace_window.py "$OUT/${w}_stats" > "$OUT/${w}_one_step_in_order.txt" 2>&1; head -3 "$OUT/${w}_one_step_in_order.txt"
      find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete;;
    smoke)   timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -3 "$OUT/smoke.log";;
    *) echo "unknown: $w";;
  esac
done
