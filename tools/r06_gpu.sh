#!/bin/bash
# Round-6 GPU session helper: tools/r06_gpu.sh <tag> <what...>; everything lands under gpurun_out/<tag>/.
set -u
TAG="${1:-r06a}"; shift || true
OUT="gpurun_out/$TAG"; mkdir -p "$OUT"
R="$(pwd)"; export TMPDIR=/tmp
short() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 3), d.get("step_ms"), "gather frac", (d.get("roofline") or {}).get("frac"), "roofline_step", (d.get("roofline_step") or {}).get("frac"))
for k in ((d.get("roofline_all") or {}).get("kernels") or [])[:18]:
    print("  ", k["kernel"], k["launches_per_step"], k["avg_us"], k["ms_per_step"], k.get("frac_of_roofline"))
for n, o in (d.get("other_configs") or {}).items():
    print(n, {k: o.get(k) for k in ("ms_per_step", "value", "error")}, "gather", (o.get("roofline") or {}).get("frac"), "cpu", (o.get("cpu_baseline") or {}).get("value"))
cb = d.get("cpu_baseline") or {}
print("cpu_baseline", cb.get("value"), "linearity", cb.get("linearity_check"))
PY
}
for w in "$@"; do
  case $w in
    tests)   timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 -o faulthandler_timeout=500 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -15 "$OUT/pytest.log";;
    t:*)     f=${w#t:}; timeout 900 python -m pytest tests/$f -x -q --timeout 600 > "$OUT/pytest_${f%%.py*}.log" 2>&1; echo "pytest $f rc=$?"; tail -25 "$OUT/pytest_${f%%.py*}.log";;
    bench)   timeout 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"; tail -c 600 "$OUT/bench_default.err"; short "$OUT/bench_default.json";;
    benchq)  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err"; echo "benchq rc=$?"; tail -c 400 "$OUT/bench_quick.err"; short "$OUT/bench_quick.json";;
    bq:*)    # bq:<name>:<ENV=V,ENV=V>: quick headline under environment switches (A/B experiments)
             n=$(echo "$w" | cut -d: -f2); ev=$(echo "$w" | cut -d: -f3 | tr ',' ' ')
             env $ev timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --profile-steps 2 > "$OUT/bench_$n.json" 2> "$OUT/bench_$n.err"; echo "bq $n rc=$?"; tail -c 300 "$OUT/bench_$n.err"; short "$OUT/bench_$n.json" | head -12;;
    g3:*)    # g3:<name>:<ENV=V,...>: tools/gemm3_probe.py on the step's shapes under environment switches
             n=$(echo "$w" | cut -d: -f2); ev=$(echo "$w" | cut -d: -f3 | tr ',' ' ')
             env $ev timeout 600 python tools/gemm3_probe.py ${GEMM3_SPECS:-nt:1310720x1024x512 tn:1024x512x1310720 nn:1310720x512x1024 head:1310720x1024x512:7} > "$OUT/gemm3_$n.jsonl" 2> "$OUT/gemm3_$n.err"; echo "g3 $n rc=$?"; cat "$OUT/gemm3_$n.jsonl"; tail -3 "$OUT/gemm3_$n.err";;
    share8|share4|share2)
      N=${w#share}
      timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --mbatch $((512/N)) --envs $((256/N)) --replay-size $((1000000/N)) ${SHARE_FLAGS:-} > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; echo "$w rc=$?"; tail -c 400 "$OUT/bench_$w.err"; short "$OUT/bench_$w.json" | head -14;;
    bench23) for c in rainbow_iqn dqn_uniform; do timeout 600 python bench.py --config $c --steps 200 --warmup 20 > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err"; echo "bench $c rc=$?"; tail -c 400 "$OUT/bench_$c.err"; short "$OUT/bench_$c.json" | head -8; done;;
    prof)    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/stats" -o bench -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --profile-steps 0 > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/bench_under_rocprof.err"); echo "prof rc=$?"; python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1; head -60 "$OUT/summary.txt"; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete;;
    rebuildall:*) # rebuildall:<-DFLAG=V,...>: the whole library rebuilt on the box with extra compiler flags (objects do not travel)
             fl=$(echo "$w" | cut -d: -f2 | tr ',' ' '); rm -f rltime_amd/csrc/*.o
             MIRL_EXTRA_HIPCC_FLAGS="$fl" bash rltime_amd/csrc/build.sh > "$OUT/rebuild.log" 2>&1; echo "rebuildall [$fl] rc=$?"; tail -2 "$OUT/rebuild.log";;
    rebuild:*) # rebuild:<unit>:<-DFLAG=V,...>: recompile ONE translation unit on the box with extra flags and relink (same-box A/B of a compile-time switch)
             u=$(echo "$w" | cut -d: -f2); fl=$(echo "$w" | cut -d: -f3 | tr ',' ' ')
             ( cd rltime_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $fl -c $u.hip -o $u.o \
               && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC replay.o qmath.o lstm.o lstm_seq.o convert.o nnops.o acting.o actnet.o conv_in.o conv_mid.o gemm3.o conv3.o conv_wrw.o optim.o -o ../librltime_hip.so ); echo "rebuild $u [$fl] rc=$?";;
    c3:*)    # c3:<name>:<ENV=V,...>: tools/conv3_probe.py under environment switches
             n=$(echo "$w" | cut -d: -f2); ev=$(echo "$w" | cut -d: -f3 | tr ',' ' ')
             env $ev timeout 300 python tools/conv3_probe.py ${CONV3_FRAMES:-41472} > "$OUT/conv3_$n.jsonl" 2> "$OUT/conv3_$n.err"; echo "c3 $n rc=$?"; cat "$OUT/conv3_$n.jsonl"; tail -3 "$OUT/conv3_$n.err";;
    variants) for v in no-acting no-policy-outputs; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --profile-steps 0 --$v > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"; echo "variant $v rc=$?"; short "$OUT/bench_$v.json" | head -1; done;;
    smoke)   timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3;;
    *)       echo "unknown item $w";;
  esac
done
