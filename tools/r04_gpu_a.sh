#!/bin/bash
# round 4, GPU session A: new parity tests, full GPU suite, default bench line with roofline_step, one rank's share of 8 GPUs
set -u
OUT=gpurun_out/r04a; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_network_ab_gpu.py -x -q -s --timeout 600 > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -25 $OUT/pytest_new.log
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 --deselect tests/test_e2e_gpu.py --deselect tests/test_network_ab_gpu.py > $OUT/pytest_all.log 2>&1; echo "suite rc=$?"; tail -6 $OUT/pytest_all.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -c 600 $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "value", d["value"]); print("roofline_step", {k: v for k, v in d.get("roofline_step", {}).items() if k != "how"})
for k in d["roofline_all"]["kernels"][:24]:
    print(k["kernel"], k["launches_per_step"], k["avg_us"], k["ms_per_step"], k.get("bound"), k.get("frac_of_roofline"))
PY
for extra in "" "--no-acting" "--overlap-acting on"; do
  tag=$(echo "share8$extra" | tr -d ' -')
  timeout 300 python bench.py --mbatch 64 --envs 32 --replay-size 125000 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 $extra > $OUT/$tag.json 2> $OUT/$tag.err; echo "$tag rc=$?"
  python - $OUT/$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("  ms/step", d["ms_per_step"], d["step_ms"])
PY
done
BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --mbatch 64 --envs 32 --replay-size 125000 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > $OUT/share8_rccl_world1.json 2> $OUT/share8_rccl_world1.err; echo "rccl world1 rc=$?"
python - $OUT/share8_rccl_world1.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("  ms/step", d["ms_per_step"], d.get("rccl"))
PY
