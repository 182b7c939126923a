#!/bin/bash
# round 4, GPU session G: masked conv1 wrw, env clock without atomics, gather variant 2, whole suite, bench
set -u
OUT=gpurun_out/r04g; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_all.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest_all.log | head -30
run() { tag=$1; shift; timeout 400 env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA > $OUT/$tag.json 2> $OUT/$tag.err; echo "$tag rc=$?"; python - $OUT/$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("  ms/step", round(d["ms_per_step"], 3), "median", round(d["step_ms"]["median"], 3), "gather", round(d["roofline"]["avg_launch_ms"], 4), round(d["roofline"]["frac"], 4))
    ks = {k["kernel"]: k for k in (d.get("roofline_all") or {}).get("kernels", [])}
    for n in ("k_conv1_u8_wrw", "k_relu_bwd_bias_rows", "k_gather_rows(frames)"):
        if n in ks: print("   ", n, ks[n]["launches_per_step"], ks[n]["avg_us"], ks[n]["ms_per_step"])
except Exception as e:
    print("  no line", e)
PY
}
EXTRA="" run bench_default A=1
EXTRA="" run bench_gather_v2 MIRL_GATHER_VARIANT=2
EXTRA="" run bench_wrw_unmasked MIRL_CONV1_WRW_MASK=0
EXTRA="--no-acting --profile-steps 0" run bench_noacting A=1
