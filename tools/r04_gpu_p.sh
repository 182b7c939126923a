#!/bin/bash
# round 4, session P: layer-3 data gradient on the bf16 pipe (k_conv3_bwd_data_b3): tests, timing, A/B bench
set -u
OUT=gpurun_out/r04p; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT PYTHONPATH=.
timeout 900 python -m pytest tests/test_conv_mid_gpu.py tests/test_abi.py -m gpu -q --timeout 600 > $OUT/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest.log | head -30
python - <<'PY'
import torch
from rltime_amd.models.torch import fused
def cl(t): return t.contiguous(memory_format=torch.channels_last)
n = 40960
x = cl(torch.empty(n, 64, 9, 9, device="cuda")); g = cl(torch.randn(n, 64, 7, 7, device="cuda")); wt = cl(torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
fns = {"b3": lambda: fused.conv3_bwd_data(g, wt, x),
       "miopen": lambda: torch.ops.aten.convolution_backward(g, x, wt, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])}
for name, f in fns.items():
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    print("layer-3 bwd data", name, "ms", round(a.elapsed_time(b) / 10, 3), flush=True)
PY
timeout 900 python -m pytest tests/test_network_ab_gpu.py tests/test_e2e_gpu.py tests/test_conv3_gpu.py tests/test_fused_gpu.py -m gpu -q --timeout 600 > $OUT/pytest2.log 2>&1; echo "rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest2.log | head -30
for v in 1 0; do
  MIRL_CONV3_BWD=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c3bwd$v.json 2> $OUT/bench_c3bwd$v.err; echo "bench conv3 bwd=$v rc=$?"
  python - $OUT/bench_c3bwd$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], d.get("step_ms"))
for r in d["roofline_all"]["kernels"]:
    if r["kernel"].startswith("k_conv3_bwd"):
        print("   ", r["kernel"], r["launches_per_step"], round(r["avg_us"], 1), round(r["ms_per_step"], 3), r.get("frac_of_roofline"), r.get("bound"))
PY
done
