#!/bin/bash
# round 4, session R: conv layers 2-3 forward with whole frames staged and split once (k_conv3_fwd_wf): tests, timing, A/B bench
set -u
OUT=gpurun_out/r04r; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT PYTHONPATH=.
timeout 900 python -m pytest tests/test_conv3_gpu.py tests/test_fast_acting_gpu.py -m gpu -q --timeout 600 > $OUT/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest.log | head -30
for wf in 1 0; do
MIRL_CONV3_WF=$wf python - <<'PY'
import os, torch
from rltime_amd.models.torch import fused
def cl(t): return t.contiguous(memory_format=torch.channels_last)
n = 40960
for (c, hw, k, s) in ((32, 20, 4, 2), (64, 9, 3, 1)):
    x = cl(torch.randn(n, c, hw, hw, device="cuda")); wt = cl(torch.randn(64, c, k, k, device="cuda") * 0.05); b = torch.randn(64, device="cuda")
    f = lambda: fused.conv3_bias_relu(x, wt, b, (s, s))
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    print("conv fwd", (c, hw, k, s), "whole-frame" if os.environ["MIRL_CONV3_WF"] == "1" else "streaming", "ms", round(a.elapsed_time(e) / 10, 3), flush=True)
PY
done
