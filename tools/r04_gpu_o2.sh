#!/bin/bash
set -u
OUT=gpurun_out/r04o; mkdir -p $OUT
export PYTHONPATH=.
timeout 900 python -m pytest tests/test_conv_wrw_gpu.py -m gpu -q --timeout 600 > $OUT/pytest3.log 2>&1; echo "rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest3.log | head -30
python - <<'PY'
import torch
from rltime_amd.models.torch import fused
def cl(t): return t.contiguous(memory_format=torch.channels_last)
n = 40960
for (c, hw, k, s) in ((32, 20, 4, 2), (64, 9, 3, 1)):
    o = (hw - k) // s + 1
    x = cl(torch.randn(n, c, hw, hw, device="cuda")); g = cl(torch.randn(n, 64, o, o, device="cuda")); wt = cl(torch.empty(64, c, k, k, device="cuda"))
    fns = {"b3": lambda: fused.conv_wgrad_b3(g, x, wt, (s, s)),
           "miopen": lambda: torch.ops.aten.convolution_backward(g, x, wt, None, [s, s], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])}
    for name, f in fns.items():
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        print("wrw", (c, hw, k, s), name, "ms", round(a.elapsed_time(b) / 10, 3), flush=True)
PY
