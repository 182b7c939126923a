#!/bin/bash
set -u
OUT=gpurun_out/r04l; mkdir -p $OUT
export PYTHONPATH=.
timeout 600 python -m pytest tests/test_conv_mid_gpu.py -m gpu -q --timeout 600 > $OUT/pytest2.log 2>&1; echo "rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest2.log | head -10
python - <<'PY'
import torch, ctypes as C
from rltime_amd.models.torch import fused
from rltime_amd import _lib
def cl(t): return t.contiguous(memory_format=torch.channels_last)
n=40960
x = cl(torch.empty(n, 32, 20, 20, device="cuda")); wt = cl(torch.randn(64, 32, 4, 4, device="cuda")*0.05); g = cl(torch.randn(n, 64, 9, 9, device="cuda"))
for pipe in (1, 0):
    for _ in range(3): fused.conv2_bwd_data(g, wt, x, pipe)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fused.conv2_bwd_data(g, wt, x, pipe)
    b.record(); torch.cuda.synchronize()
    print("pipe", pipe, "ms", a.elapsed_time(b)/10)
PY
