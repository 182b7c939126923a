#!/bin/bash
set -u
export PYTHONPATH=.
for d in 0 1 2 3 4 6 7; do
MIRL_C2B_DBG=$d python - <<'PY'
import torch, os
from rltime_amd.models.torch import fused
def cl(t): return t.contiguous(memory_format=torch.channels_last)
n=40960
x = cl(torch.empty(n, 32, 20, 20, device="cuda")); wt = cl(torch.randn(64, 32, 4, 4, device="cuda")*0.05); g = cl(torch.randn(n, 64, 9, 9, device="cuda"))
for _ in range(3): fused.conv2_bwd_data(g, wt, x, 1)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): fused.conv2_bwd_data(g, wt, x, 1)
b.record(); torch.cuda.synchronize()
print("dbg", os.environ["MIRL_C2B_DBG"], "ms", round(a.elapsed_time(b)/10, 3))
PY
done
