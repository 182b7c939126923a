#!/bin/bash
set -u
export PYTHONPATH=.
python - <<'PY'
import torch, torch.nn as nn, torch.nn.functional as F
from rltime_amd.models.torch import fused
def cl(t): return t.contiguous(memory_format=torch.channels_last)
for n in (256, 32):
    for (c, hw, k, s) in ((32, 20, 4, 2), (64, 9, 3, 1)):
        conv = nn.Conv2d(c, 64, k, s).cuda().to(memory_format=torch.channels_last)
        x = cl(torch.randn(n, c, hw, hw, device="cuda"))
        def lib():
            y = F.conv2d(x, conv.weight, None, s)
            fused.bias_relu_rows_(y, conv.bias, 64)
            return y
        own = lambda: fused.conv3_bias_relu(x, conv.weight, conv.bias, (s, s))
        with torch.no_grad():
            for name, f in (("miopen+bias_relu", lib), ("k_conv3_fwd", own)):
                for _ in range(5): f()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(20): f()
                g.replay(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10): g.replay()
                b.record(); torch.cuda.synchronize()
                print("frames", n, (c, hw, k, s), name, "us", round(a.elapsed_time(b) / 200 * 1e3, 2), flush=True)
PY
