"""Weight gradient of conv layers 2-3 at small frame counts: csrc/conv_wrw.hip against MIOpen's.  usage: python tools/conv_wrw_small_probe.py [frames ...]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rltime_amd.models.torch import fused


def timed(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


for n in [int(v) for v in sys.argv[1:]] or [256, 512, 1536]:
    for c, hw, f, k, s in ((32, 20, 64, 4, 2), (64, 9, 64, 3, 1)):
        oh = (hw - k) // s + 1
        x = cl(torch.randn(n, c, hw, hw, device="cuda")); g = cl(torch.randn(n, f, oh, oh, device="cuda"))
        w = cl(torch.randn(f, c, k, k, device="cuda") * 0.05)
        ok = fused.conv_wrw_supported(x, w, (s, s), g, min_work=0)
        rec = {"layer": "%d->%d k%d s%d" % (c, f, k, s), "frames": n, "supported": bool(ok)}
        lib = lambda: torch.ops.aten.convolution_backward(g, x, w, None, (s, s), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False))[1]
        rec["ms_miopen"] = round(timed(lib), 4)
        if ok:
            rec["ms_conv_wrw_b3"] = round(timed(lambda: fused.conv_wgrad_b3(g, x, w, (s, s))), 4)
            ref = lib()
            rec["max_diff_vs_miopen"] = float((fused.conv_wgrad_b3(g, x, w, (s, s)) - ref).abs().max() / ref.abs().max())
        print(json.dumps(rec), flush=True)
