"""Time csrc/conv3.hip (split-bf16 implicit GEMM, bias + ReLU fused) against MIOpen conv + the in-place bias/ReLU pass
on the Atari models' layers 2 and 3 at the learner step's batch.  usage: python tools/conv3_probe.py [frames]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from rltime_amd.models.torch import fused


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 41472
    for c, hw, f, k, s in ((32, 20, 64, 4, 2), (64, 9, 64, 3, 1)):
        conv = nn.Conv2d(c, f, k, s).cuda().to(memory_format=torch.channels_last)
        x = torch.randn(n, c, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
        oh = (hw - k) // s + 1
        flop = 2.0 * n * oh * oh * f * c * k * k
        with torch.no_grad():
            t3 = timed(lambda: fused.conv3_bias_relu(x, conv.weight, conv.bias, conv.stride))
            os.environ["MIRL_CONV3"] = "0"
            fused._CONV3 = False
            tl = timed(lambda: fused.conv_bias_relu(x, conv))
            fused._CONV3 = True
            y3 = fused.conv3_bias_relu(x, conv.weight, conv.bias, conv.stride)
            ref = torch.relu(torch.nn.functional.conv2d(x[:64].double(), conv.weight.double(), conv.bias.double(), s))
            err = float((y3[:64].double() - ref).abs().max()) / float(ref.abs().max())
        print(json.dumps({"layer": "%d->%d k%d s%d on %dx%d" % (c, f, k, s, hw, hw), "frames": n, "ms_conv3": round(t3, 4),
                          "ms_miopen_plus_bias_relu": round(tl, 4), "tflops_conv3": round(flop / t3 / 1e9, 1),
                          "tflops_library": round(flop / tl / 1e9, 1), "frac_of_bf16x6_peak": round(flop / t3 / 1e9 / 416.7, 3),
                          "l2_to_cu_GBps_conv3": round((n * oh * oh * (c * k * k) * 4 + (n * oh * oh / 256) * f * c * k * k * 4) / t3 / 1e6, 0),
                          "max_err_vs_float64": err}), flush=True)


if __name__ == "__main__":
    main()
