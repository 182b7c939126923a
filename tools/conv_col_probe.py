"""Time the backward of conv layers 2 / 3 at the learner's frame counts: MIOpen (+ conv_mid's layer-2 data gradient)
against im2col / col2im + split-bf16 GEMMs (csrc/conv_col.hip).  One JSON line per layer and frame count."""
import json
import sys

import torch

from rltime_amd.models.torch import fused


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def timed(fn, reps=10):
    for _ in range(3):
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
    frames = [int(v) for v in sys.argv[1:]] or [5120, 640]
    for n in frames:
        for (c, hw, f, k, s) in ((32, 20, 64, 4, 2), (64, 9, 64, 3, 1)):
            x = cl(torch.randn(n, c, hw, hw, device="cuda"))
            wt = cl(torch.randn(f, c, k, k, device="cuda") * 0.05)
            o = (hw - k) // s + 1
            g = cl(torch.randn(n, f, o, o, device="cuda"))
            rec = {"layer": f"{c}->{f} k{k} s{s}", "frames": n}
            bw = lambda mask: torch.ops.aten.convolution_backward(g, x, wt, None, [s, s], [0, 0], [1, 1], False, [0, 0], 1, mask)
            rec["miopen_wrw_ms"] = timed(lambda: bw([False, True, False]))
            rec["miopen_bwd_data_ms"] = timed(lambda: bw([True, False, False]))
            if fused.conv2_bwd_data_supported(x, wt, (s, s), g):
                rec["conv_mid_bwd_data_ms"] = timed(lambda: fused.conv2_bwd_data(g, wt, x))
            rec["im2col_ms"] = timed(lambda: fused.im2col_nhwc(x, k, k, s))
            rec["col_wgrad_ms"] = timed(lambda: fused.conv_wgrad_col(g, x, wt, (s, s)))
            rec["col_dgrad_ms"] = timed(lambda: fused.conv_dgrad_col(g, x, wt, (s, s)))
            dcol = torch.randn(n * o * o, k * k * c, device="cuda")
            rec["col2im_ms"] = timed(lambda: fused.col2im_nhwc(dcol, x, k, k, s))
            rec["col2im_masked_ms"] = timed(lambda: fused.col2im_nhwc(dcol, x, k, k, s, relu_mask=x))
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
