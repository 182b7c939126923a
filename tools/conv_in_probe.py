"""Timing probe for the one-pass input layer (csrc/conv_in.hip) against the path it
replaces (convert.hip + MIOpen conv + bias/ReLU pass), HIP events on the launch
stream.  One JSON line per variant.  Usage: python tools/conv_in_probe.py [frames ...]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn as nn

from rltime_amd._lib import lib, check
from rltime_amd.models.torch.fused import conv_bias_relu, frames_to_f32_nhwc

PEAK_TFLOPS = 157.3


def timed(fn, iters):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [32, 20480, 41472]
    conv = nn.Conv2d(4, 32, 8, 4).cuda().to(memory_format=torch.channels_last)
    p = lambda t: C.c_void_p(t.data_ptr())
    for n in sizes:
        x = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device="cuda")
        y = torch.empty((n, 32, 20, 20), device="cuda").contiguous(memory_format=torch.channels_last)
        wpk = torch.empty(12288, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        so, sc, sh, sw = conv.weight.stride()
        flop = n * 400 * 2.0 * 256 * 32
        iters = 20 if n > 1000 else 200
        with torch.no_grad():
            ms = timed(lambda: conv_bias_relu(frames_to_f32_nhwc(x, 1 / 255.0), conv), iters)
            xf = frames_to_f32_nhwc(x, 1 / 255.0)
            ms_conv = timed(lambda: conv_bias_relu(xf, conv), iters)
            del xf
        print(json.dumps({"frames": n, "variant": "convert + MIOpen conv + bias/ReLU pass", "ms": round(ms, 4),
                          "ms_without_convert": round(ms_conv, 4), "tflops": round(flop / ms / 1e9, 1)}), flush=True)
        if n > 1000:
            variants = [("default", None), ("fpi2 nt", 2 << 8), ("fpi1 nt", 1 << 8), ("fpi2 cached", (2 << 8) | 1),
                        ("fpi2 nt conversions-before-chain", (2 << 8) | 4)]
            # timing experiments (results are NOT the convolution): what each part costs
            for dbg, what in ((1, "no u8->f32 conversion"), (2, "no output stores"), (4, "no LDS refill"), (7, "MFMA chain only")):
                variants.append(("fpi2 nt EXPERIMENT " + what, (2 << 8) | (dbg << 24)))
        else:
            variants = [("default", None)] + [("fpi1 split%d %s" % (s, "cached" if c else "nt"), c | (1 << 8) | (s << 16))
                                             for s in (0, 2, 4, 7) for c in (0, 1)]
        for name, flags in variants:
            if flags is None:
                fn = lambda: check(lib.mirl_conv1_u8_fwd(n, 84, 84, p(x), p(conv.weight), so, sc, sh, sw, p(conv.bias),
                                                         1 / 255.0, p(wpk), p(y), st))
            else:
                fn = lambda flags=flags: check(lib.mirl_conv1_u8_fwd_ex(n, 84, 84, p(x), p(conv.weight), so, sc, sh, sw,
                                                                         p(conv.bias), 1 / 255.0, p(wpk), p(y), flags, st))
            ms = timed(fn, iters)
            print(json.dumps({"frames": n, "variant": "conv1_u8 " + name, "ms": round(ms, 4),
                              "tflops": round(flop / ms / 1e9, 1), "frac_of_f32_mfma_peak": round(flop / ms / 1e9 / PEAK_TFLOPS, 3),
                              "hbm_GBps": round(n * (28224 + 51200) / ms / 1e6, 1)}), flush=True)


def wrw_probe(sizes):
    from rltime_amd.models.torch.fused import relu_bwd_bias_rows  # noqa: F401
    p = lambda t: C.c_void_p(t.data_ptr())
    need = C.c_int64()
    check(lib.mirl_conv1_u8_wrw_scratch_floats(C.byref(need)))
    for n in sizes:
        x = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device="cuda")
        g = torch.randn(n, 32, 20, 20, device="cuda").contiguous(memory_format=torch.channels_last)
        wt = torch.empty(32, 4, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)
        dw = torch.empty_like(wt)
        scratch = torch.empty(need.value, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        so, sc, sh, sw = dw.stride()
        flop = n * 400 * 2.0 * 256 * 32
        iters = 20

        def lib_path():
            xf = frames_to_f32_nhwc(x, 1 / 255.0)
            return torch.ops.aten.convolution_backward(g, xf, wt, None, [4, 4], [0, 0], [1, 1], False, [0, 0], 1,
                                                       [False, True, False])[1]
        ms = timed(lib_path, iters)
        print(json.dumps({"frames": n, "variant": "weight gradient: convert + MIOpen wrw", "ms": round(ms, 4),
                          "tflops": round(flop / ms / 1e9, 1)}), flush=True)
        for name, fl in (("conversions hoisted per k-step", 0), ("conversions interleaved", 1)):
            ms = timed(lambda fl=fl: check(lib.mirl_conv1_u8_wrw_ex(n, 84, 84, p(x), p(g), 1 / 255.0, p(scratch), p(dw), so, sc, sh,
                                                                    sw, fl, st)), iters)
            print(json.dumps({"frames": n, "variant": "weight gradient: conv1_u8_wrw (+ slab reduce), " + name, "ms": round(ms, 4),
                              "tflops": round(flop / ms / 1e9, 1), "frac_of_f32_mfma_peak": round(flop / ms / 1e9 / PEAK_TFLOPS, 3)}),
                  flush=True)


def bwd2_probe(sizes):
    p = lambda t: C.c_void_p(t.data_ptr())
    for n in sizes:
        g = torch.randn(n, 64, 9, 9, device="cuda").contiguous(memory_format=torch.channels_last)
        x = torch.empty(n, 32, 20, 20, device="cuda").contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(64, 32, 4, 4, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        wpk = torch.empty(32768, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        so, sc, sh, sw = wt.stride()
        useful = n * 81 * 2.0 * 512 * 64
        issued = n * 400 * 2.0 * 256 * 32
        ms = timed(lambda: torch.ops.aten.convolution_backward(g, x, wt, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1,
                                                               [True, False, False]), 20)
        print(json.dumps({"frames": n, "variant": "conv2 data gradient: MIOpen", "ms": round(ms, 4),
                          "useful_tflops": round(useful / ms / 1e9, 1)}), flush=True)
        ms = timed(lambda: check(lib.mirl_conv2_bwd_data(n, 9, 9, p(g), p(wt), so, sc, sh, sw, p(wpk), p(dx), st)), 20)
        print(json.dumps({"frames": n, "variant": "conv2 data gradient: conv2_bwd_data", "ms": round(ms, 4),
                          "useful_tflops": round(useful / ms / 1e9, 1), "issued_tflops": round(issued / ms / 1e9, 1),
                          "issued_frac_of_f32_mfma_peak": round(issued / ms / 1e9 / PEAK_TFLOPS, 3)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "all":          # one process for a counter pass
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 42496
        sys.argv = [sys.argv[0], str(n)]
        main()
        wrw_probe([n])
        bwd2_probe([n])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bwd2":
        bwd2_probe([int(a) for a in sys.argv[2:]] or [42496])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wrw":
        wrw_probe([int(a) for a in sys.argv[2:]] or [22016])
        sys.exit(0)
    main()
