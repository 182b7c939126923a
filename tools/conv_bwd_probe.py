"""One launch set of the conv stack's backward kernels at the learner's frame count (for rocprofv3 --pmc / --kernel-trace):
layer-2 / layer-3 data gradients and weight gradients, the input layer's masked weight gradient — bf16-pipe forms and, for
comparison, the f32-pipe / MIOpen forms they replaced."""
import ctypes as C
import sys

import torch

from rltime_amd._lib import lib, check
from rltime_amd.models.torch import fused


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    x2 = cl(torch.randn(n, 32, 20, 20, device="cuda")); g2 = cl(torch.randn(n, 64, 9, 9, device="cuda")); w2 = cl(torch.randn(64, 32, 4, 4, device="cuda") * 0.05)
    x3 = cl(torch.randn(n, 64, 9, 9, device="cuda")); g3 = cl(torch.randn(n, 64, 7, 7, device="cuda")); w3 = cl(torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
    x1 = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device="cuda")
    dy1 = cl(torch.randn(n, 32, 20, 20, device="cuda")); y1 = cl(torch.randn(n, 32, 20, 20, device="cuda").clamp(min=0))
    need = C.c_int64(); check(lib.mirl_conv1_u8_wrw_scratch_floats(C.byref(need)))
    scratch = torch.empty(need.value, device="cuda"); dw1 = torch.empty(32, 4, 8, 8, device="cuda"); db1 = torch.empty(32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())                                                        # noqa: E731
    so, sc, sh, sw = dw1.stride()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def wrw1():
        check(lib.mirl_conv1_u8_wrw_masked(n, 84, 84, p(x1), p(dy1), p(y1), 1 / 255., p(scratch), p(dw1), so, sc, sh, sw, p(db1), st))
    bw = lambda g, x, w, s, m: torch.ops.aten.convolution_backward(g, x, w, None, [s, s], [0, 0], [1, 1], False, [0, 0], 1, m)   # noqa: E731
    for _ in range(reps):
        fused.conv2_bwd_data(g2, w2, x2, 1); fused.conv2_bwd_data(g2, w2, x2, 0)
        fused.conv3_bwd_data(g3, w3, x3); bw(g3, x3, w3, 1, [True, False, False])
        fused.conv_wgrad_b3(g2, x2, w2, (2, 2)); bw(g2, x2, w2, 2, [False, True, False])
        fused.conv_wgrad_b3(g3, x3, w3, (1, 1)); bw(g3, x3, w3, 1, [False, True, False])
        check(lib.mirl_conv1_wrw_bf16_set(1)); wrw1(); check(lib.mirl_conv1_wrw_bf16_set(0)); wrw1(); check(lib.mirl_conv1_wrw_bf16_set(-1))
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
