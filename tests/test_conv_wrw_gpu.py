"""GPU: weight gradient of the middle conv layers on the bf16 matrix pipe (csrc/conv_wrw.hip, mirl_conv_wrw_b3) against
autograd's convolution weight gradient (rltime/models/torch/modules/cnn.py:43-50 at the Atari models' layers 2 and 3).
Small-integer operands are exact in one bf16 part and every partial sum is exact in f32: position / tap addressing, the
padded position list, frames-per-fill tails and the slab reduction are checked BIT-exactly; real operands within 1e-5 of
the float64 gradient and no further from it than twice the library's f32 kernel; reruns are bit-identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (N, C, H, W, F, K, S): layers 2 and 3 at frame counts around the frames-per-fill / grid boundaries, and odd small shapes
SHAPES = [(1, 32, 20, 20, 64, 4, 2), (3, 64, 9, 9, 64, 3, 1), (513, 32, 20, 20, 64, 4, 2), (515, 64, 9, 9, 64, 3, 1),
          (7, 16, 11, 13, 64, 2, 1), (5, 16, 6, 6, 64, 3, 3), (2, 4, 5, 4, 64, 4, 1), (9, 32, 7, 7, 64, 2, 2)]


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _want(g, x, wt, s):
    return torch.ops.aten.convolution_backward(g.double(), x.double(), wt.double(), None, [s, s], [0, 0], [1, 1], False, [0, 0], 1,
                                               [False, True, False])[1]


@pytest.mark.parametrize("n,c,h,w,f,k,s", SHAPES)
def test_integer_operands_are_bit_exact(n, c, h, w, f, k, s):
    from rltime_amd.models.torch import fused
    gen = torch.Generator(device="cuda").manual_seed(5 * n + c)
    x = _cl(torch.randint(-8, 9, (n, c, h, w), device="cuda", generator=gen).float())
    wt = _cl(torch.empty(f, c, k, k, device="cuda"))
    oh, ow = (h - k) // s + 1, (w - k) // s + 1
    g = _cl(torch.randint(-3, 4, (n, f, oh, ow), device="cuda", generator=gen).float())
    assert fused.conv_wrw_supported(x, wt, (s, s), g, min_work=0)
    dw = fused.conv_wgrad_b3(g, x, wt, (s, s))
    assert dw.shape == wt.shape and dw.stride() == wt.stride()
    assert torch.equal(dw.double(), _want(g, x, wt, s))


@pytest.mark.parametrize("n,c,h,w,f,k,s", SHAPES[:4] + [(4096, 32, 20, 20, 64, 4, 2), (4096, 64, 9, 9, 64, 3, 1)])
def test_real_operands_are_an_f32_weight_gradient(n, c, h, w, f, k, s):
    from rltime_amd.models.torch import fused
    gen = torch.Generator(device="cuda").manual_seed(n + f)
    x = _cl(torch.randn(n, c, h, w, device="cuda", generator=gen))
    wt = _cl(torch.empty(f, c, k, k, device="cuda"))
    oh, ow = (h - k) // s + 1, (w - k) // s + 1
    g = _cl(torch.randn(n, f, oh, ow, device="cuda", generator=gen) * (torch.rand(n, f, oh, ow, device="cuda", generator=gen) < 0.5))
    want = _want(g, x, wt, s)
    lib = torch.ops.aten.convolution_backward(g, x, wt, None, [s, s], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    dw = fused.conv_wgrad_b3(g, x, wt, (s, s))
    scale = float(want.abs().max())
    e = float((dw.double() - want).abs().max()) / scale
    el = float((lib.double() - want).abs().max()) / scale
    assert e <= max(2.0 * el, 2e-6), (e, el)
    assert e <= 1e-5
    assert torch.equal(dw, fused.conv_wgrad_b3(g, x, wt, (s, s)))          # fixed partition, fixed order


def test_unsupported_shapes_keep_the_library_path():
    from rltime_amd.models.torch import fused
    from rltime_amd._lib import lib
    assert lib.mirl_conv_wrw_b3_supported(32, 64, 4, 4, 2, 20, 20) == 1 and lib.mirl_conv_wrw_b3_supported(64, 64, 3, 3, 1, 9, 9) == 1
    for c, f, kh, kw, s, h, w in [(32, 32, 4, 4, 2, 20, 20), (3, 64, 4, 4, 2, 20, 20), (64, 64, 4, 4, 1, 9, 9), (4, 64, 3, 3, 1, 9, 9),
                                  (32, 64, 4, 4, 2, 3, 20), (32, 64, 4, 4, 2, 84, 84)]:
        assert lib.mirl_conv_wrw_b3_supported(c, f, kh, kw, s, h, w) == 0, (c, f, kh, kw, s, h, w)
    x = torch.randn(64, 32, 20, 20, device="cuda")                        # NCHW memory: not taken
    g = _cl(torch.randn(64, 64, 9, 9, device="cuda"))
    wt = _cl(torch.randn(64, 32, 4, 4, device="cuda"))
    assert not fused.conv_wrw_supported(x, wt, (2, 2), g, min_work=0)
    assert not fused.conv_wrw_supported(_cl(x), wt, (2, 2), g)             # below the work threshold
    assert fused.conv_wrw_supported(_cl(x), wt, (2, 2), g, min_work=0)


def test_layer_backward_uses_it_and_matches_autograd():
    """fused.conv_bias_relu's backward above the work threshold: k_conv_wrw_b3 runs for both layers, MIOpen's weight-gradient
    kernel does not, and all three gradients equal autograd's (float64) on the same ReLU mask."""
    import torch.nn as nn
    import torch.nn.functional as F
    from rltime_amd import _lib
    from rltime_amd.models.torch import fused
    torch.manual_seed(11)
    for (cin, cout, k, s, hw, n) in ((32, 64, 4, 2, 20, 2208), (64, 64, 3, 1, 9, 2209)):
        conv = nn.Conv2d(cin, cout, k, s).cuda().to(memory_format=torch.channels_last)
        x = _cl(torch.randn(n, cin, hw, hw, device="cuda")).requires_grad_(True)
        y = fused.conv_bias_relu(x, conv)
        up = torch.randn_like(y)
        _lib.check(_lib.lib.mirl_profile_reset())
        _lib.check(_lib.lib.mirl_profile_set(2))
        try:
            (y * up).sum().backward()
            torch.cuda.synchronize()
            ran = {r["name"]: r["calls"] for r in _lib.profile_table()}
        finally:
            _lib.check(_lib.lib.mirl_profile_set(0))
        assert ran.get("k_conv_wrw_b3") == 1 and ran.get("k_conv_wrw_reduce") == 1, ran
        got = (x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
        assert got[1].stride() == conv.weight.stride()
        x2 = x.detach().clone().requires_grad_(True)
        pre = F.conv2d(x2.double(), conv.weight.double(), conv.bias.double(), conv.stride)
        want = torch.autograd.grad(pre, (x2, conv.weight, conv.bias), grad_outputs=(up * (y.detach() > 0)).double())
        for a, b, what in zip(got, want, ("dx", "dW", "db")):
            err = float((a.double() - b.double()).abs().max()) / (float(b.abs().max()) + 1e-12)
            assert err <= (1e-4 if what == "dx" else 1e-5), (what, err)
        conv.zero_grad(set_to_none=True)
