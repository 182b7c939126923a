"""GPU: the second conv layer's data gradient on either matrix pipe (csrc/conv_mid.hip, mirl_conv2_bwd_data_ex:
f32 MFMA, or bf16 MFMA with the exact three-way split = the default) against what autograd derives for conv2d(x, W, stride 2)
(rltime/models/torch/modules/cnn.py:47-49 at Conv2d(32 -> 64, k 4, s 2)).  Integer-
valued operands make every partial sum exact in fp32: indexing is checked bit-exactly;
real operands to the north-star 1e-4."""
import ctypes as C

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


PIPES = [0, 1]      # 0: f32 MFMA, 1: bf16 MFMA, three-way split


def _call(g, w, pipe):
    from rltime_amd._lib import lib, check
    n, _, oh, ow = g.shape
    dx = torch.full((n, 32, 2 * oh + 2, 2 * ow + 2), float("nan"), device="cuda").contiguous(memory_format=torch.channels_last)
    floats = C.c_int64()
    check(lib.mirl_conv2_bwd_data_wpk_floats(C.byref(floats)))
    assert floats.value >= 32768
    wpk = torch.empty(floats.value, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    so, sc, sh, sw = w.stride()
    assert lib.mirl_conv2_bwd_data_ex(n, oh, ow, p(g), p(w), so, sc, sh, sw, p(wpk), 1024, p(dx), pipe, None) != 0      # scratch too small: refused
    check(lib.mirl_conv2_bwd_data_ex(n, oh, ow, p(g), p(w), so, sc, sh, sw, p(wpk), wpk.numel(), p(dx), pipe,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "conv2_bwd_data_ex")
    return dx


def _reference(g, w):
    n, _, oh, ow = g.shape
    x = torch.zeros(n, 32, 2 * oh + 2, 2 * ow + 2, dtype=torch.float64, device="cuda", requires_grad=True)
    (F.conv2d(x, w.double(), None, 2) * g.double()).sum().backward()
    return x.grad


@pytest.mark.parametrize("pipe", PIPES)
@pytest.mark.parametrize("n,oh,ow", [(1, 9, 9), (2, 9, 9), (3, 9, 9), (1025, 9, 9), (2, 1, 1), (3, 4, 7), (5, 12, 3), (1027, 5, 6)])
def test_integer_operands_are_bit_exact(n, oh, ow, pipe):
    gen = torch.Generator(device="cuda").manual_seed(n * 31 + oh)
    g = torch.randint(-3, 4, (n, 64, oh, ow), device="cuda", generator=gen).float().contiguous(memory_format=torch.channels_last)
    w = torch.randint(-2, 3, (64, 32, 4, 4), device="cuda", generator=gen).float()
    want = _reference(g, w).float()
    got = _call(g, w, pipe)
    assert got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got, want)
    assert torch.equal(_call(g, w.contiguous(memory_format=torch.channels_last), pipe), want)


@pytest.mark.parametrize("pipe", PIPES)
@pytest.mark.parametrize("n,oh,ow", [(4, 9, 9), (1031, 9, 9)])
def test_real_operands_within_tolerance(n, oh, ow, pipe):
    gen = torch.Generator(device="cuda").manual_seed(n)
    g = torch.randn(n, 64, oh, ow, device="cuda", generator=gen).contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, 32, 4, 4, device="cuda", generator=gen) * 0.05
    want = _reference(g, w)
    got = _call(g, w, pipe)
    err = float((got.double() - want).abs().max()) / float(want.abs().max())
    assert err <= 1e-5
    if pipe == 1:          # the split products are held to the f32 pipe's own distance from float64
        err32 = float((_call(g, w, 0).double() - want).abs().max()) / float(want.abs().max())
        assert err <= max(2.0 * err32, 2e-6), (err, err32)


def test_layer_backward_uses_it_and_matches_the_library_path(monkeypatch):
    import rltime_amd.models.torch.fused as fused
    from rltime_amd._lib import lib
    assert lib.mirl_conv2_bwd_data_supported(32, 64, 4, 2, 20, 20, 9, 9) == 1
    for args in [(16, 64, 4, 2, 20, 20, 9, 9), (32, 32, 4, 2, 20, 20, 9, 9), (32, 64, 3, 1, 20, 20, 9, 9),
                 (32, 64, 4, 2, 21, 20, 9, 9), (32, 64, 4, 2, 20, 20, 9, 8)]:
        assert lib.mirl_conv2_bwd_data_supported(*args) == 0, args
    torch.manual_seed(0)
    conv = nn.Conv2d(32, 64, 4, 2).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(37, 32, 20, 20, device="cuda").contiguous(memory_format=torch.channels_last)
    up = None
    res = []
    assert fused._CONV2_BWD_PIPE == 1                       # the split-bf16 kernel is the default
    for on in (True, False):
        monkeypatch.setattr(fused, "_CONV2_BWD", on)
        conv.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = fused.conv_bias_relu(xi, conv)
        up = torch.randn_like(y) if up is None else up
        (y * up).sum().backward()
        res.append((xi.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()))
    for a, b, what in zip(res[0], res[1], ("dx", "dW", "db")):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()), what
    # a 21-row input leaves a forward row uncovered: the library path must take it
    x2 = torch.randn(3, 32, 21, 20, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    monkeypatch.setattr(fused, "_CONV2_BWD", True)
    fused.conv_bias_relu(x2, conv).sum().backward()
    assert x2.grad is not None and x2.grad.shape == x2.shape


# ---- the third conv layer's data gradient (64 -> 64, k 3, s 1) on the bf16 pipe --------------------------------------

def _call3(g, w):
    from rltime_amd._lib import lib, check
    n, _, oh, ow = g.shape
    dx = torch.full((n, 64, oh + 2, ow + 2), float("nan"), device="cuda").contiguous(memory_format=torch.channels_last)
    floats = C.c_int64()
    check(lib.mirl_conv3_bwd_data_wpk_floats(C.byref(floats)))
    wpk = torch.empty(floats.value, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    so, sc, sh, sw = w.stride()
    assert lib.mirl_conv3_bwd_data(n, oh, ow, p(g), p(w), so, sc, sh, sw, p(wpk), 1024, p(dx), None) != 0               # scratch too small: refused
    check(lib.mirl_conv3_bwd_data(n, oh, ow, p(g), p(w), so, sc, sh, sw, p(wpk), wpk.numel(), p(dx),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "conv3_bwd_data")
    return dx


def _reference3(g, w):
    n, _, oh, ow = g.shape
    x = torch.zeros(n, 64, oh + 2, ow + 2, dtype=torch.float64, device="cuda", requires_grad=True)
    (F.conv2d(x, w.double(), None, 1) * g.double()).sum().backward()
    return x.grad


@pytest.mark.parametrize("n,oh,ow", [(1, 7, 7), (2, 7, 7), (3, 7, 7), (1025, 7, 7), (2, 1, 1), (3, 4, 6), (5, 9, 2), (1027, 5, 3), (700, 10, 10)])
def test_layer3_integer_operands_are_bit_exact(n, oh, ow):
    gen = torch.Generator(device="cuda").manual_seed(n * 17 + oh)
    g = torch.randint(-3, 4, (n, 64, oh, ow), device="cuda", generator=gen).float().contiguous(memory_format=torch.channels_last)
    w = torch.randint(-2, 3, (64, 64, 3, 3), device="cuda", generator=gen).float()
    want = _reference3(g, w).float()
    got = _call3(g, w)
    assert got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got, want)
    assert torch.equal(_call3(g, w.contiguous(memory_format=torch.channels_last)), want)


@pytest.mark.parametrize("n,oh,ow", [(4, 7, 7), (1031, 7, 7)])
def test_layer3_real_operands_within_tolerance(n, oh, ow):
    gen = torch.Generator(device="cuda").manual_seed(n)
    g = torch.randn(n, 64, oh, ow, device="cuda", generator=gen).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, device="cuda", generator=gen) * 0.05).contiguous(memory_format=torch.channels_last)
    want = _reference3(g, w)
    got = _call3(g, w)
    err = float((got.double() - want).abs().max()) / float(want.abs().max())
    x = torch.empty(n, 64, oh + 2, ow + 2, device="cuda").contiguous(memory_format=torch.channels_last)
    lib_dx = torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    err_lib = float((lib_dx.double() - want).abs().max()) / float(want.abs().max())
    assert err <= 1e-5 and err <= max(2.0 * err_lib, 2e-6), (err, err_lib)
    assert torch.equal(got, _call3(g, w))


def test_layer3_backward_uses_it(monkeypatch):
    import rltime_amd.models.torch.fused as fused
    from rltime_amd import _lib
    from rltime_amd._lib import lib
    assert lib.mirl_conv3_bwd_data_supported(64, 64, 3, 1, 9, 9, 7, 7) == 1
    for args in [(32, 64, 3, 1, 9, 9, 7, 7), (64, 32, 3, 1, 9, 9, 7, 7), (64, 64, 4, 1, 9, 9, 6, 6), (64, 64, 3, 2, 9, 9, 4, 4),
                 (64, 64, 3, 1, 10, 9, 7, 7), (64, 64, 3, 1, 20, 20, 18, 18)]:
        assert lib.mirl_conv3_bwd_data_supported(*args) == 0, args
    torch.manual_seed(1)
    conv = nn.Conv2d(64, 64, 3, 1).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(41, 64, 9, 9, device="cuda").contiguous(memory_format=torch.channels_last)
    res = []
    up = None
    for on in (True, False):
        monkeypatch.setattr(fused, "_CONV3_BWD", on)
        conv.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = fused.conv_bias_relu(xi, conv)
        up = torch.randn_like(y) if up is None else up
        _lib.check(lib.mirl_profile_reset())
        _lib.check(lib.mirl_profile_set(2))
        try:
            (y * up).sum().backward()
            torch.cuda.synchronize()
            ran = {r["name"]: r["calls"] for r in _lib.profile_table()}
        finally:
            _lib.check(lib.mirl_profile_set(0))
        assert bool(ran.get("k_conv3_bwd_data_b3")) == on, ran
        res.append(xi.grad.clone())
    assert float((res[0] - res[1]).abs().max()) <= 1e-4 * float(res[1].abs().max())
