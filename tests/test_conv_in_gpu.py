"""GPU: the one-pass input layer (csrc/conv_in.hip, mirl_conv1_u8_fwd) against the
reference's expression for it, rltime/models/torch/modules/cnn.py:44-49:
relu(conv2d(x.float() * scale, W, b, stride 4)).  Integer-valued weights with
scale 1 make every product and partial sum exact in fp32, so indexing is checked
BIT-exactly; real weights are held to the north-star 1e-4 (same products, other
summation order)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(1, 84, 84), (2, 84, 84), (3, 84, 84), (33, 84, 84), (1025, 84, 84), (7, 36, 36), (5, 44, 52), (4, 12, 16),
          (6, 8, 8), (2, 100, 100)]


def _call(x, w, b, scale, flags=None):
    from rltime_amd._lib import lib, check
    n, _, h, ww = x.shape
    oh, ow = (h - 8) // 4 + 1, (ww - 8) // 4 + 1
    y = torch.full((n, 32, oh, ow), float("nan"), device="cuda").contiguous(memory_format=torch.channels_last)
    wpk = torch.empty(12288, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    so, sc, sh, sw = w.stride()
    if flags is None:
        check(lib.mirl_conv1_u8_fwd(n, h, ww, p(x), p(w), so, sc, sh, sw, p(b), scale, p(wpk), p(y), st), "conv1")
    else:
        check(lib.mirl_conv1_u8_fwd_ex(n, h, ww, p(x), p(w), so, sc, sh, sw, p(b), scale, p(wpk), p(y), flags, st), "conv1_ex")
    return y


@pytest.mark.parametrize("n,h,w", SHAPES)
def test_integer_weights_are_bit_exact(n, h, w):
    g = torch.Generator(device="cuda").manual_seed(n * 1000 + h)
    x = torch.randint(0, 256, (n, 4, h, w), dtype=torch.uint8, device="cuda", generator=g)
    wt = torch.randint(-2, 3, (32, 4, 8, 8), device="cuda", generator=g).float()
    b = torch.randint(-50000, 50000, (32,), device="cuda", generator=g).float()
    want = F.relu(F.conv2d(x.double(), wt.double(), b.double(), 4)).float()
    got = _call(x, wt, b, 1.0)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)
    # weights handed over in NHWC memory (how the channels_last model stores them): same result
    assert torch.equal(_call(x, wt.contiguous(memory_format=torch.channels_last), b, 1.0), want)


@pytest.mark.parametrize("n,h,w", [(3, 84, 84), (1025, 84, 84), (5, 44, 52)])
def test_launch_shapes_agree_bitwise(n, h, w):
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randint(0, 256, (n, 4, h, w), dtype=torch.uint8, device="cuda", generator=g)
    wt = torch.randn(32, 4, 8, 8, device="cuda", generator=g) * 0.05
    b = torch.randn(32, device="cuda", generator=g) * 0.1
    base = _call(x, wt, b, 1.0 / 255.0)
    for fpi in (1, 2):
        for split in (0, 1, 3, 7):
            for cached in (0, 1):
                got = _call(x, wt, b, 1.0 / 255.0, flags=cached | (fpi << 8) | (split << 16))
                assert torch.equal(got, base), (fpi, split, cached)
    # the f32-MFMA kernel (flag bit 5), conversions hoisted in front of its chain or interleaved (bit 2): same sums in
    # the same order; against the default bf16-pipe kernel (exact pixels x three-way split weights) only the order of
    # the 256-term f32 sum differs
    f32k = _call(x, wt, b, 1.0 / 255.0, flags=32 | (2 << 8))
    assert torch.equal(_call(x, wt, b, 1.0 / 255.0, flags=4 | 32 | (2 << 8)), f32k)
    assert float((f32k - base).abs().max()) <= 2e-6 * float(base.abs().max())


@pytest.mark.parametrize("n,h,w", [(2, 84, 84), (257, 84, 84), (2050, 84, 84), (5, 44, 52)])
def test_real_weights_within_tolerance(n, h, w):
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randint(0, 256, (n, 4, h, w), dtype=torch.uint8, device="cuda", generator=g)
    conv = nn.Conv2d(4, 32, 8, 4).cuda()
    with torch.no_grad():
        conv.bias.uniform_(-0.3, 0.3)
    want = F.relu(F.conv2d(x.double() * (1.0 / 255.0), conv.weight.double(), conv.bias.double(), 4))
    got = _call(x, conv.weight.detach(), conv.bias.detach(), 1.0 / 255.0)
    err = float((got.double() - want).abs().max()) / float(want.abs().max())
    assert err <= 1e-5, err            # far inside the 1e-4 bar: 256 fp32 terms


def test_autograd_function_matches_plain_expression():
    from rltime_amd.models.torch.fused import conv_u8_bias_relu, conv_u8_supported
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randint(0, 256, (37, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    conv = nn.Conv2d(4, 32, 8, 4).cuda().to(memory_format=torch.channels_last)
    assert conv_u8_supported(x, conv)
    conv.zero_grad(set_to_none=True)
    y = conv_u8_bias_relu(x, conv, 1.0 / 255.0)
    up = torch.randn(y.shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    (y * up).sum().backward()
    fused = (y.detach(), conv.weight.grad.clone(), conv.bias.grad.clone())
    # the plain expression; its backward gets the FUSED forward's ReLU mask: an output within an ulp of zero may sit
    # on the other side in the two forwards (different f32 summation order), which would move a whole filter's
    # gradient — the mask is compared through `y`, the gradients on the same mask
    pre = conv(x.float() * (1.0 / 255.0))
    dw, db = torch.autograd.grad(pre, (conv.weight, conv.bias), grad_outputs=up * (fused[0] > 0))
    for a, b, what in zip(fused, (F.relu(pre).detach(), dw, db), ("y", "dW", "db")):
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)
        assert err <= 1e-4, (what, err)


def test_unsupported_shapes_are_refused_and_the_module_falls_back():
    from rltime_amd._lib import lib
    from rltime_amd.models.torch.modules import CNN
    assert lib.mirl_conv1_u8_supported(4, 84, 84, 32, 8, 4) == 1
    for c, h, w, f, k, s in [(3, 84, 84, 32, 8, 4), (4, 84, 84, 16, 8, 4), (4, 84, 84, 32, 4, 2), (4, 42, 42, 32, 8, 4),
                             (4, 84, 86, 32, 8, 4), (4, 4, 84, 32, 8, 4), (4, 200, 200, 32, 8, 4)]:
        assert lib.mirl_conv1_u8_supported(c, h, w, f, k, s) == 0, (c, h, w, f, k, s)
    x = torch.zeros((2, 4, 42, 42), dtype=torch.uint8, device="cuda")
    wt, b = torch.zeros(32, 4, 8, 8, device="cuda"), torch.zeros(32, device="cuda")
    from rltime_amd._lib import MirlError
    with pytest.raises(MirlError):
        _call(x, wt, b, 1.0)
    layers = [{"filters": 32, "kernel": 8, "stride": 4}, {"filters": 64, "kernel": 4, "stride": 2}]
    torch.manual_seed(0)
    direct = CNN((4, 84, 84), layers, channels_last=True, direct_u8=True).cuda()
    plain = CNN((4, 84, 84), layers, channels_last=True, direct_u8=False).cuda()
    plain.load_state_dict(direct.state_dict())
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randint(0, 256, (9, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    assert direct.prepare_input(x) is None and plain.prepare_input(x) is not None
    a, bb = direct(x), plain(x)
    assert float((a - bb).abs().max()) <= 1e-5 * float(bb.abs().max())
    # a frame shape the kernel does not cover: both modules take the generic path
    small = CNN((4, 42, 42), layers, channels_last=True, direct_u8=True).cuda()
    xs = torch.randint(0, 256, (3, 4, 42, 42), dtype=torch.uint8, device="cuda", generator=g)
    assert small.prepare_input(xs) is not None
    assert small(xs).shape == (3, 64, 3, 3)


def _wrw(x, g, scale, like, flags=None):
    from rltime_amd._lib import lib, check
    need = C.c_int64()
    check(lib.mirl_conv1_u8_wrw_scratch_floats(C.byref(need)))
    scratch = torch.empty(need.value, device="cuda")
    dw = torch.full_like(like, float("nan"))
    n, _, h, w = x.shape
    so, sc, sh, sw = dw.stride()
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if flags is None:
        check(lib.mirl_conv1_u8_wrw(n, h, w, p(x), p(g), scale, p(scratch), p(dw), so, sc, sh, sw, st), "conv1_wrw")
    else:
        check(lib.mirl_conv1_u8_wrw_ex(n, h, w, p(x), p(g), scale, p(scratch), p(dw), so, sc, sh, sw, flags, st), "conv1_wrw_ex")
    return dw


@pytest.fixture(params=[1, 0], ids=["wrw-bf16-pipe", "wrw-f32-pipe"])
def wrw_pipe(request):
    """Both matrix pipes of the weight gradient (k_conv1_u8_wrw_b3: exact three-way split of g on the bf16 pipe, the
    default; k_conv1_u8_wrw: f32 MFMA) through the same entry points."""
    from rltime_amd._lib import lib, check
    check(lib.mirl_conv1_wrw_bf16_set(request.param))
    yield request.param
    check(lib.mirl_conv1_wrw_bf16_set(-1))


def _wrw_reference(x, g, scale):
    # d/dW of sum(conv2d(x*scale, W) * g) in float64
    w = torch.zeros(32, 4, 8, 8, dtype=torch.float64, device="cuda", requires_grad=True)
    (F.conv2d(x.double() * scale, w, None, 4) * g.double()).sum().backward()
    return w.grad


@pytest.mark.parametrize("n,h,w", [(1, 84, 84), (2, 84, 84), (3, 36, 36), (2, 44, 52), (4, 12, 16), (3, 8, 8), (1, 100, 100)])
def test_weight_gradient_integer_case_is_bit_exact(n, h, w, wrw_pipe):
    g_ = torch.Generator(device="cuda").manual_seed(n * 100 + w)
    x = torch.randint(0, 256, (n, 4, h, w), dtype=torch.uint8, device="cuda", generator=g_)
    oh, ow = (h - 8) // 4 + 1, (w - 8) // 4 + 1
    g = torch.randint(-3, 4, (n, 32, oh, ow), device="cuda", generator=g_).float().contiguous(memory_format=torch.channels_last)
    want = _wrw_reference(x, g, 1.0).float()          # |sums| <= n*oh*ow*255*3 < 2^24: exact in fp32 in any order
    for like in (torch.empty(32, 4, 8, 8, device="cuda"),
                 torch.empty(32, 4, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)):
        got = _wrw(x, g, 1.0, like)
        assert got.stride() == like.stride() and torch.equal(got, want)


@pytest.mark.parametrize("n,h,w", [(5, 84, 84), (1025, 84, 84), (2051, 84, 84), (1030, 44, 52)])
def test_weight_gradient_within_tolerance_and_reproducible(n, h, w, wrw_pipe):
    g_ = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randint(0, 256, (n, 4, h, w), dtype=torch.uint8, device="cuda", generator=g_)
    oh, ow = (h - 8) // 4 + 1, (w - 8) // 4 + 1
    g = (torch.randn(n, 32, oh, ow, device="cuda", generator=g_)
         * (torch.rand(n, 32, oh, ow, device="cuda", generator=g_) < 0.5)).contiguous(memory_format=torch.channels_last)
    like = torch.empty(32, 4, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)
    a, b = _wrw(x, g, 1.0 / 255.0, like), _wrw(x, g, 1.0 / 255.0, like)
    assert torch.equal(a, b)                          # fixed partition and order: bit-identical reruns
    # both instruction schedules (conversions hoisted / interleaved) do the same sums in the same order
    assert torch.equal(_wrw(x, g, 1.0 / 255.0, like, flags=0), a) and torch.equal(_wrw(x, g, 1.0 / 255.0, like, flags=1), a)
    want = _wrw_reference(x, g, 1.0 / 255.0)
    err = float((a.double() - want).abs().max()) / float(want.abs().max())
    assert err <= 1e-4, err
    if wrw_pipe == 1:       # the split products are held to the f32 pipe's own distance from float64 (flags bit 1: that kernel)
        f32 = _wrw(x, g, 1.0 / 255.0, like, flags=2)
        err32 = float((f32.double() - want).abs().max()) / float(want.abs().max())
        assert err <= max(2.0 * err32, 2e-6), (err, err32)


@pytest.mark.parametrize("n,h,w", [(2, 84, 84), (3, 36, 36), (1, 8, 8), (1027, 84, 84), (1030, 44, 52)])
def test_masked_weight_gradient_equals_mask_pass_plus_weight_gradient(n, h, w, wrw_pipe):
    """mirl_conv1_u8_wrw_masked: the layer's ReLU mask (y > 0, cnn.py:47-49) applied while dy is loaded and the bias
    gradient from the same pass, against the two-launch form (k_relu_bwd_bias_rows, then mirl_conv1_u8_wrw on its
    output): the weight gradient is BIT-identical (the same masked values enter the same sums in the same order), the
    bias gradient is the masked gradient's column sums (bit-exact on integer-valued dy, 1e-5 of scale on real dy) and
    both are bit-reproducible."""
    from rltime_amd._lib import lib, check
    from rltime_amd.models.torch.fused import relu_bwd_bias_rows
    g_ = torch.Generator(device="cuda").manual_seed(7 * n + w)
    x = torch.randint(0, 256, (n, 4, h, w), dtype=torch.uint8, device="cuda", generator=g_)
    oh, ow = (h - 8) // 4 + 1, (w - 8) // 4 + 1
    y = (torch.randn(n, 32, oh, ow, device="cuda", generator=g_).clamp(min=0)).contiguous(memory_format=torch.channels_last)
    like = torch.empty(32, 4, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)
    p = lambda t: C.c_void_p(t.data_ptr())                                  # noqa: E731
    need = C.c_int64()
    check(lib.mirl_conv1_u8_wrw_scratch_floats(C.byref(need)))

    def masked(dy, scale):
        scratch = torch.empty(need.value, device="cuda")
        dw, db = torch.full_like(like, float("nan")), torch.full((32,), float("nan"), device="cuda")
        so, sc, sh, sw = dw.stride()
        check(lib.mirl_conv1_u8_wrw_masked(n, h, w, p(x), p(dy), p(y), scale, p(scratch), p(dw), so, sc, sh, sw, p(db),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), "conv1_wrw_masked")
        return dw, db
    for integer in (True, False):
        dy = (torch.randint(-3, 4, (n, 32, oh, ow), device="cuda", generator=g_).float() if integer
              else torch.randn(n, 32, oh, ow, device="cuda", generator=g_)).contiguous(memory_format=torch.channels_last)
        scale = 1.0 if integer else 1.0 / 255.0
        g, db_ref = relu_bwd_bias_rows(dy, y, 32)
        dw_ref = _wrw(x, g, scale, like)
        dw, db = masked(dy, scale)
        assert torch.equal(dw, dw_ref)
        want_db = (dy.double() * (y > 0)).sum((0, 2, 3))
        if integer:
            assert torch.equal(db.double(), want_db) and torch.equal(db, db_ref)
        else:
            assert float((db.double() - want_db).abs().max()) <= 1e-5 * max(float(want_db.abs().max()), 1.0)
        dw2, db2 = masked(dy, scale)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
