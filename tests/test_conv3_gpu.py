"""GPU: forward of the middle conv layers on the bf16 matrix pipe (csrc/conv3.hip, mirl_conv3_fwd) against
`F.relu(conv2d(x, W, b, stride))` (rltime/models/torch/modules/cnn.py:47-49 at the Atari models' layers 2 and 3).
Small-integer operands are exact in one bf16 part and every partial sum is exact in f32: addressing (row -> (n, oh,
ow), the (kh | kw, c) run jumps), tile tails and the epilogue are checked BIT-exactly; real operands within 1e-5 of
the float64 result and no further from it than twice the library's f32 convolution."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (N, C, H, W, F, K, S)
SHAPES = [(3, 32, 20, 20, 64, 4, 2), (5, 64, 9, 9, 64, 3, 1), (1025, 32, 20, 20, 64, 4, 2), (300, 64, 9, 9, 64, 3, 1),
          (2, 8, 11, 13, 20, 2, 1), (7, 16, 6, 6, 4, 3, 3), (1, 4, 5, 4, 8, 4, 1)]


def _operands(n, c, h, w, f, k, gen, integer):
    if integer:
        x = torch.randint(-8, 9, (n, c, h, w), device="cuda", generator=gen).float()
        wt = torch.randint(-4, 5, (f, c, k, k), device="cuda", generator=gen).float()
        b = torch.randint(-100, 101, (f,), device="cuda", generator=gen).float()
    else:
        x = torch.randn(n, c, h, w, device="cuda", generator=gen)
        wt = torch.randn(f, c, k, k, device="cuda", generator=gen) / (c * k * k) ** 0.5
        b = torch.randn(f, device="cuda", generator=gen) * 0.1
    return x.contiguous(memory_format=torch.channels_last), wt.contiguous(memory_format=torch.channels_last), b


@pytest.mark.parametrize("n,c,h,w,f,k,s", SHAPES)
def test_integer_operands_are_bit_exact(n, c, h, w, f, k, s):
    from rltime_amd.models.torch import fused
    gen = torch.Generator(device="cuda").manual_seed(n * 13 + c + k)
    x, wt, b = _operands(n, c, h, w, f, k, gen, True)
    assert fused.conv3_supported(x, wt, (s, s), min_work=0)
    want = F.relu(F.conv2d(x.double(), wt.double(), b.double(), s))
    got = fused.conv3_bias_relu(x, wt, b, (s, s))
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got.double(), want)
    # no ReLU, no bias, weights handed over in NCHW memory (repacked by the wrapper)
    want2 = F.conv2d(x.double(), wt.double(), None, s)
    got2 = fused.conv3_bias_relu(x, wt.contiguous(), None, (s, s), relu=False)
    assert torch.equal(got2.double(), want2)


@pytest.mark.parametrize("n,c,h,w,f,k,s", SHAPES[:5])
def test_real_operands_are_an_f32_convolution(n, c, h, w, f, k, s):
    from rltime_amd.models.torch import fused
    gen = torch.Generator(device="cuda").manual_seed(n + f)
    x, wt, b = _operands(n, c, h, w, f, k, gen, False)
    want = F.relu(F.conv2d(x.double(), wt.double(), b.double(), s))
    got = fused.conv3_bias_relu(x, wt, b, (s, s))
    lib = F.relu(F.conv2d(x, wt, b, s))
    scale = float(want.abs().max())
    err3 = float((got.double() - want).abs().max()) / scale
    errl = float((lib.double() - want).abs().max()) / scale
    assert err3 <= max(2.0 * errl, 2e-6), (err3, errl)      # the library's direct kernels can be unusually exact on tiny shapes
    assert err3 <= 1e-5


def test_unsupported_shapes_keep_the_library_path():
    from rltime_amd.models.torch import fused
    from rltime_amd._lib import lib
    assert lib.mirl_conv3_fwd_supported(32, 64, 4, 4, 2, 20, 20) == 1 and lib.mirl_conv3_fwd_supported(64, 64, 3, 3, 1, 9, 9) == 1
    for c, f, kh, kw, s, h, w in [(3, 64, 4, 4, 2, 20, 20), (32, 128, 4, 4, 2, 20, 20), (32, 62, 4, 4, 2, 20, 20), (4, 32, 3, 3, 1, 9, 9),
                                  (32, 64, 4, 4, 2, 3, 20)]:
        assert lib.mirl_conv3_fwd_supported(c, f, kh, kw, s, h, w) == 0, (c, f, kh, kw, s, h, w)
    x = torch.randn(4, 32, 20, 20, device="cuda")                      # NCHW memory: not taken
    assert not fused.conv3_supported(x, torch.randn(64, 32, 4, 4, device="cuda"), (2, 2), min_work=0)


def test_conv_bias_relu_module_path_uses_it_and_matches(monkeypatch):
    """fused.conv_bias_relu above the work threshold: forward equal to the library path within f32 noise, gradients
    computed on the fused forward's own ReLU mask equal to autograd's."""
    import torch.nn as nn
    from rltime_amd.models.torch import fused
    torch.manual_seed(5)
    conv = nn.Conv2d(32, 64, 4, 2).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(2200, 32, 20, 20, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert fused.conv3_supported(x, conv.weight, conv.stride)
    y = fused.conv_bias_relu(x, conv)
    up = torch.randn_like(y)
    (y * up).sum().backward()
    got = (y.detach(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
    x2 = x.detach().clone().requires_grad_(True)
    pre = F.conv2d(x2, conv.weight, conv.bias, conv.stride)
    dx, dw, db = torch.autograd.grad(pre, (x2, conv.weight, conv.bias), grad_outputs=up * (got[0] > 0))
    for a, b, what in zip(got, (F.relu(pre).detach(), dx, dw, db), ("y", "dx", "dW", "db")):
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)
        assert err <= 1e-4, (what, err)
