"""GPU: the (opt-in) backward of conv layers 2-3 as wide split-bf16 GEMMs over the explicit window matrix (csrc/conv_col.hip
mirl_im2col_nhwc / mirl_col2im_nhwc around mirl_gemm3) against autograd's convolution gradients
(rltime/models/torch/modules/cnn.py:43-50).  The two data movements are checked bit-exactly (im2col is a copy; col2im on
small integers has exact partial sums); the composed gradients on small-integer operands bit-exactly against float64,
on real operands within 1e-5 of the float64 gradients and no further from them than twice the library's f32 kernels."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _column_path_on(monkeypatch):
    """The path is opt-in (MIRL_CONV_COL=1: measured slower than MIOpen's implicit GEMMs at the learner's frame counts)."""
    from rltime_amd.models.torch import fused
    monkeypatch.setattr(fused, "_CONV_COL", True)

# (N, C, H, W, F, K, S): the Atari layers 2 and 3 at several frame counts, and odd small shapes
LAYERS = [(16, 32, 20, 20, 64, 4, 2), (16, 64, 9, 9, 64, 3, 1), (304, 32, 20, 20, 64, 4, 2), (400, 64, 9, 9, 64, 3, 1),
          (32, 8, 11, 13, 16, 2, 1), (64, 16, 6, 6, 32, 3, 3), (128, 4, 5, 4, 16, 4, 1)]


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _col_reference(x, k, s):
    """[(n, oy, ox)][(ky, kx, c)] from F.unfold's [(c, ky, kx)][(oy, ox)] blocks."""
    n, c, h, w = x.shape
    oh, ow = (h - k) // s + 1, (w - k) // s + 1
    u = F.unfold(x.contiguous(), k, stride=s).view(n, c, k, k, oh, ow)
    return u.permute(0, 4, 5, 2, 3, 1).reshape(n * oh * ow, k * k * c)


@pytest.mark.parametrize("n,c,h,w,f,k,s", LAYERS)
def test_im2col_is_the_window_matrix(n, c, h, w, f, k, s):
    from rltime_amd.models.torch import fused
    gen = torch.Generator(device="cuda").manual_seed(n + 7 * c)
    x = _cl(torch.randn(n, c, h, w, device="cuda", generator=gen))
    assert torch.equal(fused.im2col_nhwc(x, k, k, s), _col_reference(x, k, s))


@pytest.mark.parametrize("n,c,h,w,f,k,s", LAYERS)
def test_col2im_sums_every_window_once(n, c, h, w, f, k, s):
    from rltime_amd.models.torch import fused
    gen = torch.Generator(device="cuda").manual_seed(n + 11 * c)
    oh, ow = (h - k) // s + 1, (w - k) // s + 1
    dcol = torch.randint(-50, 51, (n * oh * ow, k * k * c), device="cuda", generator=gen).float()
    x_like = _cl(torch.empty(n, c, h, w, device="cuda"))
    # F.fold wants [(c, ky, kx)][(oy, ox)] blocks
    blocks = dcol.view(n, oh, ow, k, k, c).permute(0, 5, 3, 4, 1, 2).reshape(n, c * k * k, oh * ow)
    want = F.fold(blocks.double(), (h, w), k, stride=s)
    got = fused.col2im_nhwc(dcol, x_like, k, k, s)
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got.double(), want)
    mask = _cl(torch.randn(n, c, h, w, device="cuda", generator=gen))
    got = fused.col2im_nhwc(dcol, x_like, k, k, s, relu_mask=mask)
    assert torch.equal(got.double(), want * (mask > 0))


def _grads64(g, x, wt, s):
    return torch.ops.aten.convolution_backward(g.double(), x.double(), wt.double(), None, [s, s], [0, 0], [1, 1], False, [0, 0], 1,
                                               [True, True, False])[:2]


@pytest.mark.parametrize("n,c,h,w,f,k,s", LAYERS)
def test_integer_gradients_are_bit_exact(n, c, h, w, f, k, s):
    from rltime_amd.models.torch import fused
    gen = torch.Generator(device="cuda").manual_seed(3 * n + c)
    x = _cl(torch.randint(-8, 9, (n, c, h, w), device="cuda", generator=gen).float())
    wt = _cl(torch.randint(-4, 5, (f, c, k, k), device="cuda", generator=gen).float())
    oh, ow = (h - k) // s + 1, (w - k) // s + 1
    g = _cl(torch.randint(-3, 4, (n, f, oh, ow), device="cuda", generator=gen).float())
    assert fused.conv_col_supported(x, wt, (s, s), g, min_work=0)
    dx64, dw64 = _grads64(g, x, wt, s)
    dw = fused.conv_wgrad_col(g, x, wt, (s, s))
    dx = fused.conv_dgrad_col(g, x, wt, (s, s))
    assert dw.shape == wt.shape and dw.stride() == wt.stride()
    assert dx.shape == x.shape and dx.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(dw.double(), dw64)
    assert torch.equal(dx.double(), dx64)


@pytest.mark.parametrize("n,c,h,w,f,k,s", LAYERS[:4] + [(5120, 64, 9, 9, 64, 3, 1)])
def test_real_gradients_are_f32_convolution_gradients(n, c, h, w, f, k, s):
    from rltime_amd.models.torch import fused
    gen = torch.Generator(device="cuda").manual_seed(n + f)
    x = _cl(torch.randn(n, c, h, w, device="cuda", generator=gen))
    wt = _cl(torch.randn(f, c, k, k, device="cuda", generator=gen) / (c * k * k) ** 0.5)
    oh, ow = (h - k) // s + 1, (w - k) // s + 1
    g = _cl(torch.randn(n, f, oh, ow, device="cuda", generator=gen))
    dx64, dw64 = _grads64(g, x, wt, s)
    ldx, ldw, _ = torch.ops.aten.convolution_backward(g, x, wt, None, [s, s], [0, 0], [1, 1], False, [0, 0], 1, [True, True, False])
    dw = fused.conv_wgrad_col(g, x, wt, (s, s))
    dx = fused.conv_dgrad_col(g, x, wt, (s, s))
    for got, lib, want, what in ((dw, ldw, dw64, "dW"), (dx, ldx, dx64, "dx")):
        scale = float(want.abs().max())
        e = float((got.double() - want).abs().max()) / scale
        el = float((lib.double() - want).abs().max()) / scale
        assert e <= max(2.0 * el, 2e-6), (what, e, el)
        assert e <= 1e-5, (what, e)


def test_shapes_the_column_path_leaves_to_the_library():
    from rltime_amd.models.torch import fused
    x = _cl(torch.randn(16, 32, 20, 20, device="cuda"))
    wt = _cl(torch.randn(64, 32, 4, 4, device="cuda"))
    g = _cl(torch.randn(16, 64, 9, 9, device="cuda"))
    assert fused.conv_col_supported(x, wt, (2, 2), g, min_work=0)
    assert not fused.conv_col_supported(x, wt, (2, 2), g)                                  # below the work threshold
    assert not fused.conv_col_supported(x.contiguous(), wt, (2, 2), g, min_work=0)         # NCHW memory
    assert not fused.conv_col_supported(x[:15], wt, (2, 2), g[:15], min_work=0)            # 15 * 81 rows: not a multiple of 16
    assert not fused.conv_col_supported(x, wt, (2, 1), g, min_work=0)
    w3 = _cl(torch.randn(60, 32, 4, 4, device="cuda"))
    assert not fused.conv_col_supported(x, w3, (2, 2), _cl(torch.randn(16, 60, 9, 9, device="cuda")), min_work=0)


def test_module_backward_takes_the_column_path_and_matches_autograd(monkeypatch):
    """fused.conv_bias_relu's backward above the work threshold: k_im2col_nhwc / k_col2im_nhwc run, MIOpen's backward
    kernels and conv_mid's four-GEMM data gradient do not, and the gradients equal autograd's on the same ReLU mask."""
    import torch.nn as nn
    from rltime_amd import _lib
    from rltime_amd.models.torch import fused
    torch.manual_seed(9)
    for (cin, cout, k, s, hw, n) in ((32, 64, 4, 2, 20, 2208), (64, 64, 3, 1, 9, 2208)):
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
        assert ran.get("k_im2col_nhwc") == 1 and ran.get("k_col2im_nhwc") == 1 and ran.get("k_gemm3_tn") == 1 and ran.get("k_gemm3_nn") == 1, ran
        assert not ran.get("k_conv2_bwd_data"), ran
        got = (x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
        assert got[1].stride() == conv.weight.stride()
        x2 = x.detach().clone().requires_grad_(True)
        pre = F.conv2d(x2.double(), conv.weight.double(), conv.bias.double(), conv.stride)
        want = torch.autograd.grad(pre, (x2, conv.weight, conv.bias), grad_outputs=(up * (y.detach() > 0)).double())
        for a, b, what in zip(got, want, ("dx", "dW", "db")):
            err = float((a.double() - b.double()).abs().max()) / (float(b.abs().max()) + 1e-12)
            assert err <= 1e-5, (what, err)
        conv.zero_grad(set_to_none=True)
