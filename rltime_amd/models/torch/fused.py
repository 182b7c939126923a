"""Fused network glue on the GPU (csrc/nnops.hip + hipBLASLt epilogues).

The contractions stay on MIOpen / hipBLASLt; these autograd functions replace
the streaming PyTorch ops around them — each of which is a full HBM pass over a
multi-GB activation at the benchmark shapes — by single-pass HIP kernels, with
the backward spelled out (the same math autograd derives for the plain
expressions; parity tests in tests/test_fused_gpu.py compare against those):

  linear_relu        relu(x @ W^T + b): ReLU in the GEMM epilogue; backward = one
                     mask + bias-gradient pass, two GEMMs
  conv_bias_relu     relu(conv2d(x, W) + b) (cnn.py:47-49): bias-less MIOpen conv,
                     one in-place bias+ReLU pass; backward = one mask + bias-gradient
                     pass, MIOpen data / weight gradients
  conv_u8_bias_relu  the INPUT layer, relu(conv2d(x_u8 * scale, W) + b) (cnn.py:44-49),
                     straight from the replay's uint8 frames on the f32 MFMA pipe
                     (csrc/conv_in.hip): no converted copy of the frames, no separate
                     bias / ReLU pass; backward = mask + bias-gradient pass and MIOpen's
                     weight gradient on a conversion made there
  cos_embed          IQN cosine features (iqn.py:78-81) in one kernel
  quantile_product   x[m] * relu(phi @ Wq^T + bq)[m, n] (iqn.py:82-102) with a
                     backward that never materialises g*x / g*emb / the mask
"""
import ctypes as C
import os

import torch
import torch.nn.functional as F


from . import gemm3

_TAIL_WGRAD = os.environ.get("MIRL_TAIL_WGRAD", "1") != "0"
_QP_EPILOGUE = os.environ.get("MIRL_QP_EPILOGUE", "1") != "0"
_CONV3 = os.environ.get("MIRL_CONV3", "1") != "0"
# multiply-adds below which a conv layer's forward stays on MIOpen; MIRL_CONV3_MIN_WORK=0 forces every supported shape
# (512 frames: 35 us against 70 / 54 for MIOpen + the bias / ReLU pass, 256 frames: 33 against 54 / 41 — tools/conv3_probe.py)
_CONV3_MIN_WORK = int(os.environ.get("MIRL_CONV3_MIN_WORK", "400000000"))


def _lib():
    from rltime_amd import _lib as L
    return L


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pow2_quads(c):
    q = c // 4
    return c % 4 == 0 and 1 <= q <= 256 and 256 % q == 0


def _fusable(*tensors):
    return all(t.is_cuda and t.dtype == torch.float32 for t in tensors) and not torch.is_autocast_enabled()


def bias_relu_rows_(y, bias, channels):
    """y viewed as (rows, channels) row-major <- relu(y + bias), in place."""
    L = _lib()
    L.check(L.lib.mirl_bias_relu_rows(y.numel() // channels, channels, _p(y), _p(bias), _stream()), "mirl_bias_relu_rows")
    return y


def relu_bwd_bias_rows(dy, y, channels):
    """-> (g = dy * (y > 0), db = column sums of g) for row-major (rows, channels) views."""
    L = _lib()
    rows = y.numel() // channels
    blocks = C.c_int32()
    L.check(L.lib.mirl_colsum_blocks(rows, channels, C.byref(blocks)))
    g = torch.empty_like(y)
    db = torch.empty(channels, dtype=torch.float32, device=y.device)
    partial = torch.empty((blocks.value, channels), dtype=torch.float32, device=y.device)
    L.check(L.lib.mirl_relu_bwd_bias_rows(rows, channels, _p(dy), _p(y), _p(g), _p(db), _p(partial), blocks.value,
                                          _stream()), "mirl_relu_bwd_bias_rows")
    return g, db


class _LinearReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        out = gemm3.linear_fwd(x, weight, bias, relu=True)
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad):
        x, weight, out = ctx.saved_tensors
        n = out.shape[1]
        if _pow2_quads(n) and out.is_contiguous():
            g, db = relu_bwd_bias_rows(grad.contiguous(), out, n)
        else:
            g = torch.ops.aten.threshold_backward(grad.contiguous(), out, 0.0)
            db = g.sum(0) if ctx.needs_input_grad[2] else None
        dx = gemm3.grad_input(g, weight) if ctx.needs_input_grad[0] else None
        dw = gemm3.grad_weight(g, x) if ctx.needs_input_grad[1] else None
        return dx, dw, (db if ctx.needs_input_grad[2] else None)


def linear_relu(x, weight, bias):
    """relu(F.linear(x, weight, bias)) for 2-D x."""
    if x.dim() == 2 and x.dtype == weight.dtype and _fusable(x, weight):
        return _LinearReLU.apply(x, weight, bias)
    return F.relu(F.linear(x, weight, bias))


def conv3_supported(x, weight, stride, min_work=None):
    """Does the split-bf16 implicit GEMM (csrc/conv3.hip) take this NHWC conv forward?"""
    if not (_CONV3 and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0
            and stride[0] == stride[1]):
        return False
    n, c, h, w = x.shape
    f, _, kh, kw = weight.shape
    s = int(stride[0])
    if h < kh or w < kw:
        return False
    work = n * ((h - kh) // s + 1) * ((w - kw) // s + 1) * f * c * kh * kw
    return work >= (_CONV3_MIN_WORK if min_work is None else min_work) and bool(_lib().lib.mirl_conv3_fwd_supported(c, f, kh, kw, s, h, w))


def conv3_bias_relu(x, weight, bias, stride, relu=True):
    """relu(conv2d(x, weight, bias, stride)) for NHWC x in ONE launch; returns the NHWC (channels_last) result."""
    L = _lib()
    n, c, h, w = x.shape
    f, _, kh, kw = weight.shape
    s = int(stride[0])
    wk = weight.permute(0, 2, 3, 1)                   # (F, KH, KW, C): a view for channels_last weights
    if not wk.is_contiguous():
        wk = wk.contiguous()
    y = torch.empty((n, f, (h - kh) // s + 1, (w - kw) // s + 1), dtype=torch.float32, device=x.device,
                    memory_format=torch.channels_last)
    b = bias.contiguous() if bias is not None else None
    L.check(L.lib.mirl_conv3_fwd(n, h, w, c, f, kh, kw, s, _p(x), _p(wk), _p(b) if b is not None else None,
                                 1 if relu else 0, _p(y), _stream()), "mirl_conv3_fwd")
    return y


# no-grad conv layers 2-3 below the implicit GEMM's work threshold (the acting batch of the non-recurrent policies, whose
# actor runs the generic graph): csrc/actnet.hip's kernels, one launch with bias + ReLU instead of MIOpen's fill + implicit
# GEMM + the bias pass (32 frames: ~10 us against 25 / 19).  0: the library.
_ACT_CONV = os.environ.get("MIRL_ACT_CONV", "1") != "0"


def act_conv_layer(x, weight, bias, stride):
    """2 / 3 when csrc/actnet.hip's conv kernel of that layer takes this NHWC forward as stored, else 0."""
    if not (_ACT_CONV and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and weight.dtype == torch.float32 and bias is not None
            and x.is_contiguous(memory_format=torch.channels_last) and weight.is_contiguous(memory_format=torch.channels_last)
            and stride[0] == stride[1] and weight.shape[2] == weight.shape[3] and bias.is_contiguous()
            and x.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0 and x.shape[0] > 0):
        return 0
    f, c, k, _ = weight.shape
    layer = 2 if (c, k, int(stride[0])) == (32, 4, 2) else 3 if (c, k, int(stride[0])) == (64, 3, 1) else 0
    if layer and _lib().lib.mirl_act_conv_supported(layer, c, f, k, int(stride[0]), x.shape[2], x.shape[3]):
        return layer
    return 0


def act_conv_bias_relu(layer, x, weight, bias, stride):
    L = _lib()
    n, c, h, w = x.shape
    f, _, k, _ = weight.shape
    s = int(stride[0])
    ho, wo = (h - k) // s + 1, (w - k) // s + 1
    y = torch.empty((n, f, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    # an NHWC weight's memory IS (Co, kh, kw, Ci): the kernel's tap-major operand without a copy
    L.check(L.lib.mirl_act_conv_fwd(layer, n, h, w, _p(x), _p(weight), _p(bias), _p(y), ho * wo * f, _stream()), "mirl_act_conv_fwd")
    return y


class _ConvBiasReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, track=True):
        # track = the caller's torch.is_grad_enabled(): ctx.needs_input_grad mirrors requires_grad even under no_grad
        layer = 0
        if conv3_supported(x, weight, stride):
            y = conv3_bias_relu(x, weight, bias, stride)    # implicit GEMM on the bf16 pipe, bias + ReLU in its epilogue
        elif not (track and any(ctx.needs_input_grad)) and (layer := act_conv_layer(x, weight, bias, stride)):
            y = act_conv_bias_relu(layer, x, weight, bias, stride)
        else:
            y = F.conv2d(x, weight, None, stride)
            if not y.is_contiguous(memory_format=torch.channels_last):
                y = y.contiguous(memory_format=torch.channels_last)
            bias_relu_rows_(y, bias, y.shape[1])          # NHWC memory: (N*H*W, C) rows
        ctx.stride = stride
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, grad):
        x, weight, y = ctx.saved_tensors
        c = y.shape[1]
        grad = grad.contiguous(memory_format=torch.channels_last)
        if _pow2_quads(c):
            g, db = relu_bwd_bias_rows(grad, y, c)     # same NHWC strides as y (empty_like preserves them)
        else:
            g = torch.ops.aten.threshold_backward(grad, y, 0.0)
            db = g.sum((0, 2, 3))
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = dw = None
        if need_w and conv_wrw_supported(x, weight, ctx.stride, g):
            dw = conv_wgrad_b3(g, x, weight, ctx.stride)      # split-bf16 implicit GEMM, reduction over positions (csrc/conv_wrw.hip)
            need_w = False
        if need_x and _CONV2_BWD and g.shape[0] and conv2_bwd_data_supported(x, weight, ctx.stride, g):
            dx = conv2_bwd_data(g, weight, x)          # f32 MFMA, four parity-class GEMMs (csrc/conv_mid.hip)
            need_x = False
        if need_x and _CONV3_BWD and g.shape[0] and conv3_bwd_data_supported(x, weight, ctx.stride, g):
            dx = conv3_bwd_data(g, weight, x)          # the same on the bf16 pipe for the stride-1 3 x 3 layer
            need_x = False
        lib_dx, lib_dw, _ = torch.ops.aten.convolution_backward(
            g, x, weight, None, list(ctx.stride), [0, 0], [1, 1], False, [0, 0], 1, [need_x, need_w, False]) \
            if (need_x or need_w) else (None, None, None)
        return (dx if dx is not None else lib_dx), (dw if dw is not None else lib_dw), \
            (db if ctx.needs_input_grad[2] else None), None, None


_CONV2_BWD = os.environ.get("MIRL_CONV2_BWD", "1") != "0"   # 0: MIOpen data gradient for the second conv layer
# weight gradient of conv layers 2-3 on the bf16 pipe (csrc/conv_wrw.hip); 0: MIOpen.  Frame counts whose multiply-adds stay
# below MIRL_CONV_WRW_MIN_WORK keep the library either way (launch-bound shapes)
_CONV_WRW = os.environ.get("MIRL_CONV_WRW", "1") != "0"
_CONV_WRW_MIN_WORK = int(os.environ.get("MIRL_CONV_WRW_MIN_WORK", str(1 << 31)))
_wrw_scratch = {}


def conv_wrw_supported(x, weight, stride, g, min_work=None):
    """Does csrc/conv_wrw.hip take this NHWC conv's weight gradient?"""
    if not (_CONV_WRW and x.is_cuda and x.dtype == torch.float32 and g.dtype == torch.float32 and weight.dtype == torch.float32
            and x.dim() == 4 and stride[0] == stride[1] and g.shape[0] > 0
            and x.is_contiguous(memory_format=torch.channels_last) and g.is_contiguous(memory_format=torch.channels_last)
            and x.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0 and not torch.is_autocast_enabled()):
        return False
    n, c, h, w = x.shape
    f, _, kh, kw = weight.shape
    work = g.shape[0] * g.shape[2] * g.shape[3] * f * c * kh * kw
    if work < (_CONV_WRW_MIN_WORK if min_work is None else min_work):
        return False
    return bool(_lib().lib.mirl_conv_wrw_b3_supported(c, f, kh, kw, int(stride[0]), h, w))


def conv_wgrad_b3(g, x, weight, stride):
    """d loss / d weight of conv2d(x, weight, stride) in the weight's channels_last memory format."""
    L = _lib()
    n, c, h, w = x.shape
    f, _, kh, kw = weight.shape
    key = (x.device.index, torch.cuda.current_stream().cuda_stream, c, f, kh, kw)
    scratch = _wrw_scratch.get(key)
    if scratch is None:
        nbytes = C.c_int64()
        L.check(L.lib.mirl_conv_wrw_b3_scratch_bytes(c, f, kh, kw, C.byref(nbytes)), "mirl_conv_wrw_b3_scratch_bytes")
        scratch = _wrw_scratch[key] = torch.empty(nbytes.value, dtype=torch.uint8, device=x.device)
    dw = torch.empty((f, kh, kw, c), dtype=torch.float32, device=x.device)
    L.check(L.lib.mirl_conv_wrw_b3(n, h, w, c, f, kh, kw, int(stride[0]), _p(x), _p(g), _p(scratch), scratch.numel(), _p(dw),
                                   _stream()), "mirl_conv_wrw_b3")
    return dw.permute(0, 3, 1, 2)                     # logical (F, C, KH, KW) with channels_last strides
def conv2_bwd_data_supported(x, weight, stride, g):
    if not (g.is_contiguous(memory_format=torch.channels_last) and g.data_ptr() % 16 == 0
            and tuple(stride) == (2, 2) and weight.shape[2] == weight.shape[3]):
        return False
    return bool(_lib().lib.mirl_conv2_bwd_data_supported(weight.shape[1], weight.shape[0], weight.shape[2], 2, x.shape[2],
                                                         x.shape[3], g.shape[2], g.shape[3]))


_CONV3_BWD = os.environ.get("MIRL_CONV3_BWD", "1") != "0"   # 0: MIOpen data gradient for the third conv layer
_c3_wpk_floats = None


def conv3_bwd_data_supported(x, weight, stride, g):
    if not (g.is_contiguous(memory_format=torch.channels_last) and g.data_ptr() % 16 == 0 and g.dtype == torch.float32
            and tuple(stride) == (1, 1) and weight.shape[2] == weight.shape[3] and not torch.is_autocast_enabled()):
        return False
    return bool(_lib().lib.mirl_conv3_bwd_data_supported(weight.shape[1], weight.shape[0], weight.shape[2], 1, x.shape[2],
                                                         x.shape[3], g.shape[2], g.shape[3]))


def conv3_bwd_data(g, weight, x_like):
    """d loss / d input of conv2d(x, weight, stride 1) for the (64 -> 64, k 3) layer; g is NHWC."""
    global _c3_wpk_floats
    L = _lib()
    if _c3_wpk_floats is None:
        n = C.c_int64()
        L.check(L.lib.mirl_conv3_bwd_data_wpk_floats(C.byref(n)), "mirl_conv3_bwd_data_wpk_floats")
        _c3_wpk_floats = n.value
    dx = torch.empty_like(x_like, memory_format=torch.channels_last)
    wpk = torch.empty(_c3_wpk_floats, dtype=torch.float32, device=g.device)
    so, sc, sh, sw = weight.stride()
    L.check(L.lib.mirl_conv3_bwd_data(g.shape[0], g.shape[2], g.shape[3], _p(g), _p(weight), so, sc, sh, sw, _p(wpk), wpk.numel(),
                                      _p(dx), _stream()), "mirl_conv3_bwd_data")
    return dx


# which matrix pipe the layer-2 data gradient runs on: "bf16" = exact three-way split, f32 results (default), "f32" = f32 MFMA
_CONV2_BWD_PIPE = 0 if os.environ.get("MIRL_CONV2_BWD_PIPE", "bf16") == "f32" else 1
_c2_wpk_floats = None


def conv2_bwd_data(g, weight, x_like, pipe=None):
    """d loss / d input of conv2d(x, weight, stride 2) for the (32 -> 64, k 4) layer; g is NHWC."""
    global _c2_wpk_floats
    L = _lib()
    if _c2_wpk_floats is None:
        n = C.c_int64()
        L.check(L.lib.mirl_conv2_bwd_data_wpk_floats(C.byref(n)), "mirl_conv2_bwd_data_wpk_floats")
        _c2_wpk_floats = n.value
    dx = torch.empty_like(x_like, memory_format=torch.channels_last)
    wpk = torch.empty(_c2_wpk_floats, dtype=torch.float32, device=g.device)
    so, sc, sh, sw = weight.stride()
    L.check(L.lib.mirl_conv2_bwd_data_ex(g.shape[0], g.shape[2], g.shape[3], _p(g), _p(weight), so, sc, sh, sw, _p(wpk), wpk.numel(),
                                         _p(dx), _CONV2_BWD_PIPE if pipe is None else pipe, _stream()), "mirl_conv2_bwd_data_ex")
    return dx


def conv_bias_relu(x, conv):
    """relu(conv(x)) for an nn.Conv2d (valid padding, no dilation / groups) on NHWC input."""
    if (_fusable(x, conv.weight) and conv.bias is not None and conv.padding == (0, 0) and conv.dilation == (1, 1)
            and conv.groups == 1 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and conv.weight.is_contiguous(memory_format=torch.channels_last)):
        return _ConvBiasReLU.apply(x, conv.weight, conv.bias, tuple(conv.stride), torch.is_grad_enabled())
    return F.relu(conv(x))


def frames_to_f32_nhwc(x, scale):
    """uint8 [N, C, H, W] -> float32 * scale with channels_last memory, one pass (csrc/convert.hip)."""
    L = _lib()
    n, c, h, w = x.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    L.check(L.lib.mirl_frames_to_f32_nhwc(n, c, h * w, _p(x), float(scale), _p(out), _stream()), "mirl_frames_to_f32_nhwc")
    return out


def _wpk_floats():
    need = C.c_int64()
    L = _lib()
    L.check(L.lib.mirl_conv1_u8_wpk_floats(C.byref(need)))
    return need.value


def conv_u8_supported(x, conv):
    """Does the one-pass input layer (mirl_conv1_u8_fwd) cover this conv on this uint8 block?"""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.uint8 and x.dim() == 4 and x.is_contiguous()
            and x.data_ptr() % 16 == 0 and conv.bias is not None and conv.weight.dtype == torch.float32
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.kernel_size[0] == conv.kernel_size[1] and conv.stride[0] == conv.stride[1]
            and conv.in_channels == x.shape[1] and not torch.is_autocast_enabled()):
        return False
    return bool(_lib().lib.mirl_conv1_u8_supported(x.shape[1], x.shape[2], x.shape[3], conv.out_channels,
                                                   conv.kernel_size[0], conv.stride[0]))


_U8_WRW = os.environ.get("MIRL_CONV1_WRW", "1") != "0"     # 0: MIOpen weight gradient on a converted copy
_U8_WRW_MASK = os.environ.get("MIRL_CONV1_WRW_MASK", "1") != "0"   # 0: separate ReLU-mask + bias-gradient pass in front of it


class _ConvU8BiasReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, scale, stride):
        L = _lib()
        n, _, h, w = x.shape
        f, k = weight.shape[0], weight.shape[2]
        oh, ow = (h - k) // stride + 1, (w - k) // stride + 1
        y = torch.empty((n, f, oh, ow), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        if n:
            wpk = torch.empty(_wpk_floats(), dtype=torch.float32, device=x.device)
            so, sc, sh, sw = weight.stride()
            b = bias if bias.data_ptr() % 16 == 0 else bias.clone()
            L.check(L.lib.mirl_conv1_u8_fwd(n, h, w, _p(x), _p(weight), so, sc, sh, sw, _p(b), float(scale), _p(wpk), _p(y),
                                            _stream()), "mirl_conv1_u8_fwd")
        ctx.scale, ctx.stride = scale, stride
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, grad):
        x, weight, y = ctx.saved_tensors
        grad = grad.contiguous(memory_format=torch.channels_last)
        if ctx.needs_input_grad[1] and _U8_WRW and _U8_WRW_MASK and x.shape[0] and grad.data_ptr() % 16 == 0:
            # ReLU mask and bias gradient inside the weight-gradient kernel (this layer's input takes no gradient, so the
            # masked gradient itself is not needed anywhere else): one pass over dy, y and the uint8 frames
            L = _lib()
            n, _, h, w = x.shape
            need = C.c_int64()
            L.check(L.lib.mirl_conv1_u8_wrw_scratch_floats(C.byref(need)))
            scratch = torch.empty(need.value, dtype=torch.float32, device=x.device)
            dw = torch.empty_like(weight)
            db = torch.empty(weight.shape[0], dtype=torch.float32, device=x.device)
            so, sc, sh, sw = dw.stride()
            L.check(L.lib.mirl_conv1_u8_wrw_masked(n, h, w, _p(x), _p(grad), _p(y), float(ctx.scale), _p(scratch), _p(dw), so, sc, sh, sw,
                                                   _p(db), _stream()), "mirl_conv1_u8_wrw_masked")
            return None, dw, (db if ctx.needs_input_grad[2] else None), None, None
        g, db = relu_bwd_bias_rows(grad, y, y.shape[1])
        dw = None
        if ctx.needs_input_grad[1] and _U8_WRW and x.shape[0]:
            # weight gradient from the uint8 frames as well (csrc/conv_in.hip): no float pixels anywhere
            L = _lib()
            n, _, h, w = x.shape
            need = C.c_int64()
            L.check(L.lib.mirl_conv1_u8_wrw_scratch_floats(C.byref(need)))
            scratch = torch.empty(need.value, dtype=torch.float32, device=x.device)
            dw = torch.empty_like(weight)
            so, sc, sh, sw = dw.stride()
            L.check(L.lib.mirl_conv1_u8_wrw(n, h, w, _p(x), _p(g), float(ctx.scale), _p(scratch), _p(dw), so, sc, sh, sw,
                                            _stream()), "mirl_conv1_u8_wrw")
        elif ctx.needs_input_grad[1] and not x.shape[0]:
            dw = torch.zeros_like(weight)
        elif ctx.needs_input_grad[1]:
            # library weight gradient: the only consumer of float pixels, converted here for
            # the rows that take part in the backward only
            xf = frames_to_f32_nhwc(x, ctx.scale)
            _, dw, _ = torch.ops.aten.convolution_backward(
                g, xf, weight, None, [ctx.stride, ctx.stride], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])
        return None, dw, (db if ctx.needs_input_grad[2] else None), None, None


def conv_u8_bias_relu(x, conv, scale):
    """relu(conv(x * scale)) for uint8 NCHW x; the caller checked conv_u8_supported(x, conv)."""
    return _ConvU8BiasReLU.apply(x, conv.weight, conv.bias, float(scale), int(conv.stride[0]))


def cos_embed(taus, freq):
    """cos(freq * taus[:, None]) -> (len(taus), len(freq)); freq = embedding_range * pi as float32."""
    D = freq.shape[0]
    if D % 4 == 0 and _fusable(taus, freq) and taus.is_contiguous():
        L = _lib()
        out = torch.empty((taus.shape[0], D), dtype=torch.float32, device=taus.device)
        L.check(L.lib.mirl_cos_embed(taus.shape[0], D, _p(taus), _p(freq), _p(out), _stream()), "mirl_cos_embed")
        return out
    return torch.cos(freq * taus.unsqueeze(1))


class _QPLink:
    """Hand-over between the feature product and the layer that consumes its (rows, C) output — at the benchmark shape a
    2.7 GB tensor whose gradient would be written by the consumer's data-gradient GEMM only to be read back by the
    product's backward.  The product announces its output; a consumer that recognises its input (the dueling tail) runs
    the product's backward in the epilogue of that GEMM (csrc/gemm3.hip EP 3), leaves the results here and returns an
    all-zero, zero-stride placeholder as the gradient — autograd adds zeros to whatever other consumers contribute, so
    the product's backward stays correct for any graph."""
    __slots__ = ("out", "x", "emb", "n", "done", "placeholder", "__weakref__")


_qp_links = {}


def _qp_announce(out, x, emb, n):
    import weakref
    link = _QPLink()
    link.out, link.x, link.emb, link.n, link.done, link.placeholder = weakref.ref(out), x, emb, n, None, None
    if len(_qp_links) > 16:
        for k in [k for k, v in _qp_links.items() if v() is None or v().out() is None]:
            del _qp_links[k]
    _qp_links[out.data_ptr()] = weakref.ref(link)
    return link


def _qp_find(x):
    """The link of the feature product whose output `x` is (the tensor itself or a same-shape view of it), or None."""
    ref = _qp_links.get(x.data_ptr())
    link = ref() if ref is not None else None
    out = link.out() if link is not None else None
    if out is None or not (x is out or x._base is out) or x.shape != out.shape or not x.is_contiguous() or x._version != out._version:
        return None
    return link


class _QuantileProduct(torch.autograd.Function):
    """out[m*N+n] = x[m] * relu(phi[m*N+n] @ Wq^T + bq)."""

    @staticmethod
    def forward(ctx, x, phi, weight, bias, n, track=True):
        L = _lib()
        M, Cf = x.shape
        # no-grad passes (target / selection / acting): the embedding is not needed again
        # (`track` comes from the caller's torch.is_grad_enabled(): ctx.needs_input_grad mirrors
        # requires_grad even under no_grad)
        need = track and any(ctx.needs_input_grad)
        if _QP_EPILOGUE and gemm3.quantile_product_supported(x, phi, weight, bias, n):
            # embedding GEMM + ReLU + product in ONE launch (split-bf16 kernel, multiply in its epilogue)
            out, emb = gemm3.quantile_product(x, phi, weight, bias, n, need)
        else:
            emb = torch._addmm_activation(bias, phi, weight.t(), use_gelu=False)
            out = torch.empty_like(emb) if need else emb     # no-grad: multiply in place
            L.check(L.lib.mirl_iqn_mul_fwd(M, n, Cf, _p(x), _p(emb), _p(out), _stream()), "mirl_iqn_mul_fwd")
        ctx.n = n
        ctx.save_for_backward(x, phi, weight, emb)
        ctx.link = _qp_announce(out, x, emb, n) if need else None
        return out

    @staticmethod
    def backward(ctx, grad):
        L = _lib()
        x, phi, weight, emb = ctx.saved_tensors
        M, Cf = x.shape
        link = getattr(ctx, "link", None)
        done, fused = (link.done, link.placeholder) if link is not None else (None, None)
        if link is not None:
            link.done = link.placeholder = None
        # the consumer's backward already ran this product's backward in its data-gradient GEMM (see _QPLink): `grad` is
        # then its all-zero placeholder — or placeholder + the gradients of OTHER consumers of the output, whose share
        # (the backward is linear in grad) goes through the stand-alone pass and is added
        only_placeholder = done is not None and grad.data_ptr() == fused.data_ptr() and all(s == 0 for s in grad.stride())
        if done is None or not only_placeholder:
            grad = grad.contiguous()
            if _pow2_quads(Cf):
                blocks = min(M, 2048)
                d_pre = torch.empty_like(emb)
                dx = torch.empty_like(x)
                db = torch.empty(Cf, dtype=torch.float32, device=x.device)
                partial = torch.empty((blocks, Cf), dtype=torch.float32, device=x.device)
                L.check(L.lib.mirl_iqn_mul_bwd(M, ctx.n, Cf, _p(grad), _p(emb), _p(x), _p(d_pre), _p(dx), _p(db), _p(partial),
                                               blocks, _stream()), "mirl_iqn_mul_bwd")
            else:
                # feature widths the stand-alone kernel does not take (3136 = the conv stack's output without an FC layer in
                # front of the product): only reached when no consumer ran this backward in its data-gradient GEMM
                g3, e3 = grad.view(M, ctx.n, Cf), emb.view(M, ctx.n, Cf)
                dx = (g3 * e3).sum(1)
                d_pre = (g3 * x.unsqueeze(1)).masked_fill_(e3 <= 0, 0.0).view(M * ctx.n, Cf)
                db = d_pre.sum(0)
            if done is not None:
                d_pre += done[0]
                dx += done[1]
                db += done[2]
        else:
            d_pre, dx, db = done
        dw = gemm3.grad_weight(d_pre, phi) if ctx.needs_input_grad[2] else None
        return (dx if ctx.needs_input_grad[0] else None), None, dw, (db if ctx.needs_input_grad[3] else None), None, None


def quantile_product(x, phi, weight, bias, n):
    """x (M, C), phi (M*n, D) -> (M*n, C): x[m] * relu(linear(phi))[m*n + j]   (iqn.py:82-102)."""
    if x.dim() == 2 and _fusable(x, phi, weight) and x.is_contiguous() and phi.is_contiguous() and not phi.requires_grad:
        # any width the GEMM kernel's epilogue takes; the stand-alone multiply only 4 * 2^k columns
        if _pow2_quads(x.shape[1]) or (_QP_EPILOGUE and gemm3.quantile_product_supported(x, phi, weight, bias, n)):
            return _QuantileProduct.apply(x, phi, weight, bias, n, torch.is_grad_enabled())
    emb = linear_relu(phi, weight, bias)
    return (x.unsqueeze(1) * emb.reshape(x.shape[0], n, -1)).reshape(x.shape[0] * n, -1)


class _DuelingTail(torch.autograd.Function):
    """The dueling head's two parallel hidden layers as ONE GEMM.

    reference: the model's last FC layer h = relu(x W1^T + b1) feeding the
    advantage outputs a = h Wo^T + bo (policies/torch/dqn.py:101-112), and in
    parallel the value branch hv = relu(x Wv^T + bv), v = hv Wq^T + bq
    (dqn.py:50-66,74-87).  Both hidden layers read the same (rows, F) input — at
    the IQN benchmark shape 1.31 M x 512 floats = 2.7 GB — so they run as one
    (rows, H1 + Hv) GEMM with the ReLU in its epilogue; the backward builds the
    gradient of that joint activation in place (two small GEMMs into its two column
    blocks), masks it and reduces the bias gradients in one pass, and produces
    dx with one GEMM instead of two GEMMs and an add."""

    @staticmethod
    def forward(ctx, x, w1, b1, wo, bo, wv, bv, wq, bq, track=True):
        h1 = w1.shape[0]
        wj, bj = gemm3.joint_rows((w1, wv)), gemm3.joint_rows((b1, bv))
        na, nv = wo.shape[0], wq.shape[0]
        w2 = gemm3.joint_blockdiag((wo, wq)) if (na + nv <= 8 and wo.shape[1] == h1) else None
        if w2 is not None and gemm3.head_supported(x, wj, bj, w2):
            # both output layers in the epilogue of the joint hidden layer's product (csrc/gemm3.hip EP 2): the
            # (rows, H1 + Hv) activation is not read back by two more GEMMs and, when no backward follows, never stored
            keep = bool(track) and any(ctx.needs_input_grad)
            both, out = gemm3.linear_relu_head(x, wj, bj, w2, gemm3.joint_rows((bo, bq)), keep)
            a, v = out[:, :na], out[:, na:]
        else:
            both = gemm3.linear_fwd(x, wj, bj, relu=True)
            a = torch.addmm(bo, both[:, :h1], wo.t())
            v = torch.addmm(bq, both[:, h1:], wq.t())
        ctx.h1 = h1
        ctx.link = _qp_find(x) if (track and ctx.needs_input_grad[0]) else None
        ctx.save_for_backward(x, w1, wo, wv, wq, both)
        return a, v

    @staticmethod
    def backward(ctx, ga, gv):
        x, w1, wo, wv, wq, both = ctx.saved_tensors
        h1 = ctx.h1
        ga, gv = ga.contiguous(), gv.contiguous()
        width = both.shape[1]
        dwo_dwq = None
        if (_pow2_quads(width) and h1 % 4 == 0 and ga.shape[1] <= 16 and gv.shape[1] <= 16
                and wo.is_contiguous() and wq.is_contiguous()):
            # [ga @ wo | gv @ wq], ReLU mask and bias gradient in ONE pass over `both`
            L = _lib()
            M = both.shape[0]
            blocks = C.c_int32()
            L.check(L.lib.mirl_colsum_blocks(M, width, C.byref(blocks)))
            g = torch.empty_like(both)
            db = torch.empty(width, dtype=torch.float32, device=both.device)
            partial = torch.empty((blocks.value, width), dtype=torch.float32, device=both.device)
            kw = max(ga.shape[1], gv.shape[1])
            if kw <= 8 and _TAIL_WGRAD:
                # the output layers' weight gradients from the same pass over `both` (no (A | Q) x rows x H GEMMs)
                dwj = torch.empty((kw, width), dtype=torch.float32, device=both.device)
                partial_w = torch.empty((blocks.value, kw, width), dtype=torch.float32, device=both.device)
                L.check(L.lib.mirl_dueling_tail_bwd_w(M, h1, width - h1, ga.shape[1], gv.shape[1], _p(ga), _p(gv), _p(wo), _p(wq),
                                                      _p(both), _p(g), _p(db), _p(partial), blocks.value, _p(dwj), _p(partial_w),
                                                      _stream()), "mirl_dueling_tail_bwd_w")
                dwo_dwq = (dwj[:ga.shape[1], :h1], dwj[:gv.shape[1], h1:])
            else:
                L.check(L.lib.mirl_dueling_tail_bwd(M, h1, width - h1, ga.shape[1], gv.shape[1], _p(ga), _p(gv), _p(wo), _p(wq),
                                                    _p(both), _p(g), _p(db), _p(partial), blocks.value, _stream()),
                        "mirl_dueling_tail_bwd")
        else:
            d_both = torch.empty_like(both)
            torch.mm(ga, wo, out=d_both[:, :h1])
            torch.mm(gv, wq, out=d_both[:, h1:])
            if _pow2_quads(width):
                g, db = relu_bwd_bias_rows(d_both, both, width)
            else:
                g = torch.ops.aten.threshold_backward(d_both, both, 0.0)
                db = g.sum(0)
        # data / weight gradients per branch on the two column blocks (strided views):
        # hipBLASLt's 512-wide kernels measured faster than one 1024-wide GEMM here
        # (10.3 + 11.5 ms merged vs 4 x 3.55 ms, profiles/r02b)
        g1, g2 = g[:, :h1], g[:, h1:]
        dx = None
        wj = gemm3.joint_rows((w1, wv)) if gemm3.enabled() else None
        if wj is not None and gemm3.supported(gemm3.NN, g, wj) and gemm3.supported(gemm3.TN, g, x):
            # the split-bf16 kernel takes the joint (rows, H1 + Hv) gradient as ONE K = H1 + Hv data gradient
            # and ONE weight gradient whose row blocks are dW1 | dWv
            link = getattr(ctx, "link", None)
            if ctx.needs_input_grad[0] and link is not None and gemm3.grad_input_qp_supported(g, wj, link.emb, link.x, link.n):
                # x is the IQN feature product's output: its backward rides in this GEMM's epilogue, dx is never stored
                link.done = gemm3.grad_input_qp(g, wj, link.emb, link.x)
                link.placeholder = dx = torch.zeros(1, dtype=x.dtype, device=x.device).expand(x.shape)
            elif ctx.needs_input_grad[0]:
                dx = gemm3.gemm(gemm3.NN, g, wj, weight_b=True)
            dwj = gemm3.gemm(gemm3.TN, g, x)
            dw1, dwv = dwj[:h1], dwj[h1:]
        else:
            if ctx.needs_input_grad[0]:
                dx = g1.mm(w1)
                dx.addmm_(g2, wv)
            dw1, dwv = g1.t().mm(x), g2.t().mm(x)
        if dwo_dwq is not None:
            dwo, dwq = dwo_dwq
        else:
            dwo = ga.t().mm(both[:, :h1])
            dwq = gv.t().mm(both[:, h1:])
        return dx, dw1, db[:h1], dwo, ga.sum(0), dwv, db[h1:], dwq, gv.sum(0), None


def dueling_tail(x, fc, out_layer, value_hidden, value_layer):
    """-> (advantage outputs, value outputs) for nn.Linear modules; x 2-D."""
    if x.dim() == 2 and _fusable(x, fc.weight) and x.is_contiguous():
        return _DuelingTail.apply(x, fc.weight, fc.bias, out_layer.weight, out_layer.bias,
                                  value_hidden.weight, value_hidden.bias, value_layer.weight, value_layer.bias,
                                  torch.is_grad_enabled())
    h = F.relu(fc(x))
    return out_layer(h), value_layer(F.relu(value_hidden(x)))
