"""f32 GEMMs on the bf16 matrix pipe (csrc/gemm3.hip): exact three-way bf16 split of both
operands, six part products accumulated in f32.  Drop-in for the library calls behind the
reference's nn.Linear layers (rltime/policies/torch/dqn.py:50-112, iqn.py:82-102,
modules/lstm.py:60-81) and their autograd gradients; results are f32 GEMM results
(tests/test_gemm3_gpu.py).  `enabled()` is the switch the callers consult: MIRL_GEMM3=0
keeps the library path."""
import ctypes as C
import os

import torch

NT, NN, TN = 0, 1, 2
# M*N*K below this stays on the library (launch-bound anyway).  MIRL_GEMM3_MIN_WORK=0 sends every product the
# kernel takes through it (tests/test_e2e_gpu.py pins the reference trajectories that way).
_MIN_WORK = int(os.environ.get("MIRL_GEMM3_MIN_WORK", str(1 << 31)))
_ws = {}


def _lib():
    from rltime_amd import _lib as L
    return L


def enabled():
    return os.environ.get("MIRL_GEMM3", "1") != "0"


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rowmajor(t):
    return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] and t.dtype == torch.float32 and t.is_cuda


def supported(layout, a, b, min_work=None):
    """Shapes / strides the kernel takes for `layout` with operands as stored (no copies)."""
    if not (_rowmajor(a) and _rowmajor(b)):
        return False
    if layout == NT:
        M, K, N = a.shape[0], a.shape[1], b.shape[0]
        ok = b.shape[1] == K and a.stride(0) % 4 == 0 and b.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
    elif layout == NN:
        M, K, N = a.shape[0], a.shape[1], b.shape[1]
        ok = b.shape[0] == K and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0
    else:
        K, M, N = a.shape[0], a.shape[1], b.shape[1]
        ok = b.shape[0] == K
    if not ok or M * N * K < (_MIN_WORK if min_work is None else min_work):
        return False
    return bool(_lib().lib.mirl_gemm3_supported(layout, M, N, K))


def _workspace(device, nbytes):
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def gemm(layout, a, b, bias=None, relu=False, out=None):
    """NT: a (M,K) @ b (N,K)^T [+ bias, ReLU];  NN: a (M,K) @ b (K,N);  TN: a (K,M)^T @ b (K,N)."""
    L = _lib()
    if layout == NT:
        M, K, N = a.shape[0], a.shape[1], b.shape[0]
    elif layout == NN:
        M, K, N = a.shape[0], a.shape[1], b.shape[1]
    else:
        K, M, N = a.shape[0], a.shape[1], b.shape[1]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    ws, wsb = None, 0
    if layout == TN:
        need = C.c_int64()
        L.check(L.lib.mirl_gemm3_workspace_bytes(layout, M, N, K, C.byref(need)), "mirl_gemm3_workspace_bytes")
        wsb = need.value
        ws = _workspace(a.device, wsb)
    L.check(L.lib.mirl_gemm3(layout, M, N, K, _p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0),
                             _p(bias) if bias is not None else None, 1 if relu else 0,
                             _p(ws) if ws is not None else None, wsb, _stream()), "mirl_gemm3")
    return out


# ---- the three products of a linear layer, each on the split-bf16 kernel when it takes the operands as
# ---- stored and on the library otherwise -------------------------------------------------------------------

def linear_fwd(x, w, bias=None, relu=False):
    """x (M,K) @ w (N,K)^T [+ bias] [ReLU]  (nn.Linear forward)."""
    if enabled() and supported(NT, x, w) and (bias is None or (bias.is_contiguous() and bias.dtype == torch.float32)):
        return gemm(NT, x, w, bias, relu)
    if relu:
        return torch._addmm_activation(bias, x, w.t(), use_gelu=False) if bias is not None else torch.relu(x.mm(w.t()))
    return torch.addmm(bias, x, w.t()) if bias is not None else x.mm(w.t())


def grad_input(g, w):
    """g (M,N) @ w (N,K) -> (M,K): gradient w.r.t. a linear layer's input."""
    if enabled() and supported(NN, g, w):
        return gemm(NN, g, w)
    return g.mm(w)


def grad_weight(g, x):
    """g (M,N)^T @ x (M,K) -> (N,K): gradient w.r.t. a linear layer's weight."""
    if enabled() and supported(TN, g, x):
        return gemm(TN, g, x)
    return g.t().mm(x)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        return linear_fwd(x, w, bias)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        dx = grad_input(g, w) if ctx.needs_input_grad[0] else None
        dw = grad_weight(g, x) if ctx.needs_input_grad[1] else None
        db = g.sum(0) if ctx.needs_input_grad[2] else None
        return dx, dw, db


def linear(x, w, bias):
    """F.linear(x, w, bias) for 2-D float32 x (rltime/models/torch/modules/lstm.py:60-81 input projection)."""
    if enabled() and x.dim() == 2 and supported(NT, x, w) and bias is not None:
        return _Linear.apply(x, w, bias.contiguous())
    return torch.nn.functional.linear(x, w, bias)


def quantile_product_supported(x, phi, weight, bias, n):
    """Can relu(linear(phi)) * x[row // n] run as ONE NT product with the multiply in its epilogue?"""
    return (enabled() and n > 0 and (n & (n - 1)) == 0 and supported(NT, phi, weight) and _rowmajor(x)
            and x.shape[1] == weight.shape[0] and x.shape[0] * n == phi.shape[0]
            and bias is not None and bias.is_contiguous() and bias.dtype == torch.float32)


def quantile_product(x, phi, weight, bias, n, keep_embedding):
    """-> (out, emb or None): out[m*n + j] = x[m] * relu(phi[m*n + j] @ weight^T + bias)  (iqn.py:82-102)."""
    L = _lib()
    R, K, N = phi.shape[0], phi.shape[1], weight.shape[0]
    out = torch.empty((R, N), dtype=torch.float32, device=phi.device)
    emb = torch.empty((R, N), dtype=torch.float32, device=phi.device) if keep_embedding else None
    L.check(L.lib.mirl_gemm3_nt_mul(R, N, K, _p(phi), phi.stride(0), _p(weight), weight.stride(0), _p(out), out.stride(0),
                                    _p(bias), 1, _p(x), x.stride(0), n.bit_length() - 1,
                                    _p(emb) if emb is not None else None, N, _stream()), "mirl_gemm3_nt_mul")
    return out, emb
