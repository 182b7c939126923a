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
# NT / NN run one 256 x 256 output tile per workgroup over the whole K: with fewer than ~96 tiles most of the chip idles for
# 0.104 us x K while the library's smaller tiles finish the product at ~120 TFLOP/s — 1024 x 1024 x 3136 (the 32-env acting
# batch of the Rainbow-IQN head): 325 us here, 57 us there; break-even at M x N ~ 6e6 (profiles/r05_gemm3_small_m.jsonl).
# Not applied when the work threshold is 0 (MIRL_GEMM3_MIN_WORK=0: everything the kernel takes goes through it).
_MIN_AREA = int(os.environ.get("MIRL_GEMM3_MIN_AREA", "6000000"))
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
    floor = _MIN_WORK if min_work is None else min_work
    if not ok or M * N * K < floor or (floor > 0 and layout != TN and M * N < _MIN_AREA):
        return False
    return bool(_lib().lib.mirl_gemm3_supported(layout, M, N, K))


def _workspace(device, nbytes):
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


# Set while a learner step is being captured into a HIP graph (training/torch_trainer.py): every version-keyed cache below
# then rebuilds its buffer IN PLACE inside the capture, so the rebuild is part of the graph and re-runs with every replay
# (a replay moves no version counter: a cache hit at capture time would freeze that operand for good).
REFRESH_ALWAYS = False
_joint = {}


def joint_rows(tensors):
    """torch.cat(tensors, 0) of long-lived weights / biases as ONE persistent tensor, rebuilt only when a source's version
    counter moved (the dueling head's [last FC | value-hidden] weights: fused.py)."""
    key = tuple(t.data_ptr() for t in tensors)
    vers = tuple(t._version for t in tensors)
    hit = _joint.get(key)
    if hit is not None and hit[0] == vers and not REFRESH_ALWAYS:
        return hit[1]
    with torch.no_grad():
        if hit is not None and hit[1].shape[0] == sum(t.shape[0] for t in tensors):
            out = hit[1]
            torch.cat([t.detach() for t in tensors], 0, out=out)
        else:
            out = torch.cat([t.detach() for t in tensors], 0)
    if len(_joint) > 32:
        _joint.clear()
    _joint[key] = (vers, out, tensors)
    return out


def gemm(layout, a, b, bias=None, relu=False, out=None, weight_b=False):
    """NT: a (M,K) @ b (N,K)^T [+ bias, ReLU];  NN: a (M,K) @ b (K,N);  TN: a (K,M)^T @ b (K,N).
    weight_b: b is a weight (kept for the call sites; a pre-split cache of weight operands was measured slower and removed:
    profiles/r04_gemm3_probe_presplit.jsonl)."""
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
        return gemm(NT, x, w, bias, relu, weight_b=True)
    if relu:
        return torch._addmm_activation(bias, x, w.t(), use_gelu=False) if bias is not None else torch.relu(x.mm(w.t()))
    return torch.addmm(bias, x, w.t()) if bias is not None else x.mm(w.t())


def grad_input(g, w):
    """g (M,N) @ w (N,K) -> (M,K): gradient w.r.t. a linear layer's input."""
    if enabled() and supported(NN, g, w):
        return gemm(NN, g, w, weight_b=True)
    return g.mm(w)


def grad_weight(g, x):
    """g (M,N)^T @ x (M,K) -> (N,K): gradient w.r.t. a linear layer's weight."""
    if enabled() and supported(TN, g, x):
        return gemm(TN, g, x)
    return g.t().mm(x)


_QP_BWD = os.environ.get("MIRL_GEMM3_QP_BWD", "1") != "0"   # 0: data gradient stored, then the separate feature-product backward pass


def grad_input_qp_supported(g, w, emb, x, n):
    """Can g (M,N) @ w (N,K) run with the IQN feature product's backward (out[r] = x[r // n] * emb[r]) in its epilogue?"""
    return (_QP_BWD and enabled() and n == 32 and supported(NN, g, w) and _rowmajor(emb) and _rowmajor(x)
            and emb.shape == (g.shape[0], w.shape[1]) and x.shape == (g.shape[0] // 32, w.shape[1]) and g.shape[0] % 32 == 0
            and w.shape[1] % 4 == 0 and emb.stride(0) % 4 == 0 and x.stride(0) % 4 == 0
            and emb.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)


def grad_input_qp(g, w, emb, x):
    """-> (d_pre (M,K), dx (M/32,K), db (K,)) of the feature product for the gradient d = g @ w of its output; d is not stored."""
    L = _lib()
    M, N = g.shape
    K = w.shape[1]
    rows = C.c_int64()
    L.check(L.lib.mirl_gemm3_nn_qp_partial_rows(M, C.byref(rows)), "mirl_gemm3_nn_qp_partial_rows")
    d_pre = torch.empty((M, K), dtype=torch.float32, device=g.device)
    dx = torch.empty((M // 32, K), dtype=torch.float32, device=g.device)
    part = torch.empty((rows.value, K), dtype=torch.float32, device=g.device)
    # NN form: the contraction runs over g's columns, the result is K wide
    L.check(L.lib.mirl_gemm3_nn_qp(M, K, N, _p(g), g.stride(0), _p(w), w.stride(0), _p(emb), emb.stride(0), _p(x), x.stride(0),
                                   _p(d_pre), d_pre.stride(0), _p(dx), dx.stride(0), _p(part), _stream()), "mirl_gemm3_nn_qp")
    return d_pre, dx, part.sum(0)


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


# ---- a wide layer + the narrow layer behind it in one pass (csrc/gemm3.hip mirl_gemm3_nt_head) ---------------------
_HEAD = os.environ.get("MIRL_GEMM3_HEAD", "1") != "0"


def head_supported(x, w, bias, w2):
    """relu(x @ w^T + bias) followed by <= 8 output units w2 (O, N): can both run as ONE NT launch?"""
    return (_HEAD and enabled() and supported(NT, x, w) and bias is not None and bias.is_contiguous() and bias.dtype == torch.float32
            and w.shape[0] % 4 == 0 and _rowmajor(w2) and w2.is_contiguous() and w2.shape[1] == w.shape[0] and 1 <= w2.shape[0] <= 8)


def linear_relu_head(x, w, bias, w2, bias2, keep_hidden):
    """-> (hidden = relu(x @ w^T + bias) or None, out = hidden @ w2^T + bias2).  keep_hidden=False: the hidden
    activation never reaches HBM."""
    L = _lib()
    M, K, N, O = x.shape[0], x.shape[1], w.shape[0], w2.shape[0]
    pad = joint_pad8(w2)
    hidden = torch.empty((M, N), dtype=torch.float32, device=x.device) if keep_hidden else None
    out = torch.empty((M, O), dtype=torch.float32, device=x.device)
    need = C.c_int64()
    L.check(L.lib.mirl_gemm3_head_workspace_bytes(M, N, C.byref(need)), "mirl_gemm3_head_workspace_bytes")
    ws = _workspace(x.device, need.value)
    L.check(L.lib.mirl_gemm3_nt_head(M, N, K, _p(x), x.stride(0), _p(w), w.stride(0), _p(hidden) if hidden is not None else None, N,
                                     _p(bias), 1, _p(pad), O, _p(bias2) if bias2 is not None else None, _p(out), O,
                                     _p(ws), need.value, _stream()), "mirl_gemm3_nt_head")
    return hidden, out


_pad8 = {}


def joint_pad8(w2):
    """(O, N) -> persistent zero-padded (8, N) copy, refreshed when w2's version counter moves."""
    key = (w2.data_ptr(), tuple(w2.shape))
    hit = _pad8.get(key)
    if hit is not None and hit[0] == w2._version and hit[2].untyped_storage().data_ptr() == w2.untyped_storage().data_ptr() \
            and not REFRESH_ALWAYS:
        return hit[1]
    with torch.no_grad():
        pad = hit[1] if hit is not None else torch.zeros((8, w2.shape[1]), dtype=torch.float32, device=w2.device)
        pad[:w2.shape[0]].copy_(w2.detach())
    if len(_pad8) > 32:
        _pad8.clear()
    _pad8[key] = (w2._version, pad, w2.detach())
    return pad


def joint_blockdiag(blocks):
    """Block-diagonal stack of long-lived weights [(O_i, N_i)] -> persistent (sum O_i, sum N_i) tensor (zeros elsewhere),
    refreshed when a source's version counter moves: the dueling head's [advantage | value] output weights over the joint
    [last FC | value-hidden] activation."""
    key = tuple(t.data_ptr() for t in blocks)
    vers = tuple(t._version for t in blocks)
    hit = _joint.get(("bd",) + key)
    if hit is not None and hit[0] == vers and not REFRESH_ALWAYS:
        return hit[1]
    with torch.no_grad():
        out = hit[1] if hit is not None else torch.zeros((sum(t.shape[0] for t in blocks), sum(t.shape[1] for t in blocks)),
                                                          dtype=torch.float32, device=blocks[0].device)
        r = c = 0
        for t in blocks:
            out[r:r + t.shape[0], c:c + t.shape[1]].copy_(t.detach())
            r, c = r + t.shape[0], c + t.shape[1]
    _joint[("bd",) + key] = (vers, out, blocks)
    return out
