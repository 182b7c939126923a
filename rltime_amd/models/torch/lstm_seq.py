"""Fused LSTM sequence op for the recurrent core (GPU only).

Same recurrence as the reference's LSTMCell time loop with per-step reset
(rltime/models/torch/modules/lstm.py:83-116):

    h_in = h(t-1) * keep(t) ; c_in = c(t-1) * keep(t) ; keep = 1 - initials
    gates = W_ih x(t) + b_ih + W_hh h_in + b_hh       (i, f, g, o)
    c(t) = sig(f) c_in + sig(i) tanh(g) ; h(t) = sig(o) tanh(c(t))

but the input projection of all T steps is ONE GEMM, a forward step is one rocBLAS
GEMM accumulated IN PLACE onto its slice of that projection + one fused cell kernel
(csrc/lstm.hip mirl_lstm_cell_fwd), a backward step one cell
kernel (mirl_lstm_cell_bwd) + one rocBLAS GEMM, and the weight
gradients of W_ih and W_hh are one GEMM each over all timesteps after the
backward sweep.
"""
import ctypes as C

import torch

from rltime_amd._lib import lib, check
from . import gemm3


import os

# (A one-launch step kernel — recurrent GEMM on f32 MFMA with the cell in its epilogue — was built in round 2 and measured no
# faster than GEMM + cell at B = H = 512 (18.8 vs 12.6 + 5.7 us): the gain is in keeping W_hh resident across steps, which the
# persistent kernel below does.  Removed in round 5; at acting batch sizes csrc/actnet.hip k_act_lstm is that idea done right.)

# The persistent sequence kernel (csrc/lstm_seq.hip: one launch per sweep, W_hh resident in
# LDS, h exchanged between workgroups through write-through stores + arrival counters) is
# the default wherever it supports the shape; MIRL_LSTM_PERSISTENT=0 keeps the per-step path.
_PERSISTENT = os.environ.get("MIRL_LSTM_PERSISTENT", "1") != "0"
# The persistent BACKWARD sweep exchanges 32 KB of partial sums per wave and step (write-through);
# measured 13.3 us per step at B = 64 and 17.9 at B = 256, but 28 at B = 512 where the chip-wide
# 64 MB per step saturate the fabric — no better than the cell kernel + rocBLAS GEMM per step
# (20.6 us inside the learner step; profiles/r03_lstm_probe_with_persistent_backward.jsonl).  It is
# therefore the default for small per-GPU batches only (strong scaling: B = 64 per rank at 8 GPUs).
_BWD_PERSISTENT_MAX_B = int(os.environ.get("MIRL_LSTM_BWD_PERSISTENT_MAX_B", "128"))


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def check_status(what="a persistent LSTM sweep"):
    """Raise if a persistent sweep has reported failure (a bounded spin gave up on a peer workgroup, or the recurrent state
    became non-finite — the tagged exchange cannot carry inf / nan).  The kernels set a host-visible word; the next
    mirl_lstm_seq_* call also refuses with MIRL_ERR_STATE, but by then the invalid outputs of the failed sweep may have
    reached the optimizer — the trainers call this where the host is already synchronised (log rows, checkpoints) and,
    as a cheap unsynchronised look, before every optimizer step.  Shared-GPU runs that starve the sweep's co-residency
    assumption should set MIRL_LSTM_PERSISTENT=0."""
    st = C.c_int32()
    check(lib.mirl_lstm_seq_status(C.byref(st)), "mirl_lstm_seq_status")
    if st.value:
        raise RuntimeError("%s reported failure (status %d): its outputs are invalid — non-finite recurrent state, or a "
                           "workgroup of the sweep was not resident (another process / stream occupying the GPU: "
                           "MIRL_LSTM_PERSISTENT=0 keeps the per-step kernels)" % (what, st.value))


def persistent_supported(T, B, H):
    return _PERSISTENT and T >= 2 and bool(lib.mirl_lstm_seq_supported(T, B, H))


def two_sweeps_fit(B, H):
    """Can two forward sweeps of this shape run on two streams at once?  The persistent kernel spins on peer workgroups
    and needs its whole grid resident; two grids that do not fit the chip together would starve each other until the
    bounded spin gives up.  True for shapes the per-step path serves (no residency assumption there)."""
    if not (_PERSISTENT and lib.mirl_lstm_seq_supported(2, B, H)):
        return True
    wg, lds, cus, per = C.c_int32(), C.c_int64(), C.c_int32(), C.c_int32()
    check(lib.mirl_lstm_seq_fwd_grid(B, H, C.byref(wg), C.byref(lds), C.byref(cus), C.byref(per)), "mirl_lstm_seq_fwd_grid")
    return 2 * wg.value <= cus.value * per.value


def _forward_sweep(gates, w, h0, c0, keep, need_grad):
    """The time loop over `gates` (T, B, 4H) = the input projection (+ biases), in place.
    Returns (out, hm, cm, c_all, h_last, c_last); hm / cm / c_all are None unless need_grad
    (then `gates` holds the activated gates on return)."""
    T, B, G = gates.shape
    H = G // 4
    dev = gates.device
    out = torch.empty((T, B, H), dtype=torch.float32, device=dev)
    st = _stream()
    if persistent_supported(T, B, H) and w.data_ptr() % 16 == 0:
        nbytes = C.c_int64()
        check(lib.mirl_lstm_seq_workspace_bytes(B, H, C.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        assert ws.data_ptr() % 256 == 0
        hm = cm = c_all = h_last = c_last = None
        if need_grad:
            hm = torch.empty((T + 1, B, H), dtype=torch.float32, device=dev)
            cm = torch.empty((T + 1, B, H), dtype=torch.float32, device=dev)
            c_all = torch.empty((T, B, H), dtype=torch.float32, device=dev)
        else:
            h_last = torch.empty((B, H), dtype=torch.float32, device=dev)
            c_last = torch.empty((B, H), dtype=torch.float32, device=dev)
        check(lib.mirl_lstm_seq_fwd(
            T, B, H, _p(gates), _p(w), _p(h0), _p(c0), _p(keep), _p(out), _p(c_all), _p(hm), _p(cm),
            _p(h_last), _p(c_last), 1 if need_grad else 0, _p(ws), st), "mirl_lstm_seq_fwd")
        ws.record_stream(torch.cuda.current_stream())
        if need_grad:
            h_last, c_last = hm[T], cm[T]
        return out, hm, cm, c_all, h_last, c_last
    hm = torch.empty((T + 1, B, H), dtype=torch.float32, device=dev)     # masked h inputs; hm[T] = final h
    cm = torch.empty((T + 1, B, H), dtype=torch.float32, device=dev)
    c_all = torch.empty((T, B, H), dtype=torch.float32, device=dev) if need_grad else None
    torch.mul(h0, keep[0].unsqueeze(-1), out=hm[0])
    torch.mul(c0, keep[0].unsqueeze(-1), out=cm[0])
    wt = w.t()
    for t in range(T):
        gates[t].addmm_(hm[t], wt)
        check(lib.mirl_lstm_cell_fwd(
            B, H, _p(gates[t]), _p(cm[t]), _p(keep[t + 1]) if t + 1 < T else None,
            _p(out[t]), _p(c_all[t]) if need_grad else None, _p(hm[t + 1]), _p(cm[t + 1]), st),
            "mirl_lstm_cell_fwd")
    return out, hm, cm, c_all, hm[T], cm[T]


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _backward_sweep(gates, c_all, cm, d_out, keep, w):
    """The backward time loop: `gates` (T, B, 4H) activated gates -> d loss / d pre-activation, in place.
    One persistent launch (csrc/lstm_seq.hip k_lstm_seq_bwd) where the shape is covered, else one
    cell kernel + one recurrent GEMM per step."""
    T, B, G = gates.shape
    H = G // 4
    st = _stream()
    if _PERSISTENT and T >= 2 and B <= _BWD_PERSISTENT_MAX_B and lib.mirl_lstm_seq_bwd_supported(T, B, H) \
            and w.data_ptr() % 16 == 0:
        nbytes = C.c_int64()
        check(lib.mirl_lstm_seq_bwd_workspace_bytes(B, H, C.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=gates.device)
        check(lib.mirl_lstm_seq_bwd(T, B, H, _p(gates), _p(w), _p(c_all), _p(cm), _p(d_out), _p(keep), _p(ws), st),
              "mirl_lstm_seq_bwd")
        ws.record_stream(torch.cuda.current_stream())
        return
    dh_rec = torch.empty((B, H), dtype=torch.float32, device=gates.device)
    dc_rec = torch.empty((B, H), dtype=torch.float32, device=gates.device)
    for t in range(T - 1, -1, -1):
        check(lib.mirl_lstm_cell_bwd(
            B, H, _p(gates[t]), _p(c_all[t]), _p(cm[t]), _p(d_out[t]), _p(dh_rec), _p(dc_rec),
            _p(keep[t + 1]) if t + 1 < T else None, 1 if t == T - 1 else 0, st),
            "mirl_lstm_cell_bwd")
        if t > 0:
            torch.mm(gates[t], w, out=dh_rec)


class _LSTMSequence(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, w_hh, bias, h0, c0, keep, track=True):
        # x (T*B, I); bias = b_ih + b_hh (4H); keep (T, B).  The input projection of
        # ALL timesteps is one GEMM into `gates`; every step then accumulates the
        # recurrent GEMM onto its slice IN PLACE (no per-step copy of the projection)
        # and the cell kernel turns the pre-activations into activated gates in place.
        T, B = keep.shape
        H = w_hh.shape[1]
        x = x.float()
        w = w_hh.float().contiguous()
        keep = keep.float().contiguous()
        # `track`: decided by the caller from torch.is_grad_enabled() — inside forward() grad mode is
        # always off and ctx.needs_input_grad mirrors requires_grad even in a no_grad pass
        need_grad = track and any(ctx.needs_input_grad[:4])
        dev = x.device
        gates = gemm3.linear_fwd(x, w_ih.float(), bias.float().contiguous()).view(T, B, 4 * H)
        out, hm, cm, c_all, h_last, c_last = _forward_sweep(
            gates, w, h0.float().contiguous(), c0.float().contiguous(), keep, need_grad)
        if need_grad:
            ctx.save_for_backward(x, w_ih, gates, c_all, cm, hm, keep, w)
        ctx.mark_non_differentiable(h_last, c_last)
        return out, h_last, c_last

    @staticmethod
    def backward(ctx, d_out, _dh, _dc):
        x, w_ih, gates, c_all, cm, hm, keep, w = ctx.saved_tensors
        T, B, G = gates.shape
        H = G // 4
        d_out = d_out.float().contiguous()
        _backward_sweep(gates, c_all, cm, d_out, keep, w)
        dg = gates.reshape(T * B, G)                      # now d loss / d pre-activation
        d_x = gemm3.grad_input(dg, w_ih.float()) if ctx.needs_input_grad[0] else None
        d_wih = gemm3.grad_weight(dg, x) if ctx.needs_input_grad[1] else None
        d_whh = gemm3.grad_weight(dg, hm[:T].reshape(T * B, H)) if ctx.needs_input_grad[2] else None
        d_b = dg.sum(0) if ctx.needs_input_grad[3] else None
        return d_x, d_wih, d_whh, d_b, None, None, None, None


class _LSTMSequenceFromProjection(torch.autograd.Function):
    """The same recurrence on an input projection computed elsewhere (gx (T, B, 4H) =
    x W_ih^T + b_ih + b_hh, possibly a slice of a larger shared block): gx is left
    untouched (it may have other consumers), so the gates buffer is a copy of it."""

    @staticmethod
    def forward(ctx, gx, w_hh, h0, c0, keep, track=True):
        T, B, G = gx.shape
        H = G // 4
        w = w_hh.float().contiguous()
        keep = keep.float().contiguous()
        need_grad = track and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        dev = gx.device
        if need_grad or not (persistent_supported(T, B, H) and w.data_ptr() % 16 == 0):
            gates = gx.float().clone(memory_format=torch.contiguous_format)
        else:
            gates = gx.float().contiguous()       # the persistent no-grad sweep only READS the projection: no copy
        out, hm, cm, c_all, h_last, c_last = _forward_sweep(
            gates, w, h0.float().contiguous(), c0.float().contiguous(), keep, need_grad)
        if need_grad:
            ctx.save_for_backward(gates, c_all, cm, hm, keep, w)
        ctx.mark_non_differentiable(h_last, c_last)
        return out, h_last, c_last

    @staticmethod
    def backward(ctx, d_out, _dh, _dc):
        gates, c_all, cm, hm, keep, w = ctx.saved_tensors
        T, B, G = gates.shape
        H = G // 4
        d_out = d_out.float().contiguous()
        _backward_sweep(gates, c_all, cm, d_out, keep, w)
        d_whh = gemm3.grad_weight(gates.reshape(T * B, G), hm[:T].reshape(T * B, H)) if ctx.needs_input_grad[1] else None
        return gates, d_whh, None, None, None, None


def lstm_sequence_from_projection(gx, w_hh, h0, c0, keep):
    """gx (T, B, 4H) -> (out (T,B,H), h_T, c_T)."""
    return _LSTMSequenceFromProjection.apply(gx, w_hh, h0, c0, keep, torch.is_grad_enabled())


def lstm_sequence(x, w_ih, w_hh, bias, h0, c0, keep):
    """x (T*B, I) -> (out (T,B,H), h_T (B,H), c_T (B,H)); keep (T, B) = 1 - initials."""
    return _LSTMSequence.apply(x, w_ih, w_hh, bias, h0, c0, keep, torch.is_grad_enabled())
