"""Fused Q-learning target / loss ops (librltime_hip qmath kernels) as torch
functions.  No torch fallback: inputs must be CUDA tensors.

Reference arithmetic restated by the kernels:
  torch_trainer.py:46-78,124-147   value rescaling + n-step bootstrap target
  training/torch/dqn.py:52-71      DQN / double-Q bootstrap value
  training/torch/iqn.py:36-52      IQN bootstrap value
  training/torch/dqn.py:98-130,141-161   DQN loss, IS weights, aggregation
  training/torch/iqn.py:77-120     IQN pairwise quantile-Huber loss
"""
import ctypes as C

import torch

from .._lib import lib, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _f32(t):
    assert t.is_cuda, "rltime_amd.qops needs CUDA tensors (no CPU fallback)"
    return t.detach().to(torch.float32).contiguous()


def row_scale(rows, timesteps, batch_mode="mean", time_mode=None):
    """d(loss)/d(row loss) of dqn.py:120-130 (_aggregate_losses): every
    combination of mean/sum over time then batch is one constant factor."""
    s = 1.0
    count = rows
    if time_mode:
        if time_mode == "mean":
            s /= timesteps
        count = rows // timesteps
    if batch_mode == "mean":
        s /= count
    return s


def q_target_dqn(q_target, q_select, returns, nsteps, masks, gamma, vf_eps=None):
    q_target, q_select = _f32(q_target), _f32(q_select)
    M, A = q_target.shape
    out = torch.empty(M, dtype=torch.float32, device=q_target.device)
    check(lib.mirl_q_target_dqn(
        M, A, _p(q_target), _p(q_select), _p(_f32(returns)), _p(_f32(nsteps)),
        _p(_f32(masks)), float(gamma), float(vf_eps or 0.0), _p(out), _stream()),
        "mirl_q_target_dqn")
    return out


def q_target_iqn(z_target, z_select, returns, nsteps, masks, gamma, vf_eps=None):
    z_target, z_select = _f32(z_target), _f32(z_select)
    M, Nt, A = z_target.shape
    Ns = z_select.shape[1]
    out = torch.empty((M, Nt), dtype=torch.float32, device=z_target.device)
    check(lib.mirl_q_target_iqn(
        M, Nt, Ns, A, _p(z_target), _p(z_select), _p(_f32(returns)),
        _p(_f32(nsteps)), _p(_f32(masks)), float(gamma), float(vf_eps or 0.0),
        _p(out), _stream()), "mirl_q_target_iqn")
    return out


class _DQNLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, actions, targets, weights, kappa, mode, scale):
        qc = _f32(q)
        M, A = qc.shape
        rows = torch.empty(M, dtype=torch.float32, device=qc.device)
        td = torch.empty(M, dtype=torch.float32, device=qc.device)
        dq = torch.empty_like(qc)
        w = _f32(weights) if weights is not None else None
        check(lib.mirl_loss_dqn(
            M, A, _p(qc), _p(actions.to(torch.int64).contiguous()), _p(_f32(targets)),
            _p(w), float(kappa), 1 if mode == "mse" else 0, float(scale),
            _p(rows), _p(dq), _p(td), _stream()), "mirl_loss_dqn")
        ctx.save_for_backward(dq)
        ctx.mark_non_differentiable(td)
        return rows.sum() * scale, td

    @staticmethod
    def backward(ctx, g_loss, g_td):
        (dq,) = ctx.saved_tensors
        return dq * g_loss, None, None, None, None, None, None


class _IQNLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, taus, actions, targets, weights, kappa, scale):
        zc = _f32(z)
        M, N, A = zc.shape
        Nt = targets.shape[1]
        rows = torch.empty(M, dtype=torch.float32, device=zc.device)
        rep = torch.empty(M, dtype=torch.float32, device=zc.device)
        dz = torch.empty_like(zc)
        w = _f32(weights) if weights is not None else None
        check(lib.mirl_loss_iqn(
            M, N, Nt, A, _p(zc), _p(_f32(taus).reshape(M, N)),
            _p(actions.to(torch.int64).contiguous()), _p(_f32(targets)), _p(w),
            float(kappa), float(scale), _p(rows), _p(dz), _p(rep), _stream()),
            "mirl_loss_iqn")
        ctx.save_for_backward(dz)
        ctx.mark_non_differentiable(rep)
        return rows.sum() * scale, rep

    @staticmethod
    def backward(ctx, g_loss, g_rep):
        (dz,) = ctx.saved_tensors
        return dz * g_loss, None, None, None, None, None, None


def dqn_loss(q, actions, targets, weights=None, kappa=1.0, mode="huber",
             timesteps=1, batch_mode="mean", time_mode=None):
    """-> (scalar loss differentiable w.r.t. q, signed td report (M,))."""
    scale = row_scale(q.shape[0], timesteps, batch_mode, time_mode)
    return _DQNLoss.apply(q, actions, targets, weights, kappa, mode, scale)


def iqn_loss(z, taus, actions, targets, weights=None, kappa=1.0,
             timesteps=1, batch_mode="mean", time_mode=None):
    """-> (scalar loss differentiable w.r.t. z, mean |td| report (M,))."""
    scale = row_scale(z.shape[0], timesteps, batch_mode, time_mode)
    return _IQNLoss.apply(z, taus, actions, targets, weights, kappa, scale)
