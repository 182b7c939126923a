"""Oracle: Q-learning target / loss arithmetic (TEST INFRASTRUCTURE ONLY).

torch-CPU fp32 restatement (this is floating-point work, so a torch fp32
reference is the right checker) of

  * rltime/training/torch/torch_trainer.py:46-78,96-147  (value rescaling h / h^-1,
    n-step bootstrap target)
  * rltime/training/torch/dqn.py:52-71,98-130,141-161    (double-Q select, Huber/MSE,
    importance weights, aggregation)
  * rltime/training/torch/iqn.py:36-52,77-120            (IQN target select, pairwise
    quantile-Huber loss, |td| report)

All functions are pure: network outputs come in as tensors (the golden
generator obtains the same purity from the reference with stub policies).
Tolerance against the HIP kernels: 1e-4 absolute/relative fp32 (BASELINE.json
north_star), stated again in the tests.
"""
import torch


def vf_scale(x, eps):
    """torch_trainer.py:46-52: h(x) = sign(x)(sqrt(|x|+1)-1) + eps*x."""
    if not eps:
        return x
    return torch.sign(x) * (torch.sqrt(torch.abs(x) + 1) - 1) + eps * x


def vf_unscale(y, eps):
    """torch_trainer.py:54-78: closed-form h^-1 evaluated in float64, returned
    as float32."""
    if not eps:
        return y
    y = y.double()
    a = torch.abs(y)
    x = a / eps - (1 / (2. * eps ** 2)) * torch.sqrt(
        4 * eps * a + (2. * eps + 1) ** 2) + (2. * eps + 1) / (2. * eps ** 2)
    x = x * torch.sign(y)
    return x.float()


def nstep_target(bootstrap, returns, masks, nsteps, gamma, vf_eps):
    """torch_trainer.py:124-147: y = h(ret + gamma^n * h^-1(v) * mask), with
    returns/masks/nsteps (M,) fp32 broadcast over an optional quantile dim."""
    v = vf_unscale(bootstrap, vf_eps)
    if v.dim() == 2:
        returns, masks, nsteps = (
            t.unsqueeze(-1) for t in (returns, masks, nsteps))
    return vf_scale(returns + (gamma ** nsteps) * v * masks, vf_eps)


def dqn_bootstrap(q_target, q_select):
    """dqn.py:52-71: v = q_target[argmax_a q_select]."""
    best = q_select.argmax(dim=-1, keepdim=True)
    return q_target.gather(-1, best).squeeze(-1)


def iqn_bootstrap(z_target, z_select):
    """iqn.py:36-52: a* = argmax_a mean_N z_select; v = z_target[:, :, a*]."""
    best = z_select.mean(1).argmax(dim=-1, keepdim=True)       # (M,1)
    best = best.unsqueeze(1).repeat(1, z_target.shape[1], 1)   # (M,N',1)
    return torch.gather(z_target, -1, best).squeeze(-1)        # (M,N')


def huber(err, kappa):
    """dqn.py:105-111."""
    a = torch.abs(err)
    return torch.where(a <= kappa, 0.5 * err.pow(2), kappa * (a - 0.5 * kappa))


def aggregate(rows, timesteps, batch_mode="mean", time_mode=None):
    """dqn.py:116-130 (_aggregate_losses)."""
    pick = {"mean": torch.mean, "sum": torch.sum}
    if time_mode:
        rows = pick[time_mode](rows.view(timesteps, -1), dim=0)
    return pick[batch_mode](rows)


def dqn_loss(q, actions, targets, weights=None, kappa=1.0, mode="huber",
             timesteps=1, batch_mode="mean", time_mode=None):
    """dqn.py:141-161.  Returns (scalar loss, signed td report (M,))."""
    chosen = torch.gather(q, -1, actions.long().unsqueeze(-1)).squeeze(-1)
    td = chosen - targets
    rows = td.pow(2) if mode == "mse" else huber(td, kappa)
    if weights is not None:
        rows = rows * weights
    return aggregate(rows, timesteps, batch_mode, time_mode), td.detach()


def iqn_loss(z, taus, actions, targets, weights=None, kappa=1.0,
             timesteps=1, batch_mode="mean", time_mode=None):
    """iqn.py:77-120.  z (M,N,A), taus (M*N,) or (M,N), targets (M,N').
    Returns (scalar loss, mean |td| report (M,))."""
    M, N, _ = z.shape
    idx = actions.long().view(M, 1, 1).repeat(1, N, 1)
    theta = torch.gather(z, -1, idx).squeeze(-1)               # (M,N)
    td = targets.unsqueeze(2) - theta.unsqueeze(1)             # (M,N',N)
    h = huber(td, kappa)
    tau = taus.view(M, N).unsqueeze(1).repeat(1, targets.shape[1], 1)
    under = (td < 0).float().detach()                          # iqn.py:98
    rows = (torch.abs(tau - under) * h / kappa).sum(2).mean(1)  # iqn.py:100-104
    report = td.abs().mean(1).mean(1)                          # iqn.py:112
    if weights is not None:
        rows = rows * weights
    return aggregate(rows, timesteps, batch_mode, time_mode), report.detach()


def gae_bootstrap_discount(values, nsteps, gamma, lam):
    """a2c.py:68-71."""
    return (gamma ** nsteps) * (lam ** (nsteps - 1)) * values


def actor_critic_loss(log_probs, values, entropy, targets, acting_values, old_log_probs, vf_coef, entropy_factor,
                      adv_norm, clip_value=None):
    """a2c.py:101-133 with the action gain of a2c.py:90-99 (clip_value None) or ppo.py:39-56.
    -> (total loss, value loss, action gain)."""
    adv = targets - acting_values
    if adv_norm:
        adv = (adv - adv.mean()) / (adv.std() + 1e-5)
    if clip_value is None:
        gain = (log_probs * adv).mean()
    else:
        ratio = torch.exp(log_probs - old_log_probs)
        gain = torch.min(ratio * adv, torch.clamp(ratio, 1.0 - clip_value, 1.0 + clip_value) * adv).mean()
    value_loss = (targets - values).pow(2).mean()
    return value_loss * vf_coef - gain - entropy_factor * entropy.mean(), value_loss, gain
