"""TorchTrainer (reference rltime/training/torch/torch_trainer.py:6-199):
optimizer, gradient-norm clipping, the n-step bootstrap target tail."""
import torch

from .multi_step_trainer import MultiStepTrainer
from rltime_amd.models.torch.utils import set_lr


class TorchTrainer(MultiStepTrainer):
    def _train(self, clip_grad=None, clip_grad_dynamic_alpha=None, adam_epsilon=1e-8,
               vf_scale_epsilon=None, apply_initial_lr=False, burn_in_full_forward=False,
               share_online_cnn=True, share_online_projection=True, selection_advantage_only=True, **kwargs):
        """torch_trainer.py:9-44.  apply_initial_lr=False mirrors the reference,
        whose train_init ignores `lr` (Adam starts at 1e-3, SURVEY A-14)."""
        self.clip_grad = float(clip_grad) if clip_grad is not None else None
        self.clip_grad_dynamic_alpha = clip_grad_dynamic_alpha
        self._grad_norm_moving_average = None
        self.adam_epsilon = adam_epsilon
        assert vf_scale_epsilon is None or vf_scale_epsilon > 0
        assert (not vf_scale_epsilon) or (not self.clip_rewards), \
            "Value function rescaling only makes sense with clip_rewards=False"
        self.vf_scale_epsilon = vf_scale_epsilon
        self._apply_initial_lr = apply_initial_lr
        self.burn_in_full_forward = burn_in_full_forward
        self.share_online_cnn = share_online_cnn
        self.share_online_projection = share_online_projection
        # double-Q action selection from the dueling head's advantage stream alone (policies/dqn.py predict_selection):
        # the same arg-max, half of the selection pass's widest GEMM
        self.selection_advantage_only = selection_advantage_only
        super()._train(**kwargs)

    def train_init(self, lr):
        """torch_trainer.py:80-83."""
        kw = {"lr": lr} if self._apply_initial_lr else {}
        self.optimizer = torch.optim.Adam(self.policy.parameters(), eps=self.adam_epsilon, **kw)

    def set_lr(self, lr):
        set_lr(self.optimizer, lr)

    def _compute_grads(self, states, targets, policy_outputs, extra_data, timesteps):
        raise NotImplementedError

    # -- the reference's target hooks (torch_trainer.py:91-147).  DQN / IQN override
    # calc_target_values with one fused kernel after their forwards; a subclass that
    # only supplies `_get_bootstrap_target_value` (the reference's plugin contract)
    # gets the generic composition below, same arithmetic in plain torch ops.
    def _get_bootstrap_target_value(self, target_states, timesteps):
        raise NotImplementedError

    def _discount_bootstrap_target_value(self, target_values, nsteps):
        return (self.gamma ** nsteps) * target_values

    def _vf_scale(self, x):
        """torch_trainer.py:46-52."""
        eps = self.vf_scale_epsilon
        if eps is None:
            return x
        return torch.sign(x) * (torch.sqrt(torch.abs(x) + 1) - 1) + eps * x

    def _vf_unscale(self, y):
        """torch_trainer.py:54-78 (float64 inside, float32 out)."""
        eps = self.vf_scale_epsilon
        if eps is None:
            return y
        y64 = y.double()
        a = torch.abs(y64)
        x = a / eps - (1 / (2. * eps ** 2)) * torch.sqrt(4 * eps * a + (2. * eps + 1) ** 2) + (2. * eps + 1) / (2. * eps ** 2)
        return (x * torch.sign(y64)).float()

    def calc_target_values(self, returns, target_states, target_masks, nsteps, timesteps):
        """torch_trainer.py:101-147."""
        with torch.no_grad():
            target_states, returns, target_masks, nsteps = self.target_policy.make_tensor(
                (target_states, returns, target_masks, nsteps), non_blocking=True)
            v = self._vf_unscale(self._get_bootstrap_target_value(target_states, timesteps))
            assert returns.shape == target_masks.shape == nsteps.shape == (v.shape[0],)
            assert v.dim() in (1, 2)
            if v.dim() == 2:
                returns, target_masks, nsteps = (t.unsqueeze(-1) for t in (returns, target_masks, nsteps))
            return self._vf_scale(returns + self._discount_bootstrap_target_value(v, nsteps) * target_masks)

    def _clip_value(self, norm):
        """torch_trainer.py:153-175: fixed clip or clip_grad x EMA(norm); the EMA
        lives on the device."""
        if not self.clip_grad:
            return None
        if self.clip_grad_dynamic_alpha is None:
            return self.clip_grad
        a = self.clip_grad_dynamic_alpha
        self._grad_norm_moving_average = norm.detach().clone() if self._grad_norm_moving_average is None \
            else self._grad_norm_moving_average * a + norm.detach() * (1 - a)
        self.value_log.log("grad_norm_ma", self._grad_norm_moving_average, group="train")
        return self._grad_norm_moving_average * self.clip_grad

    def train_batch(self, *args, **kwargs):
        """torch_trainer.py:177-199.  The norm is one fused device reduction and
        the clip a device-side scale: no host synchronisation per step (the
        reference does ~2 x #parameters `.item()` calls, torch_policy.py:70-78)."""
        dp = getattr(self, "data_parallel", None)
        if dp is not None and dp._flat is not None:
            dp.zero_grad()                  # .grad tensors are views of the all-reduce bucket
        else:
            self.optimizer.zero_grad(set_to_none=True)
        self._compute_grads(*args, **kwargs)
        self._reduce_gradients()
        params = [p for p in self.policy.parameters() if p.grad is not None]
        grads = [p.grad for p in params]
        norm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads, 2)), 2)
        self.value_log.log("grad_norm", norm, group="train")
        clip = self._clip_value(norm)
        if clip is not None:
            coef = torch.clamp(clip / (norm + 1e-6), max=1.0)     # torch.nn.utils.clip_grad_norm_
            torch._foreach_mul_(grads, coef)
            self.value_log.log("grad_norm_clipped", norm * coef, group="train")
        if self.policy.is_cuda():
            self._check_sweeps()
        self.optimizer.step()

    def _check_sweeps(self):
        """A persistent LSTM sweep that gave up (csrc/lstm_seq.hip: bounded spin, non-finite state) sets a pinned host
        word.  Reading it only means something at a point the host is SYNCHRONISED with the sweeps it covers — the host
        enqueues about a step ahead of the GPU.  Default: wait for the event recorded after the PREVIOUS step's backward
        (the GPU still holds a whole step of queued work, so it never idles), then read: a failed step k raises before
        step k+1's optimizer is enqueued and before anything of it is logged or checkpointed (policy_trainer.py checks
        again behind its own synchronisation).  MIRL_STRICT_SWEEP_CHECK=1 waits for THIS step's backward instead — no
        invalid gradient can reach the optimizer at all, at the price of a drained launch queue per step."""
        import os
        from rltime_amd.models.torch import lstm_seq
        if os.environ.get("MIRL_STRICT_SWEEP_CHECK", "0") == "1":
            torch.cuda.current_stream().synchronize()
            lstm_seq.check_status()
            return
        prev = getattr(self, "_backward_done", None)
        if prev is not None:
            prev.synchronize()
            lstm_seq.check_status()
        self._backward_done = torch.cuda.Event()
        self._backward_done.record()

    def _reduce_gradients(self):
        """Data-parallel hook: all-reduce the gradients across ranks (RCCL) when
        a process group is attached (rltime_amd.parallel); no-op on 1 GPU."""
        dp = getattr(self, "data_parallel", None)
        if dp is not None:
            dp.all_reduce_gradients(self.policy)
