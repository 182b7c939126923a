"""IQN trainer (reference rltime/training/torch/iqn.py:8-129)."""
import torch

from .dqn import DQN
from . import qops
from rltime_amd.policies.iqn import IQNPolicy


class IQN(DQN):
    @staticmethod
    def create_policy(**kwargs):
        return IQNPolicy.create(**kwargs)

    def calc_target_values(self, returns, target_states, target_masks, nsteps, timesteps):
        """iqn.py:15-52 + torch_trainer.py:124-147.  Two forwards in the
        reference's order (target net, then the selection net with independent
        taus) and one fused kernel for mean-argmax / gather / rescale."""
        with torch.no_grad():
            sel = self.policy if self.double_q else self.target_policy
            # the selection pass only feeds argmax_a mean_N Z: with a dueling head that is the advantage stream's
            # arg-max (DQNPolicy.predict_selection) — the value-hidden half of the head's widest GEMM is skipped
            fwd = (getattr(sel, "predict_selection", None) if getattr(self, "selection_advantage_only", True) else None) or sel.predict
            if sel is not self.target_policy and self._passes_overlap(
                    returns.shape[0], returns.shape[0] // int(timesteps) if int(timesteps) >= 2 else None):
                # two different networks on the same states: the target pass on the second stream (multi_step_trainer.py
                # _side_by_side); the target net's fractions are drawn first, as in the reference
                def no_grad(f):
                    with torch.no_grad():
                        return f(target_states, timesteps=timesteps)[0]
                z_t, z_s = self._side_by_side(lambda: no_grad(self.target_policy.predict), lambda: no_grad(fwd))
            else:
                z_t = self.target_policy.predict(target_states, timesteps=timesteps)[0]
                z_s = fwd(target_states, timesteps=timesteps)[0]
            mk = self.policy.make_tensor
            return qops.q_target_iqn(z_t, z_s, mk(returns), mk(nsteps), mk(target_masks),
                                     self.gamma, self.vf_scale_epsilon)

    def _get_bootstrap_target_value(self, target_states, timesteps):
        """iqn.py:15-52 alone (see DQN._get_bootstrap_target_value)."""
        z_t = self.target_policy.predict(target_states, timesteps=timesteps)[0]
        sel = self.policy if self.double_q else self.target_policy
        z_s = sel.predict(target_states, timesteps=timesteps)[0]
        z = torch.zeros(z_t.shape[0], device=z_t.device)
        return qops.q_target_iqn(z_t, z_s, z, z, torch.ones_like(z), self.gamma, None)

    def _compute_grads(self, states, targets, policy_outputs, extra_data, timesteps):
        """iqn.py:54-129."""
        assert self.loss_mode == "huber", "IQN supports only huber loss"
        actions = self.policy.make_tensor(policy_outputs["actions"]).long()
        z, taus = self.policy.predict(states, timesteps)
        loss, report = qops.iqn_loss(z, taus, actions, targets, self._weights(extra_data),
                                     self.huber_kappa, timesteps, self.loss_aggregation,
                                     self.loss_timestep_aggregation)
        loss.backward()
        self._report_losses_if_needed(report, extra_data)
        self.value_log.log("qloss", loss.detach(), group="train")
        self.value_log.log("td_mean", report.mean(), group="train")
