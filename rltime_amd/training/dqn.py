"""DQN trainer (reference rltime/training/torch/dqn.py:8-172) with the target
and loss arithmetic on the fused HIP kernels (rltime_amd/csrc/qmath.hip)."""
import torch

from .torch_trainer import TorchTrainer
from . import qops
from rltime_amd.policies.dqn import DQNPolicy


class DQN(TorchTrainer):
    def _train(self, double_q=False, loss_mode="huber", huber_kappa=1.0, loss_aggregation="mean",
               loss_timestep_aggregation=None, history_mode={"type": "replay"}, **kwargs):
        """dqn.py:15-46."""
        assert loss_aggregation in ("mean", "sum")
        assert loss_timestep_aggregation in (None, False, "", "mean", "sum")
        assert loss_mode in ("huber", "mse"), "%s is not a valid q-learning loss mode" % loss_mode
        self.double_q = double_q
        self.loss_mode = loss_mode
        self.huber_kappa = huber_kappa
        self.loss_aggregation = loss_aggregation
        self.loss_timestep_aggregation = loss_timestep_aggregation or None
        super()._train(history_mode=history_mode, **kwargs)

    @staticmethod
    def create_policy(**kwargs):
        return DQNPolicy.create(**kwargs)

    def calc_target_values(self, returns, target_states, target_masks, nsteps, timesteps):
        """torch_trainer.py:101-147 + dqn.py:52-71, one kernel after the
        forward(s): argmax (double-Q or not), gather, h^-1, discount, mask, h."""
        with torch.no_grad():
            q_t = self.target_policy.predict(target_states, timesteps=timesteps)
            fwd = (getattr(self.policy, "predict_selection", None) if getattr(self, "selection_advantage_only", True) else None) \
                or self.policy.predict
            q_s = q_t if not self.double_q else fwd(target_states, timesteps=timesteps)
            mk = self.policy.make_tensor
            return qops.q_target_dqn(q_t, q_s, mk(returns), mk(nsteps), mk(target_masks),
                                     self.gamma, self.vf_scale_epsilon)

    def _get_bootstrap_target_value(self, target_states, timesteps):
        """dqn.py:52-71 alone (the hook of the reference's TorchTrainer contract; the
        fused calc_target_values above does not go through it): the same kernel with a
        zero return, unit mask and gamma**0."""
        q_t = self.target_policy.predict(target_states, timesteps=timesteps)
        q_s = q_t if not self.double_q else self.policy.predict(target_states, timesteps=timesteps)
        z = torch.zeros(q_t.shape[0], device=q_t.device)
        return qops.q_target_dqn(q_t, q_s, z, z, torch.ones_like(z), self.gamma, None)

    def _report_losses_if_needed(self, losses, extra_train_data):
        """dqn.py:73-81 — the per-transition errors go to the replay as a device
        tensor (the reference copies them to the host first)."""
        if "loss_indices" not in extra_train_data:
            return
        idx = extra_train_data["loss_indices"]
        assert losses.shape == idx.shape[:1]
        if getattr(self, "_defer_losses", None) is not None:
            # a learner step being captured (torch_trainer._learner_step_graphed): the priority update follows each replay
            self._defer_losses.append((idx, losses))
            return
        self._pre_update_losses()
        self.history_buffer.update_losses(idx, losses)

    def _weights(self, extra_data):
        if "importance_weights" not in extra_data:
            return None
        w = self.policy.make_tensor(extra_data["importance_weights"])
        dp = getattr(self, "data_parallel", None)
        last = getattr(self.history_buffer, "last_sample", None)
        if dp is not None and last is not None and "stats" in last and not last.get("global"):
            # shard-local -> global importance weights (rltime_amd/parallel.py)
            w = dp.globalize_weights(w, last["stats"][0], self.history_buffer.stats()["active_sequences"],
                                     last["stats"][1], self.history_buffer.last_beta)
        self.value_log.log("importance_weights", w.mean(), group="train")
        return w

    def _compute_grads(self, states, targets, policy_outputs, extra_data, timesteps):
        """dqn.py:132-172."""
        q = self.policy.predict(states, timesteps)
        actions = self.policy.make_tensor(policy_outputs["actions"]).long()
        assert actions.shape == q.shape[:-1] and targets.shape == actions.shape
        loss, td = qops.dqn_loss(q, actions, targets, self._weights(extra_data), self.huber_kappa,
                                 self.loss_mode, timesteps, self.loss_aggregation,
                                 self.loss_timestep_aggregation)
        loss.backward()
        self._report_losses_if_needed(td, extra_data)
        self.value_log.log("qloss", loss.detach(), group="train")
