"""A2C trainer with GAE returns (reference rltime/training/torch/a2c.py:6-141).  CPU plumbing config only."""
from .torch_trainer import TorchTrainer
from rltime_amd.general.utils import anneal_value
from rltime_amd.policies.actor_critic import ActorCriticPolicy


class A2C(TorchTrainer):
    def _train(self, entropy_factor, entropy_anneal=None, vf_coef=1.0, advlam=1.0, adv_norm=False,
               history_mode={"type": "online"}, **kwargs):
        """a2c.py:21-46: on-policy history by default."""
        self.entropy_factor, self.entropy_anneal = entropy_factor, entropy_anneal
        self.vf_coef, self.advlam, self.adv_norm = vf_coef, advlam, adv_norm
        super()._train(history_mode=history_mode, **kwargs)

    @staticmethod
    def create_policy(**kwargs):
        return ActorCriticPolicy.create(**kwargs)

    def _get_discount_function(self, gamma):
        """a2c.py:48-66: the k-th term of a truncated GAE(lambda) return, from the reward and the acting-time value
        estimate of step k; lambda = 1 gives gamma^k r.  (The bootstrap and the -V(s_t) are added by the trainer.)"""
        lam = self.advlam

        def discount(nstep, reward, policy_output):
            v = policy_output["values"]
            return (gamma ** nstep) * (lam ** (nstep - 1)) * (v + lam * (reward - v))
        return discount

    def _discount_bootstrap_target_value(self, target_values, nsteps):
        """a2c.py:68-71."""
        return (self.gamma ** nsteps) * (self.advlam ** (nsteps - 1)) * target_values

    def _get_bootstrap_target_value(self, target_states, timesteps):
        """a2c.py:73-82: V(target state) from the target policy (normally the online one)."""
        return self.target_policy.get_state_value(target_states, timesteps=timesteps)

    def _calc_entropy_factor(self):
        return anneal_value(self.entropy_factor, self.get_train_progress(), self.entropy_anneal)

    def _calc_action_gain(self, action_log_probs, advantages, org_policy_outputs):
        """a2c.py:90-99: vanilla advantage actor-critic."""
        assert action_log_probs.shape == advantages.shape
        return (action_log_probs * advantages).mean()

    def _compute_grads(self, states, targets, policy_outputs, extra_data, timesteps):
        """a2c.py:101-141: value MSE * vf_coef - action gain - entropy bonus; advantages against the ACTING-time
        value estimates, optionally normalised."""
        log_probs, values, entropy = self.policy.evaluate_actions(states, timesteps, policy_outputs["actions"])
        entropy = entropy.mean()
        assert log_probs.dim() == 1 and targets.dim() == 1
        baseline = self.policy.make_tensor(policy_outputs["values"])
        assert baseline.shape == targets.shape == values.shape
        advantages = targets - baseline
        if self.adv_norm:
            advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-5)
        gain = self._calc_action_gain(log_probs, advantages, policy_outputs)
        value_loss = (targets - values).pow(2).mean()
        loss = value_loss * self.vf_coef - gain - self._calc_entropy_factor() * entropy
        loss.backward()
        log = self.value_log.log
        log("state_value_mean", values.mean().item(), group="train")
        log("value_loss", value_loss.item(), group="train")
        log("policy_loss", -gain.item(), group="train")
        log("policy_entropy", entropy.item(), group="train")
