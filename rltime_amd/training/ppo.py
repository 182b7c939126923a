"""PPO trainer (reference rltime/training/torch/ppo.py:7-56): A2C with the clipped surrogate action gain."""
import torch

from .a2c import A2C
from rltime_amd.general.utils import anneal_value


class PPO(A2C):
    def _train(self, clip_value, clip_anneal=None, adv_norm=True, **kwargs):
        """ppo.py:19-31: advantages are normalised by default."""
        self._clip_value, self._clip_anneal = clip_value, clip_anneal
        super()._train(adv_norm=adv_norm, **kwargs)

    def _calc_clip_value(self):
        return anneal_value(self._clip_value, self.get_train_progress(), self._clip_anneal)

    def _calc_action_gain(self, action_log_probs, advantages, org_policy_outputs):
        """ppo.py:39-56: min(ratio A, clip(ratio, 1 - c, 1 + c) A) against the acting-time log-probabilities."""
        old = self.policy.make_tensor(org_policy_outputs["action_log_probs"])
        assert action_log_probs.shape == old.shape == advantages.shape
        ratio = torch.exp(action_log_probs - old)
        c = self._calc_clip_value()
        return torch.min(ratio * advantages, torch.clamp(ratio, 1.0 - c, 1.0 + c) * advantages).mean()
