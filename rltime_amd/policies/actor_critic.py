"""ActorCriticPolicy and its action-distribution heads (reference
rltime/policies/torch/actor_critic.py:9-107, policies/torch/distributions/{categorical,normal}.py).

Used by the A2C / PPO trainers of the CPU plumbing config (BASELINE configs[0],
`cartpole_ppo.json`); plain PyTorch, not on the MI355X hot path."""
import numpy as np
import torch
import torch.nn as nn

from .torch_policy import TorchPolicy
from rltime_amd.models.torch.utils import linear
from rltime_amd.spaces import is_tuple_space


class CategoricalHead(nn.Module):
    """distributions/categorical.py:8-20: one action out of action_space.n."""

    def __init__(self, action_space, input_size):
        super().__init__()
        self.logits_layer = linear(input_size, action_space.n)

    def forward(self, x):
        return torch.distributions.Categorical(logits=self.logits_layer(x))


class NormalHead(nn.Module):
    """distributions/normal.py:9-36: independent Gaussians over a Box action, state-independent log-std."""

    def __init__(self, action_space, input_size):
        super().__init__()
        self.actions_shape = tuple(action_space.shape)
        flat = int(np.prod(self.actions_shape))
        self.mean_layer = linear(input_size, flat)
        self.logstd = nn.Parameter(torch.zeros(flat))

    def forward(self, x):
        mean = self.mean_layer(x)
        std = self.logstd.exp().expand_as(mean)
        shape = (x.shape[0],) + self.actions_shape
        return torch.distributions.Normal(mean.view(shape), std.view(shape))


def distribution_head(action_space):
    """distributions/__init__.py:6-16 by duck type (gym is not a dependency here): `.n` = Discrete, a shaped box otherwise."""
    if hasattr(action_space, "n"):
        return CategoricalHead
    if not is_tuple_space(action_space) and len(getattr(action_space, "shape", ())) > 0:
        return NormalHead
    raise ValueError("Unsupported action_space, there is no distribution class available for: %s" % type(action_space))


class ActorCriticPolicy(TorchPolicy):
    def __init__(self, model_config, observation_space, action_space, critic_separate_model=False):
        """actor_critic.py:10-35."""
        super().__init__(model_config, observation_space)
        self.value_model = self._create_model_from_config(model_config, observation_space) if critic_separate_model else None
        self.actor = distribution_head(action_space)(action_space, self.model.out_size)
        self.critic = linear(self.model.out_size, 1)

    def get_dist_and_state_value(self, x, timesteps):
        """actor_critic.py:95-107."""
        features = self.model(x, timesteps)["output"]
        dist = self.actor(features)
        if self.value_model is not None:
            features = self.value_model(x, timesteps)["output"]
        return dist, self.critic(features).squeeze(-1)

    def get_state_value(self, inp, timesteps):
        """actor_critic.py:73-80."""
        model = self.value_model if self.value_model is not None else self.model
        return self.critic(model(inp, timesteps)["output"]).squeeze(-1)

    @staticmethod
    def _joint(log_probs):
        # a multi-dimensional action's log-probability is the sum over its components (actor_critic.py:57-58, :88-89)
        return log_probs.sum(-1) if log_probs.dim() == 2 else log_probs

    def actor_predict(self, inp, timesteps, force_best=False):
        """actor_critic.py:37-71: sampled (or arg-max) actions with the log-probabilities and value estimates
        the trainer needs later."""
        with torch.no_grad():
            dist, values = self.get_dist_and_state_value(inp, timesteps)
            actions = dist.logits.argmax(dim=-1) if force_best else dist.sample()
            log_probs = self._joint(dist.log_prob(actions))
        return {"actions": actions.cpu().numpy(), "action_log_probs": log_probs.cpu().numpy(),
                "values": values.cpu().numpy()}

    def evaluate_actions(self, inp, timesteps, actions):
        """actor_critic.py:82-93 -> (log-probabilities of `actions`, state values, entropy)."""
        dist, values = self.get_dist_and_state_value(inp, timesteps)
        return self._joint(dist.log_prob(self.make_tensor(actions))), values, dist.entropy()
