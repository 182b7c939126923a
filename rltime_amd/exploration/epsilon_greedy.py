"""Epsilon-greedy exploration (reference rltime/exploration/{base,epsilon_greedy}.py):
linear decay of epsilon over `exploration_fraction` of training, optional list of
final values with probabilities (A3C style) and Ape-X per-actor exponent.

`remap_actions` keeps the reference's np.random consumption order
(epsilon_greedy.py:67,92,96); `remap_actions_device` is the batched device
variant used by the device-resident actor (one torch.rand / randint pair)."""
import numpy as np
import torch


class EpsilonGreedyExplorationManager:
    def __init__(self, eps_final, exploration_fraction, eps_prob=1.0, eps_start=1.0, eps_min=0.,
                 per_actor_exponent_factor=0, total_actors=None, **kwargs):
        self.total_actors = total_actors
        self.eps_start = eps_start
        self.eps_probs = eps_prob if isinstance(eps_prob, list) else [eps_prob]
        self.eps_finals = eps_final if isinstance(eps_final, list) else [eps_final]
        self.exploration_fraction = exploration_fraction
        self.eps_min = eps_min
        self.per_actor_exponent_factor = per_actor_exponent_factor
        if per_actor_exponent_factor > 0:
            assert total_actors is not None and total_actors > 1, \
                "per_actor_exponent_factor requires total_actors > 1"
        assert exploration_fraction <= 1.0

    def _get_eps(self, progress):
        """epsilon_greedy.py:64-72."""
        index = np.random.choice(range(len(self.eps_probs)), p=self.eps_probs)
        floor = self.eps_finals[index]
        if progress >= self.exploration_fraction:
            return floor
        return self.eps_start - (progress / self.exploration_fraction) * (self.eps_start - floor)

    def _actor_eps(self, eps, actor_index):
        if self.per_actor_exponent_factor:
            assert actor_index < self.total_actors
            eps = eps ** (1 + (actor_index / (self.total_actors - 1)) * self.per_actor_exponent_factor)
        return max(eps, self.eps_min)

    def remap_actions(self, actions, actor_indices, action_space, progress):
        """epsilon_greedy.py:74-100."""
        eps = self._get_eps(progress)
        used = []
        for i in range(len(actions)):
            e = self._actor_eps(eps, actor_indices[i] if self.per_actor_exponent_factor else 0)
            if np.random.rand() < e:
                actions[i] = np.random.randint(0, action_space.n)
            used.append(e)
        return actions, {"eps": np.array(used)}

    def _device_exponents(self, actor_indices, device):
        key = (device, len(actor_indices))
        cache = getattr(self, "_dev_cache", None)
        if cache is None or cache[0] != key:
            idx = torch.as_tensor([float(a) for a in actor_indices], dtype=torch.float64, device=device)
            if self.per_actor_exponent_factor:
                assert int(idx.max().item()) < self.total_actors
                expo = 1 + (idx / (self.total_actors - 1)) * self.per_actor_exponent_factor
            else:
                expo = torch.ones_like(idx)
            cache = self._dev_cache = (key, expo)
        return cache[1]

    def remap_with_eps_tensor(self, actions, eps, actor_indices, action_space, generator=None):
        """Device half of the remapping for a base epsilon held in a 0-dim float64
        tensor (so the whole thing can live inside a captured HIP graph): per-actor
        epsilons from one tensor pow, one rand / randint pair."""
        per = torch.clamp(torch.pow(eps, self._device_exponents(actor_indices, actions.device)),
                          min=self.eps_min).float()
        explore = torch.rand(actions.shape[0], device=actions.device, generator=generator) < per
        rnd = torch.randint(0, action_space.n, actions.shape, device=actions.device, generator=generator)
        return torch.where(explore, rnd.to(actions.dtype), actions), {"eps": per}

    def remap_actions_device(self, actions, actor_indices, action_space, progress, generator=None):
        """Batched device form of remap_actions (same epsilon schedule, one
        np.random draw for the final-epsilon pick like the reference)."""
        eps = torch.as_tensor(self._get_eps(progress), dtype=torch.float64, device=actions.device)
        return self.remap_with_eps_tensor(actions, eps, actor_indices, action_space, generator)
