"""Synchronous vectorised actor (reference rltime/acting/actor.py:10-149).

Two output modes:
  * device=False: the reference behaviour — per-env sample dicts (numpy);
  * device=True : DeviceSamples — policy forward, epsilon-greedy, env step and
    replay ingest all stay on the GPU with no host round trip per step
    (SURVEY.md section 8(f) item 1)."""
import numpy as np
import torch

from .acting_interface import ActingInterface, DeviceSamples
from rltime_amd.general.type_registry import get_registered_type
from rltime_amd.general.utils import deep_apply


class GraphedPredict:
    """The acting forward (policy.actor_predict on the E-env batch) captured once
    in a HIP graph and replayed per vector step: at E=256 the eager forward is
    ~30 launch-bound kernels; the replay is one graph launch.  Inputs are copied
    into static buffers; the recurrent layers' `last_state` tensors are part of
    the captured outputs, so `make_input_state` keeps working unchanged."""

    def __init__(self, policy, example_state):
        self.policy = policy
        self.static_in = deep_apply(example_state, lambda t: t.clone())
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):          # warm-up outside capture (MIOpen find, hipBLASLt workspaces)
            for _ in range(3):
                policy.actor_predict(self.static_in, timesteps=1, as_numpy=False)
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = policy.actor_predict(self.static_in, timesteps=1, as_numpy=False)
        # recurrent layers publish their running state through `last_state`; the
        # captured tensors are refreshed by every replay, but the attribute is
        # re-bound by get_state() and by the learner's own forwards -> re-attach
        self.static_last = [(layer, layer.last_state) for layer in policy.model.layers
                            if getattr(layer, "last_state", None) is not None]

    def __call__(self, state):
        src, dst = [], []
        deep_apply(state, lambda t: src.append(t))
        deep_apply(self.static_in, lambda t: dst.append(t))
        torch._foreach_copy_(dst, src)
        self.graph.replay()
        for layer, state in self.static_last:
            layer.last_state = state
        return {k: v.clone() for k, v in self.static_out.items()}


class Actor(ActingInterface):
    def __init__(self, vec_env, exploration_config=None, base_env_id=0, total_env_ids=None, device=False,
                 use_graph=False):
        self._vec_env = vec_env
        self._num_envs = vec_env.num_envs
        self._base_env_id = base_env_id
        self._env_ids = list(range(base_env_id, base_env_id + self._num_envs))
        total = total_env_ids or self._num_envs
        if exploration_config:
            cls = get_registered_type("exploration", exploration_config["type"])
            self._exploration = cls(**exploration_config.get("args"), total_actors=total)
        else:
            self._exploration = None
        self._policy = None
        self._progress = 0
        self._device_mode = device
        self._use_graph = use_graph and device
        self._graphed = None
        super().__init__(vec_env.observation_space, vec_env.action_space)

    def get_env_count(self):
        return self._num_envs

    def update_state(self, progress, policy_state=None):
        self._progress = progress
        if policy_state is not None:
            self._policy.load_state(policy_state)

    def set_actor_policy(self, actor_policy):
        """actor.py:78-89: reset the envs and build the first input state."""
        self._policy = actor_policy
        obs = self._vec_env.reset()
        self.last_state = actor_policy.make_input_state(obs, np.array([True] * self._num_envs))

    def close(self):
        self._vec_env.close()

    def get_samples(self, min_samples):
        """actor.py:97-149."""
        iters = (max(1, min_samples) + self._num_envs - 1) // self._num_envs
        return self._device_steps(iters) if self._device_mode else self._host_steps(iters)

    def _host_steps(self, iters):
        samples = []
        for _ in range(iters):
            pred = self._policy.actor_predict(self.last_state, timesteps=1)
            exp_info = None
            if self._exploration is not None:
                pred["actions"], exp_info = self._exploration.remap_actions(
                    pred["actions"], self._env_ids, self._action_space, self._progress)
            obs, rewards, dones, infos = self._vec_env.step(pred["actions"])
            states = self._policy.make_input_state(obs, np.array(dones))
            host = deep_apply(states, lambda x: x.cpu().numpy() if isinstance(x, torch.Tensor) else x)
            for i in range(self._num_envs):
                info = infos[i]
                if exp_info is not None:
                    info["exploration"] = deep_apply(exp_info, lambda x: x[i])
                samples.append(self._create_sample(
                    deep_apply(pred, lambda x: x[i]), deep_apply(host, lambda x: x[i]),
                    rewards[i], dones[i], info, self._env_ids[i]))
            self.last_state = states
        return samples

    def _predict_device(self, state):
        if not self._use_graph:
            return self._policy.actor_predict(state, timesteps=1, as_numpy=False)
        if self._graphed is None:
            try:
                self._graphed = GraphedPredict(self._policy, _to_device_tree(state, self._policy.device()))
            except Exception as e:                       # capture unsupported for this model: stay eager
                import logging
                logging.getLogger().warning("acting graph capture failed (%s); running eagerly", e)
                self._use_graph = False
                return self._policy.actor_predict(state, timesteps=1, as_numpy=False)
        return self._graphed(_to_device_tree(state, self._policy.device()))

    def _device_steps(self, iters):
        out = None
        for _ in range(iters):
            pred = self._predict_device(self.last_state)
            actions = pred["actions"]
            if self._exploration is not None:
                actions, _ = self._exploration.remap_actions_device(
                    actions, self._env_ids, self._action_space, self._progress)
            obs, rewards, dones, stats = self._vec_env.step_device(actions)
            states = self._policy.make_input_state(obs, dones)
            if out is None:
                example = deep_apply(states, lambda x: x[0].cpu().numpy())
                out = DeviceSamples(example, self._num_envs, self._base_env_id)
            fields = _pack_state(states)
            fields.update(actions=actions.to(torch.int32), policy=pred["qvalues"].contiguous(),
                          rewards=rewards.to(torch.float32), dones=dones.to(torch.uint8),
                          episode_stats=stats)
            out.append(**fields)
            self.last_state = states
        return out


def _to_device_tree(state, device):
    return deep_apply(state, lambda x: x if isinstance(x, torch.Tensor)
                      else torch.as_tensor(x, device=device))


def _pack_state(states):
    """input-state pytree -> the flat per-transition arrays of mirl_ingest."""
    x = states["x"]
    fields = {}
    if isinstance(x, (tuple, list)):
        fields["frames"] = x[0].contiguous()
        fields["extra"] = torch.cat([v.reshape(v.shape[0], -1).float() for v in x[1:]], dim=1).contiguous()
    else:
        fields["frames"] = x.contiguous()
    rec, initials = [], None
    for k, sub in states.items():
        if k == "x" or not sub:
            continue
        for name, v in sub.items():
            if name == "initials":
                initials = v
            else:
                rec.append(v.reshape(v.shape[0], -1).float())
    if rec:
        fields["state"] = torch.cat(rec, dim=1).contiguous()
    if initials is not None:
        fields["initials"] = initials.float().contiguous()
    return fields
