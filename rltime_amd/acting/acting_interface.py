"""Acting interface and sample formats (reference rltime/acting/acting_interface.py)."""


class ActingInterface:
    def __init__(self, observation_space, action_space):
        self._observation_space = observation_space
        self._action_space = action_space

    def get_spaces(self):
        return self._observation_space, self._action_space

    def get_samples(self, min_samples):
        raise NotImplementedError

    def get_env_count(self):
        raise NotImplementedError

    def set_actor_policy(self, actor_policy):
        raise NotImplementedError

    def update_state(self, progress, policy_state=None):
        raise NotImplementedError

    def close(self):
        raise NotImplementedError

    def _create_sample(self, policy_output, next_state, reward, done, info, env_id):
        """acting_interface.py:58-90."""
        return {"policy_output": policy_output, "next_state": next_state, "reward": reward,
                "done": done, "info": info, "env_id": env_id}


class DeviceSamples:
    """What the device-resident actor hands to History.update instead of a list
    of per-env dicts: whole vector steps as device tensors (the reference
    splits every step into E dicts, actor.py:132-145, only for the history to
    re-stack them).  len() is the number of transitions, like the list."""

    def __init__(self, example_state, num_envs, env_base=0):
        self.example_state = example_state
        self.num_envs = num_envs
        self.env_base = env_base
        self.vector_steps = []
        self.episode_tracker = None      # acting/episode_tracker.py when the env reports no host-side stats

    def append(self, **fields):
        self.vector_steps.append(fields)

    def __len__(self):
        return len(self.vector_steps) * self.num_envs

    def __bool__(self):
        return bool(self.vector_steps)

    def process(self, trainer):
        """policy_trainer.py:248-254 on the device: episode stats from the raw
        rewards, then sign clipping if configured."""
        import torch
        if self.episode_tracker is not None:
            for reward, length in self.episode_tracker.drain():
                trainer._log_episode(reward, length)
            trainer.episodes.device_tracker = self.episode_tracker
        for step in self.vector_steps:
            stats = step.pop("episode_stats", None)
            if stats is not None:
                for reward, length in stats:
                    trainer._log_episode(reward, length)
            if trainer.clip_rewards:
                step["rewards"] = torch.sign(step["rewards"])
