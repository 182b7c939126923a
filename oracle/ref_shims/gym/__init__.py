"""Minimal stand-in for the `gym` package, ONLY so that the golden-vector
generator (tests/golden/generate.py) can import the unmodified reference in the
development container (the reference does `import gym` at module import time,
rltime/models/torch/torch_model.py:2, and gym is not installed here).

This is test tooling for the oracle. It is not part of the product and never
travels to the GPU box in any role other than inert source text.
"""
from . import spaces  # noqa: F401


class Env:
    observation_space = None
    action_space = None

    def reset(self):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    def seed(self, seed=None):
        return [seed]

    def close(self):
        pass

    def render(self, mode="human"):
        pass

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.observation_space = env.observation_space
        self.action_space = env.action_space

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, action):
        return self.env.step(action)

    def seed(self, seed=None):
        return self.env.seed(seed)

    def close(self):
        return self.env.close()

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)


class ObservationWrapper(Wrapper):
    def reset(self, **kw):
        return self.observation(self.env.reset(**kw))

    def step(self, action):
        o, r, d, i = self.env.step(action)
        return self.observation(o), r, d, i

    def observation(self, o):
        raise NotImplementedError


class RewardWrapper(Wrapper):
    def step(self, action):
        o, r, d, i = self.env.step(action)
        return o, self.reward(r), d, i


class ActionWrapper(Wrapper):
    def step(self, action):
        return self.env.step(self.action(action))


def make(name, **kw):
    raise RuntimeError("gym shim: no environments are available (%s)" % name)
