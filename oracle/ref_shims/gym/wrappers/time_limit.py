from .. import Wrapper


class TimeLimit(Wrapper):
    def __init__(self, env, max_episode_steps=None, **kw):
        super().__init__(env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = 0

    def step(self, action):
        o, r, d, i = self.env.step(action)
        self._elapsed_steps += 1
        if self._max_episode_steps and \
                self._elapsed_steps >= self._max_episode_steps:
            d = True
        return o, r, d, i

    def reset(self, **kw):
        self._elapsed_steps = 0
        return self.env.reset(**kw)
