"""CartPole as a host-side vector env for the CPU plumbing config (BASELINE configs[0]: `cartpole_ppo.json`,
CartPole-v0 / -v1 with the reference's `max_episode_steps` override, rltime/configs/cartpole_common.json:4-7).

gym is not installed here, so this is the build's own statement of the classic cart-pole balancing task (Barto,
Sutton & Anderson 1983: a pole hinged on a cart that is pushed left or right with a fixed force; Euler integration
at 50 Hz; an episode ends when the pole leaves +-12 degrees, the cart leaves +-2.4, or after `max_episode_steps`;
reward 1 per step), vectorised over envs with NumPy and auto-resetting like the reference's vec envs
(env_wrappers/vec_env/simple.py:5-44: a finished env returns the NEW episode's first observation with done=True).
Each transition's info carries `episode_info` the way the reference's EpisodeTracker wrapper reports it
(env_wrappers/common.py:419-460), so the trainer logs true episode rewards / lengths."""
import numpy as np

from rltime_amd.spaces import Box, Discrete

GRAVITY, CART_MASS, POLE_MASS, POLE_HALF_LENGTH, PUSH, DT = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
ANGLE_LIMIT, POSITION_LIMIT = 12 * 2 * np.pi / 360, 2.4


class CartPoleVecEnv:
    def __init__(self, num_envs, max_episode_steps=200, seed=0):
        self.num_envs = int(num_envs)
        self.max_episode_steps = int(max_episode_steps)
        high = np.array([2 * POSITION_LIMIT, np.finfo(np.float32).max, 2 * ANGLE_LIMIT, np.finfo(np.float32).max], np.float32)
        self.observation_space = Box(-high, high, (4,), np.float32)
        self.action_space = Discrete(2)
        self._rng = np.random.RandomState(seed)
        self._x = np.zeros((self.num_envs, 4), np.float64)       # cart position, cart velocity, pole angle, pole angular velocity
        self._t = np.zeros(self.num_envs, np.int64)
        self._ret = np.zeros(self.num_envs, np.float64)

    def _fresh(self, count):
        return self._rng.uniform(-0.05, 0.05, size=(count, 4))

    def reset(self):
        self._x[:] = self._fresh(self.num_envs)
        self._t[:] = 0
        self._ret[:] = 0
        return self._x.astype(np.float32)

    def step(self, actions):
        actions = np.asarray(actions).reshape(self.num_envs)
        x, v, th, w = self._x.T
        force = np.where(actions == 1, PUSH, -PUSH)
        total = CART_MASS + POLE_MASS
        ml = POLE_MASS * POLE_HALF_LENGTH
        sin, cos = np.sin(th), np.cos(th)
        tmp = (force + ml * w * w * sin) / total
        th_acc = (GRAVITY * sin - cos * tmp) / (POLE_HALF_LENGTH * (4.0 / 3.0 - POLE_MASS * cos * cos / total))
        x_acc = tmp - ml * th_acc * cos / total
        self._x = np.stack([x + DT * v, v + DT * x_acc, th + DT * w, w + DT * th_acc], axis=1)
        self._t += 1
        self._ret += 1.0
        fell = (np.abs(self._x[:, 0]) > POSITION_LIMIT) | (np.abs(self._x[:, 2]) > ANGLE_LIMIT)
        dones = fell | (self._t >= self.max_episode_steps)
        rewards = np.ones(self.num_envs, np.float64)
        infos = [{"episode_info": {"reward": float(self._ret[i]), "length": int(self._t[i]), "done": bool(dones[i])}}
                 for i in range(self.num_envs)]
        if dones.any():
            idx = np.nonzero(dones)[0]
            self._x[idx] = self._fresh(len(idx))
            self._t[idx] = 0
            self._ret[idx] = 0
        return self._x.astype(np.float32), rewards, dones, infos

    def close(self):
        pass
