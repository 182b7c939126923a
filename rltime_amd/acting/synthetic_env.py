"""Synthetic vectorised Atari-shaped environment: i.i.d. uniform u8 frames
(channel-first (4,84,84) like the reference's WindowedEnv output,
env_wrappers/common.py:141-160), rewards in {-1,0,1} with p=(.1,.8,.1), done
with p=0.002.  Real emulators are CPU code and out of
scope; the benchmark contract is synthetic data of this shape."""
import numpy as np
import torch

from rltime_amd.spaces import Box, Discrete


class SyntheticAtariVecEnv:
    POOL = 512
    def __init__(self, num_envs, frame_shape=(4, 84, 84), n_actions=6, done_prob=0.002,
                 reward_probs=(0.1, 0.8, 0.1), device="cuda", seed=0, pool=8, frame_stack=False):
        self.num_envs = num_envs
        self.observation_space = Box(0, 255, frame_shape, np.uint8)
        self.action_space = Discrete(n_actions)
        self.device = torch.device(device)
        self.done_prob = done_prob
        self._g = torch.Generator(device=self.device).manual_seed(seed)
        # a small pool of pre-generated frame batches keeps frame synthesis out
        # of the timed region while every step still moves real bytes
        # frame_stack=True: observations follow the frame-stack wrapper's contract
        # (env_wrappers/common.py:141-178 under an auto-resetting vec env): each step
        # rolls the window by one NEW plane, a done step returns zeros + the new plane —
        # what a real Atari pipeline produces, and what frame_stack_dedup storage needs
        self.frame_stack = bool(frame_stack)
        shape = (num_envs,) + (tuple(frame_shape[1:]) if self.frame_stack else tuple(frame_shape))
        self._pool = [torch.randint(0, 256, shape, dtype=torch.uint8,
                                    device=self.device, generator=self._g) for _ in range(pool)]
        self._stack = torch.zeros((num_envs,) + tuple(frame_shape), dtype=torch.uint8, device=self.device) \
            if self.frame_stack else None
        self._cum = torch.tensor(np.cumsum(reward_probs), device=self.device, dtype=torch.float32)
        self._t = 0
        # rewards / dones are pre-drawn for POOL steps at a time (like the frames, their
        # synthesis is not what is being measured): one burst of kernels per 512 steps
        # instead of six tiny launches per step
        self._sched, self._sched_at = None, 0
        self._ep_reward = torch.zeros(num_envs, device=self.device)
        self._ep_len = torch.zeros(num_envs, device=self.device)

    def reset(self):
        if self.frame_stack:
            self._stack.zero_()
            self._stack[:, -1] = self._pool[0]
            return self._stack.clone()
        return self._pool[0]

    def step_device(self, actions):
        self._t += 1
        obs = self._pool[self._t % len(self._pool)]
        if self._sched is None or self._sched_at == self.POOL:
            u = torch.rand(2, self.POOL, self.num_envs, device=self.device, generator=self._g)
            self._sched = (torch.bucketize(u[0], self._cum).clamp(max=2).float() - 1.0, u[1] < self.done_prob)
            self._sched_at = 0
        rewards, dones = self._sched[0][self._sched_at], self._sched[1][self._sched_at]
        self._sched_at += 1
        if self.frame_stack:
            nxt = torch.empty_like(self._stack)
            plane = obs[0].numel()
            if self._stack.is_cuda and plane % 16 == 0:
                # roll by one plane, zero-fill on reset, append the new plane: one launch (csrc/acting.hip)
                import ctypes as C
                from rltime_amd._lib import lib, check
                p = lambda t: C.c_void_p(t.data_ptr())                      # noqa: E731
                check(lib.mirl_stack_shift(self.num_envs, self._stack.shape[1], plane, p(self._stack), p(nxt), p(obs),
                                           p(dones.view(torch.uint8)), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                      "mirl_stack_shift")
            else:
                keep = (~dones).to(torch.uint8).view(-1, 1, 1, 1)
                torch.mul(self._stack[:, 1:], keep, out=nxt[:, :-1])      # roll by one plane; a reset zero-fills
                nxt[:, -1] = obs
            self._stack = nxt
            obs = nxt
        return obs, rewards, dones, None

    def step(self, actions):
        obs, rewards, dones, _ = self.step_device(torch.as_tensor(actions, device=self.device))
        return obs, rewards.double().cpu().numpy(), dones.cpu().numpy(), [dict() for _ in range(self.num_envs)]

    def get_state(self):
        return {"t": self._t, "generator": self._g.get_state().cpu(),
                "sched": None if self._sched is None else (self._sched[0].cpu(), self._sched[1].cpu(), self._sched_at),
                "stack": None if self._stack is None else self._stack.cpu()}

    def set_state(self, state):
        self._t = state["t"]
        self._g.set_state(state["generator"].cpu())
        if state.get("sched") is not None:
            self._sched = (state["sched"][0].to(self.device), state["sched"][1].to(self.device))
            self._sched_at = state["sched"][2]
        if state.get("stack") is not None and self._stack is not None:
            self._stack = state["stack"].to(self.device)

    def close(self):
        pass
