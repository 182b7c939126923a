"""Synthetic vectorised Atari-shaped environment: i.i.d. uniform u8 frames
(channel-first (4,84,84) like the reference's WindowedEnv output,
env_wrappers/common.py:141-160), rewards in {-1,0,1} with p=(.1,.8,.1), done
with p=0.002.  Real emulators are CPU code and out of
scope; the benchmark contract is synthetic data of this shape.

A step is ONE kernel with no host-side state (csrc/acting.hip k_synth_env_step): the
step counter lives on the device, the observation is frame batch t % pool of a
pre-generated pool, reward and done come from a Philox4x32-10 block per (seed, t,
env).  `step_into` writes caller-owned static buffers and can therefore be captured
into a HIP graph of a whole acting rollout (acting/fast_step.py); `step_device` is the
same kernel into fresh tensors."""
import ctypes as C

import numpy as np
import torch

from rltime_amd.spaces import Box, Discrete


def _p(t):
    return C.c_void_p(t.data_ptr())


class SyntheticAtariVecEnv:
    def __init__(self, num_envs, frame_shape=(4, 84, 84), n_actions=6, done_prob=0.002,
                 reward_probs=(0.1, 0.8, 0.1), device="cuda", seed=0, pool=8, frame_stack=False):
        self.num_envs = num_envs
        self.observation_space = Box(0, 255, frame_shape, np.uint8)
        self.action_space = Discrete(n_actions)
        self.device = torch.device(device)
        self.done_prob = float(done_prob)
        self.seed = int(seed) & 0x7FFFFFFFFFFFFFFF
        g = torch.Generator(device=self.device).manual_seed(seed)
        # a small pool of pre-generated frame batches keeps frame synthesis out
        # of the timed region while every step still moves real bytes
        # frame_stack=True: observations follow the frame-stack wrapper's contract
        # (env_wrappers/common.py:141-178 under an auto-resetting vec env): each step
        # rolls the window by one NEW plane, a done step returns zeros + the new plane —
        # what a real Atari pipeline produces, and what frame_stack_dedup storage needs
        self.frame_stack = bool(frame_stack)
        self._row_shape = tuple(frame_shape[1:]) if self.frame_stack else tuple(frame_shape)
        self._pool = torch.randint(0, 256, (pool, num_envs) + self._row_shape, dtype=torch.uint8,
                                   device=self.device, generator=g)
        self._row_bytes = int(np.prod(self._row_shape))
        self._stack = torch.zeros((num_envs,) + tuple(frame_shape), dtype=torch.uint8, device=self.device) \
            if self.frame_stack else None
        cum = np.cumsum(reward_probs)
        self._p_neg, self._p_nonpos = float(cum[0]), float(cum[1])
        # the step counter: a PAIR of device words read / written alternately (a launch reads clock[slot], writes
        # clock[slot ^ 1]: no workgroup sees the new value, no atomics); the host only tracks the parity
        self._clock = torch.zeros(2, dtype=torch.int64, device=self.device)
        self._slot = 0

    def reset(self):
        if self.frame_stack:
            self._stack.zero_()
            self._stack[:, -1] = self._pool[0]
            return self._stack.clone()
        return self._pool[0]

    # -- one step, no host state ------------------------------------------------------------------
    def supports_step_into(self):
        """Can a step write caller-owned static buffers with a fixed launch (HIP-graph capturable)?"""
        return (not self.frame_stack) and self.device.type == "cuda" and self._row_bytes % 16 == 0

    def step_into(self, obs_out, rewards_out, dones_out):
        """obs_out uint8 [E, ...frame], rewards_out float32 [E], dones_out uint8 [E] <- step t = clock + 1."""
        from rltime_amd._lib import lib, check
        check(lib.mirl_synth_env_step(self.num_envs, self._row_bytes, _p(self._pool), self._pool.shape[0], _p(self._clock),
                                      self._slot, self.seed, self._p_neg, self._p_nonpos, self.done_prob, _p(obs_out),
                                      _p(rewards_out), _p(dones_out), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "mirl_synth_env_step")
        self._slot ^= 1

    def step_into_args(self, obs_out, rewards_out, dones_out):
        """The leading arguments of mirl_synth_env_step / mirl_synth_env_step_pre for the NEXT step (a caller that fuses
        the env step with its own work launches the kernel itself and then calls advance_host())."""
        return (self.num_envs, self._row_bytes, _p(self._pool), self._pool.shape[0], _p(self._clock), self._slot, self.seed,
                self._p_neg, self._p_nonpos, self.done_prob, _p(obs_out), _p(rewards_out), _p(dones_out))

    def advance_host(self):
        self._slot ^= 1

    def clock_parity(self):
        """Which word of the clock pair the NEXT step reads: part of the identity of a captured rollout."""
        return self._slot

    def skip_host(self, steps):
        """`steps` steps were replayed from a captured graph: advance the host-side parity like step_into would have."""
        self._slot ^= steps & 1

    def step_device(self, actions):
        E = self.num_envs
        if self.device.type != "cuda" or self._row_bytes % 16:
            return self._step_torch()
        obs = torch.empty((E,) + self._row_shape, dtype=torch.uint8, device=self.device)
        rewards = torch.empty(E, dtype=torch.float32, device=self.device)
        dones8 = torch.empty(E, dtype=torch.uint8, device=self.device)
        self.step_into(obs, rewards, dones8)
        dones = dones8.view(torch.bool)
        if self.frame_stack:
            obs = self._shift(obs, dones)
        return obs, rewards, dones, None

    def _step_torch(self):
        """Odd frame sizes / CPU: the same process with torch ops (own generator stream; used by host-side tests only)."""
        if not hasattr(self, "_g"):
            self._g = torch.Generator(device=self.device).manual_seed(self.seed + 1)
        self._clock[self._slot ^ 1] = self._clock[self._slot] + 1
        self._slot ^= 1
        t = int(self._clock[self._slot])
        obs = self._pool[t % self._pool.shape[0]]
        u = torch.rand(2, self.num_envs, device=self.device, generator=self._g)
        rewards = torch.where(u[0] < self._p_neg, -1.0, torch.where(u[0] < self._p_nonpos, 0.0, 1.0)).float()
        dones = u[1] < self.done_prob
        if self.frame_stack:
            obs = self._shift(obs, dones)
        return obs, rewards, dones, None

    def _shift(self, newest, dones):
        nxt = torch.empty_like(self._stack)
        plane = newest[0].numel()
        if self._stack.is_cuda and plane % 16 == 0:
            # roll by one plane, zero-fill on reset, append the new plane: one launch (csrc/acting.hip)
            from rltime_amd._lib import lib, check
            check(lib.mirl_stack_shift(self.num_envs, self._stack.shape[1], plane, _p(self._stack), _p(nxt), _p(newest),
                                       _p(dones.view(torch.uint8)), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                  "mirl_stack_shift")
        else:
            keep = (~dones).to(torch.uint8).view(-1, 1, 1, 1)
            torch.mul(self._stack[:, 1:], keep, out=nxt[:, :-1])      # roll by one plane; a reset zero-fills
            nxt[:, -1] = newest
        self._stack = nxt
        return nxt

    def step(self, actions):
        obs, rewards, dones, _ = self.step_device(torch.as_tensor(actions, device=self.device))
        return obs, rewards.double().cpu().numpy(), dones.cpu().numpy(), [dict() for _ in range(self.num_envs)]

    def get_state(self):
        return {"t": int(self._clock[self._slot].item()), "stack": None if self._stack is None else self._stack.cpu(),
                "generator": self._g.get_state().cpu() if hasattr(self, "_g") else None}

    def set_state(self, state):
        self._clock.zero_()
        self._clock[0] = int(state["t"])
        self._slot = 0
        if state.get("stack") is not None and self._stack is not None:
            self._stack = state["stack"].to(self.device)
        if state.get("generator") is not None:
            self._step_torch_generator_restore(state["generator"])

    def _step_torch_generator_restore(self, gstate):
        self._g = torch.Generator(device=self.device).manual_seed(self.seed + 1)
        self._g.set_state(gstate.cpu())

    def close(self):
        pass
