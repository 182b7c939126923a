"""Episode statistics and action histogram of the device-resident actor.

The reference computes them on the host from every sample dict
(rltime/training/policy_trainer.py:75-131: `_track_rewards`, `_format_action_hist`).
The device actor never brings per-step data to the host, so the running
accumulators live on the GPU (one fused launch per vector step, csrc/acting.hip)
and finished episodes come back through an asynchronous, event-guarded copy: the
acting loop never waits; statistics arrive one or two get_samples calls late."""
import ctypes as C

import torch

from rltime_amd._lib import lib, check


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


class EpisodeTracker:
    ROWS = 64

    def __init__(self, num_envs, num_actions, device):
        self.E, self.A, self.device = num_envs, int(num_actions), device
        self.ep_reward = torch.zeros(num_envs, dtype=torch.float32, device=device)
        self.ep_len = torch.zeros(num_envs, dtype=torch.int32, device=device)
        self.out_reward = torch.zeros((self.ROWS, num_envs), dtype=torch.float32, device=device)
        self.out_len = torch.zeros((self.ROWS, num_envs), dtype=torch.int32, device=device)
        self.action_counts = torch.zeros(max(self.A, 1), dtype=torch.int32, device=device)
        self.row = 0            # next row to write
        self.flushed = 0        # rows already handed to a copy
        self.in_flight = []     # (event, host_reward, host_len)
        self.free = []

    def step(self, rewards, dones_u8, actions_i32):
        """One vector step: raw rewards (f32 [E]), dones (u8 [E]), actions (i32 [E])."""
        if self.row - self.flushed == self.ROWS:
            self.flush()
        r = self.row % self.ROWS
        check(lib.mirl_episode_track(
            self.E, self.A, _p(rewards), _p(dones_u8), _p(actions_i32), _p(self.ep_reward), _p(self.ep_len),
            _p(self.out_reward[r]), _p(self.out_len[r]), _p(self.action_counts),
            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mirl_episode_track")
        self.row += 1

    def begin_step(self):
        """Reserve the ring row of the next vector step (the fused pre-step kernel of
        acting/fast_step.py writes it): same bookkeeping as step()."""
        if self.row - self.flushed == self.ROWS:
            self.flush()
        r = self.row % self.ROWS
        self.row += 1
        return r

    def begin_rollout(self, n):
        """Rows 0 .. n-1 of the ring for the next n vector steps — the SAME rows for every rollout, so that a captured
        graph of the rollout can hold their addresses.  Everything written so far is handed to a copy first (stream
        order keeps that copy ahead of the rollout's writes)."""
        assert 0 < n <= self.ROWS
        self.flush()
        self.row = self.flushed = 0

    def end_rollout(self, n):
        self.row = n

    def get_state(self):
        return {"ep_reward": self.ep_reward.cpu(), "ep_len": self.ep_len.cpu()}

    def set_state(self, state):
        self.ep_reward.copy_(state["ep_reward"].to(self.device))
        self.ep_len.copy_(state["ep_len"].to(self.device))

    def flush(self):
        """Start the asynchronous read-back of the rows written since the last flush."""
        n = self.row - self.flushed
        if n <= 0:
            return
        idx = [(self.flushed + i) % self.ROWS for i in range(n)]
        if self.free:
            hr, hl = self.free.pop()
        else:
            hr = torch.empty((self.ROWS, self.E), dtype=torch.float32, pin_memory=True)
            hl = torch.empty((self.ROWS, self.E), dtype=torch.int32, pin_memory=True)
        lo, hi = idx[0], idx[-1] + 1
        if hi - lo == n:                              # contiguous rows
            hr[:n].copy_(self.out_reward[lo:hi], non_blocking=True)
            hl[:n].copy_(self.out_len[lo:hi], non_blocking=True)
        else:                                         # wrapped around the ring
            k = self.ROWS - lo
            hr[:k].copy_(self.out_reward[lo:], non_blocking=True)
            hl[:k].copy_(self.out_len[lo:], non_blocking=True)
            hr[k:n].copy_(self.out_reward[:n - k], non_blocking=True)
            hl[k:n].copy_(self.out_len[:n - k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.in_flight.append((ev, hr, hl, n))
        self.flushed = self.row

    def drain(self, wait=False):
        """(reward, length) of every finished episode whose copy has completed."""
        out = []
        while self.in_flight and (wait or self.in_flight[0][0].query()):
            ev, hr, hl, n = self.in_flight.pop(0)
            if wait:
                ev.synchronize()
            lens = hl[:n]
            hit = lens.nonzero()
            for s, e in hit.tolist():
                out.append((float(hr[s, e]), int(lens[s, e])))
            self.free.append((hr, hl))
        return out

    def take_action_counts(self):
        """Action counts since the last call (synchronises: log-interval use only)."""
        # the counters only ever grow (an overlapped acting stream may be adding to them right
        # now): report the difference to the last read instead of zeroing them under it
        now = self.action_counts.cpu()
        last = getattr(self, "_counts_read", None)
        self._counts_read = now
        return (now if last is None else now - last).tolist()
