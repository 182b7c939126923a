"""Episode / action statistics gathered from acted samples, and the per-interval
throughput counters, factored out of the trainer (the reference keeps these inline
in PolicyTrainer: rltime/training/policy_trainer.py:93-160,187-228)."""
import time

import numpy as np


class EpisodeStats:
    def __init__(self, value_log, windows):
        self.value_log = value_log
        self.windows = list(windows)
        self.reward = {}
        self.length = {}
        self.action_counts = {}
        self.action_total = 0
        self.device_tracker = None       # the device actor's EpisodeTracker: action counts live on the GPU

    def episode_finished(self, reward, length):
        log = self.value_log.log
        log("episodes", 1, agg="sum", group="this_interval")
        log("episodes", 1, agg="sum", group="total", scope=None)
        for w in self.windows:
            group = "last%d" % w
            log("reward", reward, scope=w, group=group, precision=2)
            log("reward_max", reward, agg="max", scope=w, group=group, precision=2)
            log("episode_length", length, scope=w, group=group, precision=2)
            log("episode_length_max", length, agg="max", scope=w, group=group, precision=2)

    def observe(self, sample):
        """One reference-style sample dict (acting_interface.py:83-90)."""
        env = sample["env_id"]
        monitor = sample["info"].get("episode_info")
        if monitor is not None:          # a monitor wrapper reports the true episode
            self.reward[env], self.length[env] = monitor["reward"], monitor["length"]
            finished = monitor["done"]
        else:
            self.reward[env] = self.reward.get(env, 0) + sample["reward"]
            self.length[env] = self.length.get(env, 0) + 1
            finished = sample["done"]
        if finished:
            self.episode_finished(self.reward[env], self.length[env])
            self.reward[env] = self.length[env] = 0
        info = sample["info"]
        if "exploration" in info:
            self.value_log.log_dict(info["exploration"], group="acting->exploration")
        if "env_stats" in info:
            self.value_log.log_dict(info["env_stats"], group="acting->env_stats")
        action = sample["policy_output"].get("actions")
        if action is not None and not isinstance(action, np.ndarray):
            self.action_counts[int(action)] = self.action_counts.get(int(action), 0) + 1
            self.action_total += 1

    def action_histogram(self):
        if self.device_tracker is not None:
            for a, c in enumerate(self.device_tracker.take_action_counts()):
                if c:
                    self.action_counts[a] = self.action_counts.get(a, 0) + c
                    self.action_total += c
        if not self.action_counts:
            return []
        hist = [0] * (max(self.action_counts) + 1)
        for a, c in self.action_counts.items():
            hist[a] = round(c / self.action_total, 3)
        self.action_counts, self.action_total = {}, 0
        return hist


class IntervalClock:
    """acted / trained / learner-step counters of the current log interval."""

    def __init__(self):
        self.start = self.origin = time.time()
        self.acted = self.trained = self.learner_steps = 0

    def rates(self):
        now = time.time()
        dt = now - self.start + 1e-5
        out = {
            "steps_acted_per_second": int(self.acted / dt),
            "steps_trained_per_second": int(self.trained / dt),
            "learner_steps_per_second": round(self.learner_steps / dt, 3),
            "train_ratio": self.trained / max(self.acted, 1),
            "seconds": round(now - self.start, 2),
            "steps_acted": self.acted,
        }
        total_seconds = round(now - self.origin, 2)
        self.start, self.acted, self.trained, self.learner_steps = now, 0, 0, 0
        return out, total_seconds
