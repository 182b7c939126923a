"""PolicyTrainer (reference rltime/training/policy_trainer.py:10-339): policy /
target-policy creation, actor sampling, reward tracking, step accounting,
target sync keyed on ACTED steps, logging and weight checkpoints."""
import logging
import time

import numpy as np

from rltime_amd.general.value_log import ValueLog
from rltime_amd.general.utils import deep_dictionary_update


class PolicyTrainer:
    def __init__(self, logger, actors, model_config, policy_args={}):
        self.logger = logger
        self.actors = actors
        self.model_config = model_config
        self.policy_args = policy_args
        self.episode_rewards, self.episode_lens = {}, {}
        self.action_hist, self.action_hist_count = {}, 0
        self.target_update_freq = 0
        self.value_log = ValueLog()

    @staticmethod
    def create_policy(**kwargs):
        raise NotImplementedError

    def init_policies(self):
        """policy_trainer.py:39-66."""
        observation_space, action_space = self.actors.get_spaces()
        args = dict(model_config=self.model_config, observation_space=observation_space,
                    action_space=action_space, **self.policy_args)
        self.policy = self.create_policy(**args)
        self.target_policy = self.policy if not self.target_update_freq \
            else self.create_policy(**args)
        self.actors.set_actor_policy(self.policy)

    def sync_target(self):
        self.target_policy.copy_from(self.policy)

    def update_actors(self):
        self.actors.update_state(progress=self.get_train_progress())

    def _track_rewards(self, samples):
        """policy_trainer.py:93-160 (episode stats; batched actors report them
        through `episode_stats` instead of per-sample dicts)."""
        for sample in samples:
            info, env_id = sample["info"], sample["env_id"]
            ep = info.get("episode_info")
            if ep is None:
                self.episode_rewards[env_id] = self.episode_rewards.get(env_id, 0) + sample["reward"]
                self.episode_lens[env_id] = self.episode_lens.get(env_id, 0) + 1
                ep_done = sample["done"]
            else:
                self.episode_rewards[env_id], self.episode_lens[env_id] = ep["reward"], ep["length"]
                ep_done = ep["done"]
            if ep_done:
                self._log_episode(self.episode_rewards[env_id], self.episode_lens[env_id])
                self.episode_rewards[env_id] = 0
                self.episode_lens[env_id] = 0
            action = sample["policy_output"].get("actions")
            if action is not None and not isinstance(action, np.ndarray):
                self.action_hist[int(action)] = self.action_hist.get(int(action), 0) + 1
                self.action_hist_count += 1

    def _log_episode(self, reward, length):
        self.value_log.log("episodes", 1, agg="sum", group="this_interval")
        self.value_log.log("episodes", 1, agg="sum", group="total", scope=None)
        for key, value in (("reward", reward), ("episode_length", length)):
            for last in self.episode_history_windows:
                self.value_log.log(key, value, scope=last, group="last%d" % last, precision=2)
                self.value_log.log(key + "_max", value, agg="max", scope=last, group="last%d" % last, precision=2)

    def _update_steps_trained(self, steps):
        self.value_log.log("steps_trained", steps, agg="sum", group="this_interval")
        self.value_log.log("steps_trained", steps, agg="sum", group="total", scope=None)
        self.ts_steps_trained += steps

    def _get_train_state(self):
        return {}

    def _save_checkpoint(self):
        """policy_trainer.py:170-185: weights (+ empty train_state)."""
        self.logger.save_checkpoint(
            {"policy_state": self.policy.get_state(), "train_state": self._get_train_state()}, self.steps)

    def _log_checkpoint(self):
        """policy_trainer.py:187-228."""
        now = time.time()
        dt = now - self.ts_start + 1e-5
        log = self.value_log.log
        log("steps_acted_per_second", int(self.ts_steps / dt), group="this_interval")
        log("steps_trained_per_second", int(self.ts_steps_trained / dt), group="this_interval")
        log("learner_steps_per_second", self.ts_learner_steps / dt, group="this_interval", precision=3)
        log("train_ratio", self.ts_steps_trained / max(self.ts_steps, 1), group="this_interval")
        log("seconds", now - self.ts_start, group="this_interval", precision=2)
        log("seconds", now - self.global_start, group="total", precision=2)
        log("steps_acted", self.ts_steps, group="this_interval")
        log("steps_acted", self.ts_steps, agg="sum", group="total", scope=None)
        info = self.value_log.get()
        hist = []
        if self.action_hist:
            hist = [0] * (max(self.action_hist) + 1)
            for k, v in self.action_hist.items():
                hist[k] = round(v / self.action_hist_count, 3)
            self.action_hist, self.action_hist_count = {}, 0
        deep_dictionary_update(info, {"acting": {"actions": hist}})
        self.logger.log_result("train", info, self.steps)
        self._save_checkpoint()
        self.ts_start, self.ts_steps, self.ts_steps_trained, self.ts_learner_steps = now, 0, 0, 0

    def _start_timer(self, name):
        self._timer_name, self._timer_time = name, time.time()

    def _end_timer(self):
        ms = (time.time() - self._timer_time) * 1000.0
        self.value_log.log(self._timer_name, ms, agg="mean", group="timings_mean_ms", precision=2)
        self.value_log.log(self._timer_name, ms, agg="sum", group="timings_total_ms", precision=2)

    def _process_new_samples(self, new_samples):
        """policy_trainer.py:248-254: stats on raw rewards, then sign clipping."""
        if hasattr(new_samples, "process"):         # batched device samples
            new_samples.process(self)
            return
        self._track_rewards(new_samples)
        if self.clip_rewards:
            for sample in new_samples:
                sample["reward"] = np.sign(sample["reward"])

    def sample_actors(self, min_samples):
        """policy_trainer.py:256-282."""
        self._start_timer("sample_actors")
        samples = self.actors.get_samples(min_samples)
        if not samples:
            return None
        self._process_new_samples(samples)
        n = len(samples)
        self.steps += n
        self.ts_steps += n
        if self.target_update_freq > 0 and \
                (self.steps // self.target_update_freq) != ((self.steps - n) // self.target_update_freq):
            self.sync_target()
        if (self.steps // self.log_freq) != ((self.steps - n) // self.log_freq):
            self._log_checkpoint()
        self._end_timer()
        return samples

    def train(self, total_steps, log_freq=10000, target_update_freq=0, clip_rewards=False,
              early_stop_steps=None, episode_history_windows=[10, 100], **kwargs):
        """policy_trainer.py:284-325; remaining kwargs go down the _train chain
        and unknown ones raise TypeError there."""
        self.target_update_freq = target_update_freq
        self.episode_history_windows = episode_history_windows
        self.log_freq = log_freq
        self.clip_rewards = clip_rewards
        self.ts_start = self.global_start = time.time()
        self.steps = self.ts_steps = self.ts_steps_trained = self.ts_learner_steps = 0
        self.total_steps = total_steps
        self.early_stop_steps = early_stop_steps
        self.init_policies()
        self.update_actors()
        logging.getLogger().info("Training start with total acting ENVs: %d", self.actors.get_env_count())
        self._train(**kwargs)

    def _train(self, **kwargs):
        raise NotImplementedError

    def train_is_done(self):
        return self.get_train_progress() >= 1.0 or \
            (self.early_stop_steps is not None and self.steps >= self.early_stop_steps)

    def get_train_progress(self):
        return self.steps / self.total_steps
