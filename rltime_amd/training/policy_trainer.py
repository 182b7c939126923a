"""PolicyTrainer — the outermost trainer layer, with the reference's contract
(rltime/training/policy_trainer.py:10-339): builds the online (and optional
target) policy from the model config, pulls samples from the actors, keeps step
counters, syncs the target network on ACTED-step boundaries, writes one log row
and a weight checkpoint per `log_freq` acted steps.  Training kwargs travel down
the `_train(**kwargs)` chain of the subclasses; unknown ones raise TypeError."""
import logging

import numpy as np

from rltime_amd.general.value_log import ValueLog
from rltime_amd.general.utils import deep_dictionary_update
from .episode_stats import EpisodeStats, IntervalClock


def _crossed(before, after, every):
    """True when a multiple of `every` lies in (before, after]."""
    return every > 0 and (after // every) != (before // every)


class PolicyTrainer:
    def __init__(self, logger, actors, model_config, policy_args={}):
        self.logger, self.actors = logger, actors
        self.model_config, self.policy_args = model_config, policy_args
        self.target_update_freq = 0
        self.value_log = ValueLog()
        self._gpu_spans = []
        self.data_parallel = None       # rltime_amd.parallel.DataParallel when one process per GPU
        self.resume_from = None         # directory of a run written with full_checkpoints=True (training/resume.py)
        self.full_checkpoints = False
        self._full_checkpoint_due = False

    # -- hooks for subclasses ------------------------------------------------
    @staticmethod
    def create_policy(**kwargs):
        raise NotImplementedError

    def _train(self, **kwargs):
        raise NotImplementedError

    def _get_train_state(self):
        return {}       # the reference checkpoints weights only (policy_trainer.py:170-185)

    # -- policies ------------------------------------------------------------
    def init_policies(self):
        obs_space, act_space = self.actors.get_spaces()
        make = lambda: self.create_policy(                                    # noqa: E731
            model_config=self.model_config, observation_space=obs_space,
            action_space=act_space, **self.policy_args)
        self.policy = make()
        # a separate target network only when a sync period is configured
        self.target_policy = make() if self.target_update_freq else self.policy
        if self.data_parallel is not None and self.data_parallel.active:
            # every rank starts from rank 0's two networks (the reference initialises the
            # target network separately, policy_trainer.py:58-60) and accumulates
            # gradients straight into one all-reduce bucket
            self.data_parallel.broadcast_parameters(self.policy)
            if self.target_policy is not self.policy:
                self.data_parallel.broadcast_parameters(self.target_policy)
            self.data_parallel.attach(self.policy)
        self.actors.set_actor_policy(self.policy)

    def sync_target(self):
        self.target_policy.copy_from(self.policy)

    def update_actors(self):
        self.actors.update_state(progress=self.get_train_progress())

    # -- progress --------------------------------------------------------------
    def get_train_progress(self):
        return self.steps / self.total_steps

    def train_is_done(self):
        if self.get_train_progress() >= 1.0:
            return True
        return self.early_stop_steps is not None and self.steps >= self.early_stop_steps

    def _update_steps_trained(self, steps):
        for group, scope in (("this_interval", "interval"), ("total", None)):
            self.value_log.log("steps_trained", steps, agg="sum", group=group, scope=scope)
        self.clock.trained += steps

    @property
    def ts_learner_steps(self):
        return self.clock.learner_steps

    @ts_learner_steps.setter
    def ts_learner_steps(self, v):
        self.clock.learner_steps = v

    # -- timers (same keys as the reference's timings_* groups) -------------------
    # The reference times phases with a wall clock only (policy_trainer.py:230-246),
    # which mis-attributes asynchronous GPU work to whichever phase synchronises
    # first.  Besides the same wall-clock groups, every phase is bracketed by HIP
    # events on the current stream; they are resolved (no sync in the hot loop) when
    # the interval is logged, into the groups timings_gpu_mean_ms / _total_ms.
    def _start_timer(self, name):
        import time
        ev = None
        if self._gpu_timing():
            import torch
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        self._timer = (name, time.time(), ev)

    def _end_timer(self):
        import time
        name, t0, ev0 = self._timer
        ms = (time.time() - t0) * 1e3
        for agg, group in (("mean", "timings_mean_ms"), ("sum", "timings_total_ms")):
            self.value_log.log(name, ms, agg=agg, group=group, precision=2)
        if ev0 is not None:
            import torch
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            self._gpu_spans.append((name, ev0, ev1))
            if len(self._gpu_spans) > 4096:          # long intervals: resolve the oldest (already complete)
                self._resolve_gpu_spans(2048)

    def _gpu_timing(self):
        pol = getattr(self, "policy", None)
        if not (pol is not None and hasattr(pol, "is_cuda") and pol.is_cuda()):
            return False
        import torch
        # (a phase timed while the stream is capturing would record its events INTO the graph: nothing to resolve later)
        return not torch.cuda.is_current_stream_capturing()

    def _resolve_gpu_spans(self, count=None):
        spans = self._gpu_spans if count is None else self._gpu_spans[:count]
        for name, e0, e1 in spans:
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            for agg, group in (("mean", "timings_gpu_mean_ms"), ("sum", "timings_gpu_total_ms")):
                self.value_log.log(name, ms, agg=agg, group=group, precision=3)
        del self._gpu_spans[:len(spans)]

    # -- acting ----------------------------------------------------------------
    def _log_episode(self, reward, length):
        self.episodes.episode_finished(reward, length)

    def _process_new_samples(self, new_samples):
        """Episode statistics on the raw rewards, then optional sign clipping
        (policy_trainer.py:248-254).  Device batches do both on the GPU."""
        if hasattr(new_samples, "process"):
            new_samples.process(self)
            return
        for sample in new_samples:
            self.episodes.observe(sample)
            if self.clip_rewards:
                sample["reward"] = np.sign(sample["reward"])

    def sample_actors(self, min_samples):
        self._start_timer("sample_actors")
        samples = self.actors.get_samples(min_samples)
        if not samples:
            return None
        self._process_new_samples(samples)
        before = self.steps
        self.steps += len(samples)
        self.clock.acted += len(samples)
        if _crossed(before, self.steps, self.target_update_freq):
            self.sync_target()
        if _crossed(before, self.steps, self.log_freq):
            self._log_checkpoint()
            # the full (resumable) checkpoint is taken at the end of this loop iteration:
            # right here the new samples are acted but not yet in the replay
            self._full_checkpoint_due = self.full_checkpoints
        self._end_timer()
        return samples

    # -- logging / checkpoint ------------------------------------------------------
    def _log_checkpoint(self):
        self._resolve_gpu_spans()
        if getattr(self.policy, "is_cuda", lambda: False)():
            from rltime_amd.models.torch import lstm_seq
            lstm_seq.check_status()          # after the synchronisation above: nothing of a failed sweep gets checkpointed
        rates, total_seconds = self.clock.rates()
        for key, val in rates.items():
            self.value_log.log(key, val, group="this_interval")
        self.value_log.log("seconds", total_seconds, group="total")
        self.value_log.log("steps_acted", rates["steps_acted"], agg="sum", group="total", scope=None)
        hist = getattr(self, "history_buffer", None)
        if hist is not None and getattr(hist, "_global", None) is not None:
            # exact global sampling: strata lost to a too-small padded batch (0 by construction; raises otherwise)
            self.value_log.log("global_sampling_dropped_strata", hist.check_dropped_strata(), group="train")
        row = self.value_log.get()
        deep_dictionary_update(row, {"acting": {"actions": self.episodes.action_histogram()}})
        self.logger.log_result("train", row, self.steps)
        self.logger.save_checkpoint(
            {"policy_state": self.policy.get_state(), "train_state": self._get_train_state()}, self.steps)

    # -- entry point -----------------------------------------------------------------
    def save_full_checkpoint(self):
        """Everything the next loop iteration depends on (training/resume.py), next to
        the reference-style weights-only checkpoint.p."""
        from . import resume
        path = getattr(self.logger, "path", None)
        if path is None:
            raise ValueError("full_checkpoints needs a directory logger (train.py --log-dir)")
        self._resolve_gpu_spans()
        if getattr(self.policy, "is_cuda", lambda: False)():
            import torch
            from rltime_amd.models.torch import lstm_seq
            torch.cuda.synchronize()
            lstm_seq.check_status()          # weights updated from a failed sweep's gradients must not become the resume point
        resume.save(self, path)
        if self.data_parallel is not None:
            self.data_parallel.barrier()         # the checkpoint is complete only when every rank's files are
        self._full_checkpoint_due = False

    def train(self, total_steps, log_freq=10000, target_update_freq=0, clip_rewards=False,
              early_stop_steps=None, episode_history_windows=[10, 100], full_checkpoints=False, **kwargs):
        """policy_trainer.py:284-325.  full_checkpoints (not in the reference): also write
        a resumable checkpoint (replay shard, optimizer, RNG streams, counters) at every
        log interval; `python -m rltime_amd.train ... --resume <log dir>` continues it."""
        self.full_checkpoints = full_checkpoints
        self.total_steps, self.early_stop_steps = total_steps, early_stop_steps
        self.log_freq, self.target_update_freq = log_freq, target_update_freq
        self.clip_rewards = clip_rewards
        self.episodes = EpisodeStats(self.value_log, episode_history_windows)
        self.clock = IntervalClock()
        self.steps = 0
        self.init_policies()
        self.update_actors()
        logging.getLogger().info("training starts with %d acting envs", self.actors.get_env_count())
        self._train(**kwargs)
