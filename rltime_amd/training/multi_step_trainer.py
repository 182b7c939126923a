"""MultiStepTrainer (reference rltime/training/multi_step_trainer.py:11-379):
THE LOOP — feed/train balance, warm-up, recurrent burn-in, (T,B)->(T*B)
flattening, target values, minibatch epochs, LR anneal."""
import numpy as np
import torch

from .policy_trainer import PolicyTrainer
from rltime_amd.general.type_registry import get_registered_type
from rltime_amd.general.utils import deep_apply


def _flat(x):
    return x.reshape((x.shape[0] * x.shape[1],) + tuple(x.shape[2:]))


class MultiStepTrainer(PolicyTrainer):
    def calc_target_values(self, returns, target_states, target_masks, nsteps, timesteps):
        raise NotImplementedError

    def train_init(self, lr):
        raise NotImplementedError

    def set_lr(self, lr):
        raise NotImplementedError

    def train_batch(self, states, targets, policy_outputs, extra_data, timesteps):
        raise NotImplementedError

    def get_train_indexes(self, batch_size, mini_batch_size, nstep):
        """multi_step_trainer.py:47-68 (np.random.shuffle is consumed even when
        minibatches == 1)."""
        inds = np.arange(batch_size)
        if not self.policy.is_recurrent():
            np.random.shuffle(inds)
            return inds
        assert mini_batch_size % nstep == 0
        per = mini_batch_size // nstep
        inds = inds.reshape((nstep, -1))
        env_inds = np.arange(inds.shape[-1])
        np.random.shuffle(env_inds)
        return np.concatenate([inds[..., env_inds[i:i + per]].ravel()
                               for i in np.arange(0, inds.shape[-1], per)])

    def _get_discount_function(self, gamma):
        """multi_step_trainer.py:70-74; `.gamma` lets the device replay evaluate
        the same n-step return on the GPU."""
        def discount(nstep, reward, policy_output):
            return (gamma ** nstep) * reward
        discount.gamma = gamma
        return discount

    def _sample_and_update_history(self, min_samples):
        new_samples = self.sample_actors(min_samples)
        if new_samples:
            self._start_timer("history_update")
            self.history_buffer.update(new_samples)
            self._end_timer()

    def _burn_in(self, train_data, burn_in_timesteps, do_target_states):
        """multi_step_trainer.py:90-131: no-grad forward of the prefix steps,
        then the stored recurrent state at time index P is replaced by the
        fresh one (zeroed where initials[P] is set) and the prefix rows are
        dropped from every leaf.  All on the device."""
        self._start_timer("burn_in")
        P = burn_in_timesteps
        todo = [(self.policy, train_data["states"])]
        if do_target_states:
            todo.append((self.target_policy, train_data["target_states"]))
        # `states` and `target_states` of a gathered batch are two row ranges of ONE block (history.py:245-265: rows [0, L)
        # and [n, L + n) of the same stacked arrays — in the reference too), so with n == P the row the online pass
        # replaces (its row P) IS the row the target pass, which runs second, starts from.  The reference-pinned small
        # trajectories (P = n = 2) only match with that order kept, so it is: the passes run side by side only when n != P.
        prefixes = [deep_apply(deep_apply(states, lambda x: x[:P]), _flat) for _, states in todo]
        aliased = getattr(self, "nstep_target", None) == P

        def one(policy, states, prefix):
            last_rnn = policy.model.last_recurrent_layer()
            if getattr(self, "burn_in_full_forward", False) or last_rnn is None:
                # the reference's exact call (actor_predict: whole head + a tau draw)
                policy.actor_predict(prefix, timesteps=P, as_numpy=False)
            else:
                # only the recurrent state is consumed: stop after the last
                # recurrent layer (skips the IQN head on P*B rows; changes how many
                # torch.rand taus are consumed vs the reference)
                with torch.no_grad():
                    policy.model(prefix, P, stop_after=last_rnn)
            for i, layer in enumerate(policy.model.layers):
                key = "layer%d_state" % i
                if not states[key]:
                    continue
                fresh = layer.get_state(states[key]["initials"][P])
                for name, value in fresh.items():
                    states[key][name][P] = policy.make_tensor(value)
        rows = train_data["returns"].shape[0] * train_data["returns"].shape[1]
        if len(todo) == 2 and todo[0][0] is not todo[1][0] and self._passes_overlap(rows, train_data["returns"].shape[1]) and not aliased \
                and not getattr(self, "burn_in_full_forward", False):
            # the online and the target net's prefix passes share nothing but the (read-only) converted frames: side by side
            self._side_by_side(lambda: one(*todo[1], prefixes[1]), lambda: one(*todo[0], prefixes[0]))
        else:
            for (policy, states), prefix in zip(todo, prefixes):
                one(policy, states, prefix)
        out = deep_apply(train_data, lambda x: x[P:])
        self._end_timer()
        return out

    # -- independent no-grad passes side by side --------------------------------------------------------------------------
    # One rank of an 8-GPU job trains 64 sequences: 5 120 rows per pass, kernels that fill a quarter to a half of the chip,
    # and LSTM sweeps whose time is the exchange latency of 40 / 80 dependent steps whatever the batch.  The target net's
    # and the online net's no-grad passes of a learner step — the two burn-in prefixes (multi_step_trainer.py:90-131), then
    # the target-value pass and the double-Q selection pass (training/torch/iqn.py:15-45) — share no result, so the second
    # of each pair runs on a second HIP stream.  Same kernels on the same operands (workspaces are per stream, the
    # persistent sweeps' exchange buffers per call): identical results; the quantile fractions are still drawn in the
    # reference's order (the host issues torch.rand in program order).  "auto": only where a pass is small (<= 8 192 rows);
    # at B = 512 every kernel fills the chip and two streams only interleave.
    def _passes_overlap(self, rows, batch=None):
        """batch: sequences per pass (the B of a (T, B) block).  Whatever the mode, two passes only go side by side when
        the persistent LSTM sweeps of BOTH fit the chip together (models/torch/lstm_seq.py two_sweeps_fit): each sweep
        spins on peer workgroups that must all be resident — at B = 512, H = 512 one sweep owns every compute unit."""
        mode = getattr(self, "overlap_passes", "auto")
        on_gpu = getattr(self.policy, "is_cuda", None)        # (a bare plugin policy without the method: one stream)
        if mode in (False, None, "off") or on_gpu is None or not on_gpu() or getattr(self, "_ov", None) is not None:
            return False
        if not (mode is True or mode == "on" or rows <= 8192):
            return False
        if batch:
            from rltime_amd.models.torch import lstm_seq
            for pol in (self.policy, self.target_policy):
                for layer in getattr(getattr(pol, "model", None), "layers", []):
                    units = getattr(layer, "num_units", None)
                    if units and getattr(layer, "fused", False) and not lstm_seq.two_sweeps_fit(int(batch), int(units)):
                        return False
        return True

    def _side_by_side(self, on_side, on_main):
        """Run on_side() on the second stream while on_main() runs on the current one; returns their results."""
        main = torch.cuda.current_stream()
        side = getattr(self, "_pass_stream", None)
        if side is None:
            side = self._pass_stream = torch.cuda.Stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            a = on_side()
        b = on_main()
        main.wait_stream(side)
        return a, b

    def _init_history_buffer(self, mode, async_history, nstep_target, nstep_train, prefix_steps):
        """multi_step_trainer.py:133-150.  async_history is refused: the
        process wrapper existed to hide Python batch-assembly cost that the
        device replay no longer has (everything is stream-asynchronous)."""
        if async_history:
            raise ValueError("async_history is not supported: the device replay is already asynchronous")
        cls = get_registered_type("history", mode.get("type"))
        self.history_buffer = cls(
            **mode.get("args", {}), nstep_target=nstep_target, nstep_train=nstep_train,
            prefix_steps=prefix_steps, discount_function=self._get_discount_function(self.gamma),
            state_store=self.policy.get_state_store(async_history))
        if getattr(self.actors, "_device_mode", False) and hasattr(self.actors, "set_sink"):
            # the device actor's fused vector step writes straight into the device replay
            self.actors.set_sink(self.history_buffer, clip_rewards=self.clip_rewards)
        dp = self.data_parallel
        if getattr(self, "global_sampling", False) and dp is not None and dp.active \
                and hasattr(self.history_buffer, "enable_global_sampling"):
            self.history_buffer.enable_global_sampling(dp)

    def _train(self, gamma, nstep_train, lr, history_mode, mbatch_size=None, nstep_target=None,
               lr_anneal=False, epochs=1, minibatches=1, warmup_steps=0,
               actor_update_frequency_steps=1000, burn_in_timesteps=0, rnn_steps_train=None,
               rnn_bootstrap=False, async_history=False, overlap_acting=False, global_sampling=False,
               overlap_passes="auto"):
        """multi_step_trainer.py:152-379.  overlap_acting (not in the reference): run the
        acting + ingest of iteration k+1 on a second HIP stream while iteration k trains
        (see _loop_iteration_overlapped)."""
        self.overlap_passes = overlap_passes      # "auto" | True | False: independent no-grad passes on a second stream
        self.overlap_acting = overlap_acting     # True | "serial" (same schedule on ONE stream: race check)
        # multi-GPU + prioritized replay: sample exactly like one tree over all shards
        # (history needs device_rng=True); default: per-shard proportional + global IS weights
        self.global_sampling = bool(global_sampling)
        self._ov = None
        self.train_init(lr)
        self.gamma = gamma
        self.lr = self.base_lr = lr
        self.lr_anneal = lr_anneal
        self.mbatch_size = mbatch_size or self.actors.get_env_count()
        self.nstep_train = nstep_train
        self.nstep_target = nstep_target or nstep_train
        self.epochs, self.minibatches = epochs, minibatches
        self.warmup_steps = warmup_steps
        self.actor_update_frequency_steps = actor_update_frequency_steps
        self.burn_in_timesteps = burn_in_timesteps
        self.rnn_bootstrap = rnn_bootstrap
        if rnn_bootstrap and history_mode.get("args", {}).get("avoid_episode_crossing"):
            # the reference asserts nsteps == nstep_target under rnn_bootstrap
            # (multi_step_trainer.py:294-297); windows shifted to the ring end by
            # avoid_episode_crossing have truncated n-step targets, whose target rows are
            # not the consecutive states a recurrent target pass needs
            raise ValueError("rnn_bootstrap cannot be combined with avoid_episode_crossing: shifted windows "
                             "carry truncated n-step targets (multi_step_trainer.py:294-297 asserts against it)")
        self._init_history_buffer(history_mode, async_history, self.nstep_target, nstep_train,
                                  prefix_steps=burn_in_timesteps)
        if getattr(self, "graph_learner_step", False):
            self.history_buffer.static_batches = True      # batches in the same buffers every step: a captured graph reads them
        self._actors_last_update_steps = 0
        self.rnn_steps_train = rnn_steps_train or nstep_train
        assert (not burn_in_timesteps) or self.policy.is_recurrent(), \
            "burn_in_timesteps only makes sense for recurrent policies"
        if self.overlap_acting and getattr(self.actors, "_device_mode", False):
            self._ov_setup()
        if self.resume_from:
            from . import resume
            resume.load(self, self.resume_from)
        if getattr(self, "_setup_only", False):
            return
        while not self.train_is_done():
            self.loop_iteration()
        if self._ov is not None:
            # overlapped acting hands the target sync / log row / checkpoint triggered by a feed to the
            # START of the next iteration; the feed that ended the run has no next iteration
            self._ov_apply_deferred()
            if self._full_checkpoint_due:
                self.save_full_checkpoint()

    def setup(self, **train_args):
        """Everything `train(**args)` does before THE LOOP (policies, optimizer,
        history buffer) without entering it; bench.py and tests then drive
        `loop_iteration()` / `learner_step()` themselves."""
        self._setup_only = True
        try:
            self.train(**train_args)
        finally:
            self._setup_only = False
        return self

    def loop_iteration(self):
        """One pass of the while-body of multi_step_trainer.py:245-375.
        Returns True when a learner step was taken."""
        stepped = self._loop_iteration()
        if self._full_checkpoint_due:
            self.save_full_checkpoint()
        return stepped

    # -- acting overlapped with learning ------------------------------------------------
    # The reference's loop is strictly serial: act -> ingest -> sample -> train.  Its
    # own answer to the idle time is process-level asynchrony (async actors that run
    # on weights `actor_update_frequency_steps` old, parallel_history.py's 3-deep
    # prefetch).  The device equivalent is one more HIP stream: the ~40 launch-bound
    # acting vector steps of iteration k+1 (E=256 policy forwards, epsilon-greedy, env
    # step, ingest) fill the gaps of iteration k's large GEMMs instead of running
    # alone for ~18 ms.  Ordering, all by events (no host synchronisation):
    #   gather(k) done          -> ingest(k+1) may overwrite ring slots / touch the tree
    #   acting+ingest(k+1) done -> update_losses(k) (tree), Adam(k) -> actor weight copy,
    #                              sample(k+1)
    # The actor therefore runs on a copy of the weights that is ONE learner step older
    # than the reference's synchronous actor would use, and target-network syncs / log
    # rows triggered by the feed of iteration k+1 are applied at the start of iteration
    # k+1 (where the reference applies them), not when the feed is enqueued.
    def _ov_setup(self):
        ov = self._ov = {}
        ov["stream"] = torch.cuda.current_stream() if self.overlap_acting == "serial" else torch.cuda.Stream()
        ov["act_done"] = torch.cuda.Event()
        ov["gather_done"] = torch.cuda.Event()
        ov["weights_ready"] = torch.cuda.Event()
        obs_space, act_space = self.actors.get_spaces()
        actor_policy = self.create_policy(model_config=self.model_config, observation_space=obs_space,
                                          action_space=act_space, **self.policy_args)
        actor_policy.copy_from(self.policy)
        for p in actor_policy.parameters():
            p.requires_grad_(False)
        ov["actor_policy"] = actor_policy
        self.actors.set_actor_policy(actor_policy)
        ov["weights_ready"].record()
        ov["fed"] = False
        ov["deferred"] = None

    def _ov_feed(self, count):
        """Acting + ingest for `count` transitions on the acting stream; the step
        counters advance now, target sync / logging are handed back as deferred work."""
        ov = self._ov
        main = torch.cuda.current_stream()
        ov["stream"].wait_event(ov["weights_ready"])
        ov["stream"].wait_event(ov["gather_done"])
        self._start_timer("sample_actors")
        with torch.cuda.stream(ov["stream"]):
            samples = self.actors.get_samples(count)
            if samples:
                self._process_new_samples(samples)
                self.history_buffer.update(samples)
            ov["act_done"].record(ov["stream"])
        self._end_timer()
        assert torch.cuda.current_stream() == main
        if not samples:
            return
        from .policy_trainer import _crossed
        before = self.steps
        self.steps += len(samples)
        self.clock.acted += len(samples)
        ov["deferred"] = (_crossed(before, self.steps, self.target_update_freq),
                          _crossed(before, self.steps, self.log_freq))

    def _ov_apply_deferred(self):
        d, self._ov["deferred"] = self._ov["deferred"], None
        if d is None:
            return
        if d[0]:
            self.sync_target()
        if d[1]:
            self._log_checkpoint()
            self._full_checkpoint_due = self.full_checkpoints

    def _pre_update_losses(self):
        """Called by the trainers right before history.update_losses: the priority
        tree must not be touched while the overlapped ingest is still running."""
        if self._ov is not None:
            torch.cuda.current_stream().wait_event(self._ov["act_done"])

    def _loop_iteration_overlapped(self):
        ov = self._ov
        main = torch.cuda.current_stream()
        progress = self.get_train_progress()
        warming_up = self.steps < self.warmup_steps
        env_count = self.actors.get_env_count()
        if not ov["fed"]:
            # prologue (and every time a batch could not be formed): feed in the reference's order
            need = self.history_buffer.needed_feed_count(self.mbatch_size, env_count)
            if need is not None:
                if warming_up:
                    need = max(need, env_count)
                ov["gather_done"].record(main)
                self._ov_feed(need)
        self._ov_apply_deferred()
        ov["fed"] = False
        main.wait_event(ov["act_done"])
        self._start_timer("get_train_data")
        train_data = self.history_buffer.get_train_data(self.mbatch_size, train_progress=progress)
        if self.data_parallel is not None and not self.data_parallel.all_ready(train_data is not None):
            train_data = None
        if train_data is None:
            return False
        ov["gather_done"].record(main)
        self._end_timer()
        if warming_up:
            return False
        # the next iteration's acting + ingest, concurrent with this iteration's training
        need = self.history_buffer.needed_feed_count(self.mbatch_size, env_count)
        if need is not None:
            self._ov_feed(need)
            ov["fed"] = True
        self.learner_step(train_data, self.nstep_train, self.nstep_target, self.burn_in_timesteps,
                          self.rnn_steps_train, self.rnn_bootstrap, self.epochs, self.minibatches)
        # refresh the actor's weight copy (after the acting that is in flight finished with it)
        main.wait_event(ov["act_done"])
        ov["actor_policy"].copy_from(self.policy)
        ov["weights_ready"].record(main)
        if self.lr_anneal not in (False, None):
            anneal_to = 0.0 if self.lr_anneal is True else float(self.lr_anneal)
            self.lr = self.base_lr - progress * (self.base_lr - anneal_to)
            self.set_lr(self.lr)
        self.value_log.log("lr", self.lr, group="train")
        if not self.actor_update_frequency_steps or \
                self.steps - self._actors_last_update_steps >= self.actor_update_frequency_steps:
            self.update_actors()
            self._actors_last_update_steps = self.steps
        self._end_timer()
        return True

    def _loop_iteration(self):
        if self._ov is not None and getattr(self.actors, "_device_mode", False):
            return self._loop_iteration_overlapped()
        progress = self.get_train_progress()
        warming_up = self.steps < self.warmup_steps
        env_count = self.actors.get_env_count()
        need = self.history_buffer.needed_feed_count(self.mbatch_size, env_count)
        if need is not None:
            if warming_up:
                need = max(need, env_count)
            self._sample_and_update_history(need)
        self._start_timer("get_train_data")
        train_data = self.history_buffer.get_train_data(self.mbatch_size, train_progress=progress)
        if self.data_parallel is not None and not self.data_parallel.all_ready(train_data is not None):
            train_data = None           # some shard must feed more: nobody enters the collectives
        if train_data is None:
            return False
        self._end_timer()
        if warming_up:
            return False
        self.learner_step(train_data, self.nstep_train, self.nstep_target, self.burn_in_timesteps,
                          self.rnn_steps_train, self.rnn_bootstrap, self.epochs, self.minibatches)
        if self.lr_anneal not in (False, None):
            anneal_to = 0.0 if self.lr_anneal is True else float(self.lr_anneal)
            self.lr = self.base_lr - progress * (self.base_lr - anneal_to)
            self.set_lr(self.lr)
        self.value_log.log("lr", self.lr, group="train")
        if not self.actor_update_frequency_steps or \
                self.steps - self._actors_last_update_steps >= self.actor_update_frequency_steps:
            self.update_actors()
            self._actors_last_update_steps = self.steps
        self._end_timer()
        return True

    @staticmethod
    def _union_rows(sx, tx):
        """`states` and `target_states` leaves of a gathered batch are two row
        ranges of ONE time-major block (history.py:245-265: rows [0,L) and
        [n,L+n) when overlapped, [0,L) and [L,2L) otherwise).  Returns (a view
        of the rows from the first row of `sx` to the last row of `tx`, the row
        shift of `tx`), or None when the two are not such a pair."""
        if not (isinstance(sx, torch.Tensor) and isinstance(tx, torch.Tensor) and sx.is_cuda
                and sx.dim() >= 3 and sx.shape == tx.shape and sx.dtype == tx.dtype
                and sx.stride() == tx.stride()
                and sx.untyped_storage().data_ptr() == tx.untyped_storage().data_ptr()):
            return None
        row = sx.stride(0) * sx.element_size()
        off = tx.data_ptr() - sx.data_ptr()
        if row <= 0 or off < 0 or off % row:
            return None
        shift, T = off // row, sx.shape[0]
        if shift > T:                                  # a gap between the two ranges
            return None
        return torch.as_strided(sx, (T + shift,) + tuple(sx.shape[1:]), sx.stride(), sx.storage_offset()), shift

    def _prepare_frames(self, train_data):
        """Convert the gathered frame block for the CNN ONCE per learner step.
        Both burn-in passes, the training pass, the target pass and the double-Q
        selection pass all read row ranges of the same (L+n, B) uint8 block; the
        u8 NCHW -> f32 NHWC * scale conversion (cnn.py:44-45, csrc/convert.hip)
        was running once per pass on overlapping rows (4 launches, 12.3 GB of HBM
        traffic per step at config D).  The converted block rides along as the
        extra leaf "x_prepared" of both state trees, so every later row slice /
        flatten applies to it like to "x" (same values, SequentialModel.forward)."""
        if not getattr(self, "prepare_frames_once", True):
            return
        states, targets = train_data["states"], train_data["target_states"]
        for pol in {id(self.policy): self.policy, id(self.target_policy): self.target_policy}.values():
            m = pol.model
            if not hasattr(m.layers[0], "prepare_input") or m.extra_input_layer == 0 or 0 in m.layer_pre_processors:
                return
        sx, tx = states["x"], targets["x"]
        if isinstance(sx, (tuple, list)):
            sx, tx = sx[0], tx[0]
        if not (isinstance(sx, torch.Tensor) and sx.dtype == torch.uint8 and sx.is_contiguous() and tx.is_contiguous()):
            return
        u = self._union_rows(sx, tx)
        if u is None:
            return
        union, shift = u
        layer0 = self.policy.model.layers[0]
        other = self.target_policy.model.layers[0]
        if (layer0.scale, layer0.channels_last) != (other.scale, other.channels_last):
            return
        block = layer0.prepare_input(union)
        if block is None:
            return
        L = sx.shape[0]
        states["x_prepared"] = block[:L]
        targets["x_prepared"] = block[shift:shift + L]

    def _share_online_features(self, train_data, nstep_target):
        """The online network sees almost the same frames twice per learner step:
        `states` in the training pass and `target_states` (the same block shifted
        by n rows, history.py:245-265) in the double-Q selection pass.  When the two
        are views of one gathered block, run the online model's first (stateless,
        conv) layer ONCE over the union of rows, with grad, and hand both passes
        their slice; the values are those of the separate passes."""
        if not getattr(self, "double_q", False) or not getattr(self, "share_online_cnn", True):
            return
        model = self.policy.model
        if len(model.layers) < 2 or model.layers[0].is_recurrent() or 0 in model.layer_pre_processors \
                or model.extra_input_layer == 0:
            return
        prepared = "x_prepared" in train_data["states"] and "x_prepared" in train_data["target_states"]
        key = "x_prepared" if prepared else "x"
        sx, tx = train_data["states"][key], train_data["target_states"][key]
        if isinstance(sx, (tuple, list)):
            sx, tx = sx[0], tx[0]
        u = self._union_rows(sx, tx)
        if u is None or (not prepared and not (sx.is_contiguous() and tx.is_contiguous())):
            return
        union, shift = u
        T, B = sx.shape[0], sx.shape[1]
        if shift != nstep_target or nstep_target >= T:
            return                                     # not the overlapped layout
        rows = union.reshape(((T + shift) * B,) + tuple(sx.shape[2:]))
        feats = model.layers[0](rows, timesteps=1, prepared=True) if prepared else model.layers[0](rows, timesteps=1)
        train_data["states"]["x_features"] = {id(model): feats[:T * B].reshape((T, B) + tuple(feats.shape[1:]))}
        train_data["target_states"]["x_features"] = {
            id(model): feats[shift * B:].detach().reshape((T, B) + tuple(feats.shape[1:]))}
        # one layer further: when layer 1 is the recurrent layer, its input projection
        # (x W_ih^T + b, 40 960 x 3136 x 2048 at config D: a 4 ms GEMM) is also a pure
        # function of these rows — compute it once over the union as well
        layer1 = model.layers[1]
        if getattr(self, "share_online_projection", True) and hasattr(layer1, "project_input") \
                and getattr(layer1, "fused", False) and model.extra_input_layer != 1 and 1 not in model.layer_pre_processors:
            proj = layer1.project_input(feats)
            train_data["states"]["x_projected"] = {id(model): proj[:T * B].reshape(T, B, -1)}
            train_data["target_states"]["x_projected"] = {id(model): proj[shift * B:].detach().reshape(T, B, -1)}

    def learner_step(self, train_data, nstep_train, nstep_target, burn_in_timesteps=0,
                     rnn_steps_train=None, rnn_bootstrap=False, epochs=1, minibatches=1):
        """One pass of multi_step_trainer.py:278-353 over one batch: burn-in,
        flatten, targets, minibatch epochs.  bench.py times exactly this (plus
        sampling and ingest)."""
        rnn_steps_train = rnn_steps_train or nstep_train
        if getattr(self, "_graph_step_ok", None) is not None and self._graph_step_ok(train_data, burn_in_timesteps, epochs, minibatches):
            self._start_timer("train")
            self._learner_step_graphed(train_data, nstep_train, nstep_target, rnn_steps_train, rnn_bootstrap, burn_in_timesteps)
            return
        self._prepare_frames(train_data)
        if burn_in_timesteps:
            train_data = self._burn_in(train_data, burn_in_timesteps, do_target_states=rnn_bootstrap)
        if epochs * minibatches == 1:        # the shared features' graph is consumed by one backward
            self._share_online_features(train_data, nstep_target)
        self._start_timer("calc_target_values")
        train_data = deep_apply(train_data, _flat)
        train_data["targets"] = self.calc_target_values(
            train_data["returns"], train_data["target_states"], train_data["target_masks"],
            nsteps=train_data["nsteps"], timesteps=1 if not rnn_bootstrap else rnn_steps_train)
        self._end_timer()
        self._start_timer("train")
        batch_size = train_data["returns"].shape[0]
        assert batch_size % minibatches == 0
        mini = batch_size // minibatches
        for _ in range(epochs):
            inds = self.get_train_indexes(batch_size, mini, nstep_train)
            for i in range(0, batch_size, mini):
                if minibatches > 1:
                    tinds = torch.as_tensor(inds[i:i + mini], device=train_data["returns"].device)
                    pick = lambda y: deep_apply(y, lambda x: x[tinds])   # noqa: E731
                else:
                    pick = lambda y: y                                     # noqa: E731
                self.train_batch(pick(train_data["states"]), pick(train_data["targets"]),
                                 pick(train_data["policy_outputs"]), pick(train_data["extra_data"]),
                                 rnn_steps_train)
                self.value_log.log("batch_size", mini, group="train")
                self._update_steps_trained(mini)
                self.ts_learner_steps += 1
