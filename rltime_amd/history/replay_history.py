"""Device-resident replay history buffers with the reference's plugin API.

Mirrors (same constructor arguments, method names, return structure and error
behaviour; paths relative to the reference root):

  * rltime/history/history.py:8-335                      History
  * rltime/history/replay_history.py:6-184               ReplayHistoryBuffer
  * rltime/history/prioritized_replay_history.py:10-356  PrioritizedReplayHistoryBuffer

but every transition lives in HBM inside a ``librltime_hip`` replay shard and
``update`` / ``get_train_data`` / ``update_losses`` only enqueue HIP kernels on
the current stream (include/mirl.h).  Nothing here computes on the host: there
is no CPU fallback, and construction fails loudly without a GPU.

Differences a caller can observe (all documented in DESIGN.md):
  * batches are torch tensors already on the device with the dtypes the trainer
    would see after ``make_tensor`` (models/torch/utils.py:95-123): uint8 frames,
    everything else float32, except ``actions`` / ``loss_indices`` (int64);
  * ``update`` wants whole vector steps (one transition per env), which is what
    the synchronous actor emits (acting/actor.py:97-149);
  * ``state_store`` is accepted and ignored: the replay *is* the device store
    the reference's StateStore hook was meant for (history.py:34-37).
"""
import ctypes as C
import random

import numpy as np
import torch

from .. import _lib
from .._lib import lib, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _host_example(state):
    def leaf(v):
        if isinstance(v, torch.Tensor):
            return v.detach().cpu().numpy()
        return np.asarray(v)
    if isinstance(state, dict):
        return {k: _host_example(v) for k, v in state.items()}
    if isinstance(state, (tuple, list)):
        return type(state)(_host_example(v) for v in state)
    return leaf(state)


def global_sampling_rows(mbatch, world, share, quantum=None):
    """Rows of ONE rank's padded batch under exact global sampling (mirl_replay_sample_global) for a
    shard holding `share` = P_r / P_g of the global priority mass: the B_g = mbatch * world strata
    have width P_g / B_g, so a range of mass P_r contains at most ceil(B_g * share) + 1 stratum
    points; + 1 for the rounding of the cumulative sums, rounded up to a multiple of `quantum`
    (default max(4, mbatch // 8): few distinct batch shapes).  Returns (rows, bound)."""
    q = int(quantum) if quantum else max(4, mbatch // 8)
    bound = int(np.ceil(mbatch * world * float(share))) + 2
    return ((bound + q - 1) // q) * q, bound


class _Layout:
    """Shape of one stored ``next_state`` pytree (sequential.py:128-146:
    {"x": obs | tuple(obs, extra...), "layer{i}_state": {} | {hx, cx, initials}})."""

    def __init__(self, state):
        # one transition's pytree as host arrays: enough to re-create the shard on resume
        self.example = _host_example(state)
        x = state["x"]
        if isinstance(x, (tuple, list)):
            self.tuple_obs = True
            self.frame_shape = tuple(x[0].shape)
            self.extra_shapes = [tuple(np.shape(v)) for v in x[1:]]
        else:
            self.tuple_obs = False
            self.frame_shape = tuple(x.shape)
            self.extra_shapes = []
        self.frame_bytes = int(np.prod(self.frame_shape))
        self.extra_sizes = [int(np.prod(s)) for s in self.extra_shapes]
        self.extra_f32 = sum(self.extra_sizes)
        self.layer_keys = [k for k in state if k != "x"]
        self.recurrent = []      # (key, [(name, size), ...])
        self.has_initials = False
        self.state_f32 = 0
        for k in self.layer_keys:
            sub = state[k]
            if not sub:
                continue
            fields = []
            for name, v in sub.items():
                if name == "initials":
                    self.has_initials = True
                    continue
                n = int(np.prod(np.shape(v)))
                fields.append((name, n, tuple(np.shape(v))))
                self.state_f32 += n
            self.recurrent.append((k, fields))


class History:
    """history.py:8-59 — argument handling shared by the device buffers."""

    def __init__(self, nstep_target, nstep_train, prefix_steps=0,
                 discount_function=None, state_store=None, gamma=None):
        self.nstep_target = nstep_target
        self.nstep_train = nstep_train
        self.prefix_steps = prefix_steps
        assert (nstep_target == 1 or discount_function is not None
                or gamma is not None), \
            "History buffer must get a 'discount_function' for nstep_target>1"
        self.discount_function = discount_function
        if gamma is None:
            gamma = getattr(discount_function, "gamma", None)
        if gamma is None:
            if nstep_target > 1:
                raise ValueError(
                    "the device replay evaluates the n-step return on the GPU "
                    "and needs the discount as a number: pass gamma=... or a "
                    "discount_function carrying a .gamma attribute "
                    "(rltime_amd trainers do)")
            gamma = 1.0
        self.gamma = float(gamma)
        self.state_store = state_store


class ReplayHistoryBuffer(History):
    """replay_history.py:6-184 on the device.  Uniform sampling."""

    _MODE = _lib.MODE_UNIFORM

    def __init__(self, size, train_frequency, avoid_episode_crossing=False,
                 num_envs=None, env_base=None, device=None, device_rng=False,
                 keep_policy_outputs=True, env_ring_slack=0, frame_stack_dedup=False, seed=None, **kwargs):
        """frame_stack_dedup (not a reference argument; the reference notes buffer-side
        stacking support as planned, history.py:56-59): True = the observation's leading
        axis is a frame stack produced by the stack-shift contract of
        env_wrappers/common.py:141-178 (an int gives the number of planes explicitly);
        the shard then stores one plane per transition and rebuilds the stacks in the
        gather (include/mirl.h, mirl_replay_config.stack_planes)."""
        super().__init__(**kwargs)
        self._dedup = frame_stack_dedup
        _lib.require_gpu()
        self.size = size
        if train_frequency and float(train_frequency) != int(train_frequency):
            # the reference adds it to the quota as a number (replay_history.py:90-91,
            # :73 int(-quota / train_frequency)); the shard keeps an integer quota
            raise ValueError("train_frequency must be a whole number of trained samples per acted "
                             "sample (got %r)" % (train_frequency,))
        self.train_frequency = train_frequency
        self.avoid_episode_crossing = avoid_episode_crossing
        self._num_envs = num_envs
        self._env_base = env_base
        self._device_index = torch.cuda.current_device() if device is None \
            else torch.device(device).index
        self.device = torch.device("cuda", self._device_index)
        self._device_rng = device_rng
        self._keep_policy = keep_policy_outputs
        self._slack = env_ring_slack
        self._h = None
        self._layout = None
        # key of the device-RNG (Philox) sampling stream; the shard's env_base is mixed in at
        # creation so that ranks draw different streams (exact global sampling undoes it: all
        # ranks must draw the SAME uniforms there)
        self._seed_base = 0x5EED if seed is None else int(seed) & 0x7FFFFFFF
        self._seed = self._seed_base
        self._policy_f32 = 0

    # -- lifetime ----------------------------------------------------------
    def _per_config(self, cfg):
        pass

    def _create(self, layout, num_envs, env_base, policy_f32):
        self._layout = layout
        self._example_state = layout.example
        self._num_envs = num_envs
        self._env_base = env_base
        self._policy_f32 = policy_f32
        if getattr(self, "_global", None) is None and self._seed == self._seed_base:
            self._seed = self._seed_base + (int(env_base or 0) << 32)
        cfg = _lib.ReplayConfig(
            size=self.size, num_envs=num_envs, env_base=env_base,
            frame_bytes=layout.frame_bytes, extra_f32=layout.extra_f32,
            state_f32=layout.state_f32, has_initials=int(layout.has_initials),
            policy_f32=policy_f32, nstep_train=self.nstep_train,
            prefix_steps=self.prefix_steps, nstep_target=self.nstep_target,
            gamma=self.gamma, mode=self._MODE,
            train_frequency=int(self.train_frequency or 0),
            avoid_episode_crossing=int(bool(self.avoid_episode_crossing)),
            overlap=_lib.INT32_MIN, alpha=0.6, beta=0.4, eps=1e-6,
            max_weight_factor=0.9, beta_anneal_mode=0, beta_anneal_to=1.0,
            global_importance_scaling=0, env_ring_slack=self._slack,
            device=self._device_index, acting_priority_init=0, acting_vf_eps=0.0,
            stack_planes=0 if not self._dedup else
            (int(layout.frame_shape[0]) if self._dedup is True else int(self._dedup)))
        self._per_config(cfg)
        h = C.c_void_p()
        check(lib.mirl_replay_create(C.byref(cfg), C.byref(h)), "mirl_replay_create")
        self._h = h
        rows, ov = C.c_int32(), C.c_int32()
        check(lib.mirl_replay_state_rows(h, C.byref(rows), C.byref(ov)))
        self._rows, self._overlapped = rows.value, bool(ov.value)

    def close(self):
        if self._h is not None:
            lib.mirl_replay_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ingest --------------------------------------------------------------
    def update(self, new_samples):
        """history.py:123-176.  ``new_samples`` is the list of per-env sample
        dicts the actor emits (acting_interface.py:83-90); it is regrouped into
        vector steps and written with one batched device copy each."""
        if getattr(new_samples, "ingested", False):     # acting/fast_step.py: the actor wrote the steps itself
            return {}
        if hasattr(new_samples, "vector_steps"):        # acting_interface.DeviceSamples
            if self._h is None:
                pol = new_samples.vector_steps[0].get("policy")
                pf = int(pol.shape[1]) if (pol is not None and self._keep_policy) else 0
                self.configure(new_samples.example_state, new_samples.num_envs,
                               new_samples.env_base, policy_f32=pf)
            for step in new_samples.vector_steps:
                self.update_batch(
                    step["frames"], step["actions"], step["rewards"], step["dones"],
                    extra=step.get("extra"), state=step.get("state"),
                    initials=step.get("initials"),
                    policy=step.get("policy") if self._policy_f32 else None)
            return {}
        samples = list(new_samples)
        i = 0
        while i < len(samples):
            seen = set()
            j = i
            while j < len(samples) and samples[j]["env_id"] not in seen:
                seen.add(samples[j]["env_id"])
                j += 1
            self._update_vector_step(samples[i:j])
            i = j
        return {}

    def _update_vector_step(self, chunk):
        first = chunk[0]["next_state"]
        if self._h is None:
            layout = _Layout(first)
            ids = sorted(s["env_id"] for s in chunk)
            base = ids[0] if self._env_base is None else self._env_base
            n = (ids[-1] - base + 1) if self._num_envs is None else self._num_envs
            qv = chunk[0]["policy_output"].get("qvalues")
            pf = int(np.prod(np.shape(qv))) if (qv is not None and self._keep_policy) else 0
            self._create(layout, n, base, pf)
        lay = self._layout
        env_ids = np.array([s["env_id"] for s in chunk], dtype=np.int32)

        def stack(getter, dtype):
            return torch.from_numpy(np.ascontiguousarray(
                np.stack([np.asarray(getter(s)) for s in chunk]).astype(dtype, copy=False)))

        def obs(s):
            x = s["next_state"]["x"]
            return x[0] if lay.tuple_obs else x

        fields = {"frames": stack(obs, np.uint8)}
        if lay.extra_f32:
            fields["extra"] = stack(lambda s: np.concatenate(
                [np.asarray(v, dtype=np.float32).reshape(-1)
                 for v in s["next_state"]["x"][1:]]), np.float32)
        if lay.state_f32:
            fields["state"] = stack(lambda s: np.concatenate(
                [np.asarray(s["next_state"][k][name], dtype=np.float32).reshape(-1)
                 for k, fl in lay.recurrent for name, _, _ in fl]), np.float32)
        if lay.has_initials:
            key = lay.recurrent[0][0] if lay.recurrent else lay.layer_keys[0]
            fields["initials"] = stack(
                lambda s: s["next_state"][key]["initials"], np.float32)
        fields["actions"] = stack(lambda s: s["policy_output"]["actions"], np.int32)
        if self._policy_f32:
            fields["policy"] = stack(
                lambda s: np.asarray(s["policy_output"]["qvalues"]).reshape(-1), np.float32)
        fields["rewards"] = stack(lambda s: s["reward"], np.float32)
        fields["dones"] = stack(lambda s: s["done"], np.uint8)
        dev = {k: v.to(self.device, non_blocking=False) for k, v in fields.items()}
        self.update_batch(env_ids=env_ids, **dev)

    def update_batch(self, frames, actions, rewards, dones, extra=None, state=None,
                     initials=None, policy=None, env_ids=None, transient=False, newest_plane_only=False):
        """Fast path: one vector step of device tensors with leading dim K
        (frames u8 [K, ...], actions i32 [K], rewards f32 [K], dones u8 [K],
        extra/state/policy f32 [K, n], initials f32 [K]).  ``env_ids`` (host
        int32 array) defaults to env_base .. env_base+K-1."""
        if self._h is None:
            raise _lib.MirlError(
                "update_batch before the shard exists: call configure() first")
        K = int(frames.shape[0])
        keep = [frames, actions, rewards, dones, extra, state, initials, policy]
        plane_ptr, plane_stride = None, 0
        if newest_plane_only:
            # de-duplicated storage fed by a producer that honours the stack-shift contract (the device
            # actor): only the newest plane of every (K, P, h, w) stack is read, nothing is verified
            assert self._dedup and frames.dim() >= 3 and frames.is_contiguous()
            per_plane = frames[0, 0].numel()
            plane_ptr = frames.data_ptr() + (frames.shape[1] - 1) * per_plane
            plane_stride = frames.shape[1] * per_plane
        for t in keep:
            if t is not None:
                assert t.is_cuda and t.is_contiguous()
        assert frames.dtype == torch.uint8 and actions.dtype == torch.int32
        assert rewards.dtype == torch.float32 and dones.dtype == torch.uint8
        ids = None
        if env_ids is not None:
            ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        arg = _lib.Ingest(
            count=K, env_ids_host=_lib.np_ptr(ids) if ids is not None else None,
            frames=_ptr(frames) if plane_ptr is None else C.c_void_p(plane_ptr), extra=_ptr(extra), state=_ptr(state),
            initials=_ptr(initials), actions=_ptr(actions), policy=_ptr(policy),
            rewards=_ptr(rewards), dones=_ptr(dones), newest_plane_only=1 if newest_plane_only else 0,
            frames_stride=plane_stride)
        check(lib.mirl_replay_ingest(self._h, C.byref(arg), _stream()), "mirl_replay_ingest")
        # the kernels read the payload asynchronously: tie its lifetime to the stream
        # (transient=True: the caller owns long-lived buffers it only rewrites in stream order)
        if not transient:
            s = torch.cuda.current_stream()
            for t in keep:
                if t is not None:
                    t.record_stream(s)

    # -- a whole acting rollout ahead of its device side (mirl_replay_ingest_plan / _planned) ---------------
    def supports_planned_ingest(self):
        """The fused one-launch ingest covers this shard (no de-duplicated storage, no acting-time priority init)."""
        return self._h is not None and not getattr(self, "_dedup", False) and not getattr(self, "_acting_priority_init", False)

    def plan_ingest(self, steps, count, env_ids=None):
        """Host bookkeeping of `steps` vector steps of `count` transitions (history.py:123-176 and the PER hooks: all of
        it data-independent) + one stream-ordered upload of their op lists.  The device side of step k follows with
        ingest_planned(k, ...) — from a captured graph, if the caller wants."""
        ids = np.ascontiguousarray(env_ids, dtype=np.int32) if env_ids is not None else None
        check(lib.mirl_replay_ingest_plan(self._h, int(steps), int(count), _lib.np_ptr(ids) if ids is not None else None,
                                          _stream()), "mirl_replay_ingest_plan")

    def ingest_planned(self, step, frames, actions, rewards, dones, extra=None, state=None, initials=None, policy=None):
        """Device side of step `step` of the current plan: same payload contract as update_batch; the buffers must stay
        valid (and are re-read) on every replay of a graph this call was captured into."""
        K = int(frames.shape[0])
        for t in (frames, actions, rewards, dones, extra, state, initials, policy):
            if t is not None:
                assert t.is_cuda and t.is_contiguous()
        assert frames.dtype == torch.uint8 and actions.dtype == torch.int32
        assert rewards.dtype == torch.float32 and dones.dtype == torch.uint8
        arg = _lib.Ingest(count=K, env_ids_host=None, frames=_ptr(frames), extra=_ptr(extra), state=_ptr(state),
                          initials=_ptr(initials), actions=_ptr(actions), policy=_ptr(policy), rewards=_ptr(rewards),
                          dones=_ptr(dones), newest_plane_only=0, frames_stride=0)
        check(lib.mirl_replay_ingest_planned(self._h, int(step), C.byref(arg), _stream()), "mirl_replay_ingest_planned")

    def prime_stack(self, obs):
        """De-duplicated storage + newest-plane ingest: hand over the observation block the env's
        reset returned ((E, P, h, w) uint8) so that the first transitions' stacks can reach back to it."""
        assert self._dedup and self._h is not None and obs.is_contiguous()
        per_plane = obs[0, 0].numel()
        check(lib.mirl_replay_prime_stack(self._h, C.c_void_p(obs.data_ptr() + (obs.shape[1] - 1) * per_plane),
                                          obs.shape[1] * per_plane, _stream()), "mirl_replay_prime_stack")

    def configure(self, example_state, num_envs, env_base=0, policy_f32=0):
        """Create the shard up-front from one example ``next_state`` pytree
        (the reference learns shapes from the first sample; the fast path needs
        them before the first update_batch)."""
        if self._h is None:
            self._create(_Layout(example_state), num_envs, env_base, policy_f32)

    # -- sampling --------------------------------------------------------------
    def needed_feed_count(self, mbatch_size, num_envs):
        """replay_history.py:62-75."""
        if self._h is None:
            if not self.train_frequency:
                return 0
            return num_envs       # quota is 0 -> max(int(0), num_envs)
        out = C.c_int64()
        check(lib.mirl_replay_needed_feed_count(self._h, mbatch_size, num_envs, C.byref(out)))
        return None if out.value < 0 else out.value

    def _current_beta(self, train_progress):
        return 0.0

    def _draw_host_rng(self, mbatch):
        """replay_history.py:118 — np.random.choice(total_available, mbatch),
        only evaluated when there are enough start positions (:110-114)."""
        total = C.c_int64()
        check(lib.mirl_replay_uniform_total(self._h, C.byref(total)))
        if total.value < mbatch:
            return None
        return np.ascontiguousarray(np.random.choice(total.value, mbatch), dtype=np.int64)

    def get_train_data(self, mbatch_size, train_progress=None):
        """replay_history.py:173-184 -> history.py:203-286.  Returns None when
        more samples must be fed first."""
        B = mbatch_size
        if getattr(self, "_global", None) is not None:
            return self._get_train_data_global(B, train_progress)
        if self._h is None:
            return None
        rng = None if self._device_rng else self._draw_host_rng(B)
        dev = self.device
        new = self._buffer
        slot = new("slot", (B,), torch.int32)
        env = new("env", (B,), torch.int32)
        start = new("start", (B,), torch.int64)
        loss_start = new("loss_start", (B,), torch.int64)
        weight = new("weight", (B,), torch.float32)
        stats = new("stats", (2,), torch.float64, zero=True)
        self._seed += 1
        rc = check(lib.mirl_replay_sample(
            self._h, B, -1.0 if train_progress is None else float(train_progress),
            _lib.np_ptr(rng) if rng is not None else None, self._seed,
            _ptr(slot), _ptr(env), _ptr(start), _ptr(loss_start), _ptr(weight), _ptr(stats), _stream()),
            "mirl_replay_sample")
        if rc == _lib.MIRL_NEED_MORE:
            return None
        self.last_beta = self._current_beta(train_progress)
        self.last_sample = {"slot": slot, "env": env, "start": start, "weight": weight,
                            "stats": stats, "loss_start": loss_start}
        return self._gather(B, env, start, weight, loss_start)

    def _buffer(self, name, shape, dtype, zero=False):
        """An output tensor of a sample / gather call.  static_batches (set by a trainer that replays its learner step
        from a captured HIP graph, training/torch_trainer.py): the SAME storage on every call — a batch then lives until
        the next get_train_data call, which is what the synchronous loop needs (multi_step_trainer.py:245-375)."""
        if not getattr(self, "static_batches", False):
            return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
        cache = self.__dict__.setdefault("_static_out", {})
        key = (name, tuple(shape), dtype)
        t = cache.get(key)
        if t is None:
            t = cache[key] = torch.zeros(shape, dtype=dtype, device=self.device)
        elif zero:
            t.zero_()
        return t

    def _gather(self, B, env, start, weight, loss_start=None):
        lay, dev = self._layout, self.device
        L = self.nstep_train + self.prefix_steps
        R = self._rows
        per = self._MODE == _lib.MODE_PER
        new = self._buffer
        frames = new("frames", (R, B) + tuple(lay.frame_shape), torch.uint8)
        extra = new("extra", (R, B, lay.extra_f32), torch.float32) if lay.extra_f32 else None
        state = new("state", (R, B, lay.state_f32), torch.float32) if lay.state_f32 else None
        initials = new("initials", (R, B), torch.float32) if lay.has_initials else None
        returns = new("returns", (L, B), torch.float32)
        nsteps = new("nsteps", (L, B), torch.float32)
        masks = new("masks", (L, B), torch.float32)
        actions = new("actions", (L, B), torch.int64)
        policy = new("policy", (L, B, self._policy_f32), torch.float32) if self._policy_f32 else None
        weights = new("weights", (L, B), torch.float32) if per else None
        loss_idx = new("loss_idx", (L, B, 2), torch.int64) if per else None
        out = _lib.Batch(
            frames=_ptr(frames), extra=_ptr(extra), state=_ptr(state),
            initials=_ptr(initials), returns=_ptr(returns), nsteps=_ptr(nsteps),
            masks=_ptr(masks), actions=_ptr(actions), policy=_ptr(policy),
            weights=_ptr(weights), loss_indices=_ptr(loss_idx))
        check(lib.mirl_replay_gather(
            self._h, B, _ptr(env), _ptr(start), _ptr(loss_start), _ptr(weight), C.byref(out), _stream()),
            "mirl_replay_gather")

        if self._overlapped:
            n = self.nstep_target
            cut_s = lambda t: t[:L]           # noqa: E731  history.py:262-263
            cut_t = lambda t: t[n:]           # noqa: E731  history.py:264-265
        else:
            cut_s = lambda t: t[:L]           # noqa: E731
            cut_t = lambda t: t[L:]           # noqa: E731

        def build(cut):
            x = cut(frames)
            if lay.tuple_obs:
                parts, at = [x], 0
                for size, shape in zip(lay.extra_sizes, lay.extra_shapes):
                    e = cut(extra)[..., at:at + size]
                    parts.append(e.reshape(e.shape[:2] + shape))
                    at += size
                x = tuple(parts)
            tree = {"x": x}
            at = 0
            rec = dict(lay.recurrent)
            for k in lay.layer_keys:
                if k not in rec:
                    tree[k] = {}
                    continue
                sub = {}
                for name, size, shape in rec[k]:
                    v = cut(state)[..., at:at + size]
                    sub[name] = v.reshape(v.shape[:2] + shape) if shape != (size,) else v
                    at += size
                if lay.has_initials:
                    sub["initials"] = cut(initials)
                tree[k] = sub
            return tree

        batch = {
            "returns": returns, "nsteps": nsteps, "target_masks": masks,
            "policy_outputs": {"actions": actions},
            "states": build(cut_s), "target_states": build(cut_t),
        }
        if policy is not None:
            batch["policy_outputs"]["qvalues"] = policy
        batch["extra_data"] = {"importance_weights": weights,
                               "loss_indices": loss_idx} if per else {}
        return batch

    def update_losses(self, indices, losses):
        """history.py:332-335 — nothing to do for uniform replay."""

    # -- introspection (tests / logging) ----------------------------------------
    def stats(self):
        v = [C.c_int64() for _ in range(5)]
        check(lib.mirl_replay_stats(self._h, *[C.byref(x) for x in v]))
        keys = ("total_items", "active_sequences", "train_quota",
                "tree_capacity", "n_slots")
        return dict(zip(keys, (x.value for x in v)))

    def save(self, path):
        """Snapshot the whole shard (bookkeeping + device arrays + trees) to `path`."""
        check(lib.mirl_replay_save(self._h, str(path).encode()), "mirl_replay_save")

    def load(self, path, example_state=None, num_envs=None, env_base=0, policy_f32=0):
        """Restore a snapshot written by an identically configured buffer.  A
        buffer that has not seen data yet needs the transition layout
        (`example_state`, `num_envs`) to create its shard first."""
        if self._h is None:
            if example_state is None or num_envs is None:
                raise _lib.MirlError("load() on an empty buffer needs example_state and num_envs")
            self.configure(example_state, num_envs, env_base, policy_f32)
        check(lib.mirl_replay_load(self._h, str(path).encode()), "mirl_replay_load")

    def profile(self, enable):
        """Frame-gather kernel timing (HIP events on the launch stream):
        returns (launches, total_ms) accumulated since the last call."""
        n, ms = C.c_int64(), C.c_double()
        check(lib.mirl_replay_profile(self._h, int(enable), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    @property
    def train_quota(self):
        return self.stats()["train_quota"] if self._h is not None else 0

    @train_quota.setter
    def train_quota(self, value):
        check(lib.mirl_replay_set_train_quota(self._h, int(value)))


class PrioritizedReplayHistoryBuffer(ReplayHistoryBuffer):
    """prioritized_replay_history.py:10-356 on the device: sum-tree (and
    min-tree) in HBM, stratified descent with the reference's exact NumPy-2
    scalar promotion (csrc/np_emul.h), overlapped-sequence priorities."""

    _MODE = _lib.MODE_PER

    def __init__(self, alpha=0.6, beta=0.4, beta_anneal=False, eps=1e-6,
                 overlap=None, max_weight_factor=0.9,
                 global_importance_scaling=False, acting_priority_init=False,
                 acting_priority_vf_eps=None, **kwargs):
        """acting_priority_init / acting_priority_vf_eps are NOT reference arguments:
        the reference notes acting-time priority initialisation as missing
        (prioritized_replay_history.py:33-36).  When enabled, every ingested
        transition initialises the priority input of the transition n steps before
        it from the STORED q-values (include/mirl.h, mirl_replay_config) instead
        of leaving it at the constant maximum 1.0 until the learner first sees it."""
        super().__init__(**kwargs)
        self._acting_priority_init = bool(acting_priority_init)
        self._acting_vf_eps = float(acting_priority_vf_eps or 0.0)
        if self._acting_priority_init and not self._keep_policy:
            raise ValueError("acting_priority_init needs the stored q-values: keep_policy_outputs must stay True")
        self._alpha, self._beta, self._beta_anneal = alpha, beta, beta_anneal
        self._eps, self._overlap = eps, overlap
        self._max_weight_factor = max_weight_factor
        self._global_importance_scaling = global_importance_scaling
        if overlap is not None and overlap >= 0:
            assert overlap < self.nstep_train, "Overlap must be < nstep_train"

    # -- exact global sampling over env-sharded replays ------------------
    def enable_global_sampling(self, data_parallel, row_quantum=None):
        """Sample like ONE tree over the concatenation of all ranks' shards instead of
        per-shard proportionally: every rank draws the same Philox uniforms for the
        B_global = mbatch * world strata of the global priority mass and takes the
        strata that fall into its own cumulative range (include/mirl.h,
        mirl_replay_sample_global).  The per-rank count is data-dependent, but bounded:
        a range of mass P_r holds at most ceil(B_global * P_r / P_g) + 1 strata of width
        P_g / B_global.  Every call therefore sizes this rank's batch from the table of
        shard totals it has just exchanged — rows = that bound + 1 for rounding, rounded
        up to a multiple of `row_quantum` (default max(4, mbatch // 8): few distinct batch
        shapes) — so NO stratum can be dropped, whatever the imbalance between the
        shards.  Padding rows carry weight 0 and no loss index, and the real rows'
        importance weights are scaled by rows / mbatch so that the all-reduced mean over
        the padded per-rank batches equals the mean over the B_global global rows.
        Reading the (world, 2) table costs one small device->host copy per call; the
        kernel still counts strata beyond `rows` and `check_dropped_strata()` (called
        at every log interval) raises if that counter is ever non-zero."""
        if not self._device_rng:
            raise ValueError("global sampling draws its uniforms on the device: construct the buffer with device_rng=True")
        self._global = (data_parallel, row_quantum)
        self._seed = self._seed_base + (self._seed - self._seed_base) % (1 << 32)    # same Philox key on every rank
        self._dropped_dev = None
        self.global_rows_log = []          # (rows, strata bound) of the most recent calls (tests / logging)

    def global_rows_for(self, B, share):
        """Rows of this rank's padded batch for a shard holding `share` = P_r / P_g of
        the global priority mass (see enable_global_sampling)."""
        dp, quantum = self._global
        return global_sampling_rows(B, dp.world, share, quantum)

    def check_dropped_strata(self):
        """Strata that did not fit a padded batch since the last check (one host read;
        the trainer calls it at log intervals).  Must be 0 by construction: raises."""
        if getattr(self, "_dropped_dev", None) is None:
            return 0
        n = int(self._dropped_dev.item())
        if n:
            raise _lib.MirlError("global sampling dropped %d strata: a shard owned more strata than its padded batch "
                                 "holds (the sample would be biased) — rows are sized from the exchanged shard totals, "
                                 "so this means the totals changed between the exchange and the draw" % n)
        return 0

    def _get_train_data_global(self, B, train_progress):
        dp, _ = self._global
        R, rank = dp.world, dp.rank
        # whether THIS shard can form a batch is host bookkeeping; all ranks agree on it BEFORE
        # any collective is entered (a rank that returned None while its peer went on to the
        # next all-reduce would leave mismatched collective sequences)
        ready = C.c_int32(0)
        if self._h is not None:
            check(lib.mirl_replay_sample_ready(self._h, B, C.byref(ready)), "mirl_replay_sample_ready")
        if not dp.all_ready(bool(ready.value)):
            if self._h is not None:
                check(lib.mirl_replay_sample_skip(self._h, B), "mirl_replay_sample_skip")
                self._seed += 1
            return None
        dev = self.device
        root = torch.empty(2, dtype=torch.float64, device=dev)
        check(lib.mirl_replay_tree_root(self._h, _ptr(root), _stream()), "mirl_replay_tree_root")
        shard = dp.exchange_rows(root).contiguous()                 # (R, 2): sum of priorities, active sequences
        table = shard.cpu()                                         # the one host read of this path
        share = float(table[rank, 0]) / float(table[:, 0].sum())
        rows, bound = self.global_rows_for(B, share)
        self.global_rows_log = (self.global_rows_log + [(rows, bound)])[-64:]
        slot = torch.empty(rows, dtype=torch.int32, device=dev)
        env = torch.empty(rows, dtype=torch.int32, device=dev)
        start = torch.empty(rows, dtype=torch.int64, device=dev)
        loss_start = torch.empty(rows, dtype=torch.int64, device=dev)
        raw = torch.empty(rows, dtype=torch.float64, device=dev)
        stratum = torch.empty(rows, dtype=torch.int32, device=dev)
        stats = torch.zeros(4, dtype=torch.float64, device=dev)
        self._seed += 1
        rc = check(lib.mirl_replay_sample_global(
            self._h, B, B * R, rows, rank, R, _ptr(shard), float(train_progress or 0.0), self._seed,
            _ptr(slot), _ptr(env), _ptr(start), _ptr(loss_start), _ptr(raw), _ptr(stratum), _ptr(stats), _stream()),
            "mirl_replay_sample_global")
        if rc == _lib.MIRL_NEED_MORE:                               # cannot happen after the agreed readiness check
            raise _lib.MirlError("mirl_replay_sample_global disagreed with mirl_replay_sample_ready")
        if self._dropped_dev is None:
            self._dropped_dev = torch.zeros((), dtype=torch.float64, device=dev)
        self._dropped_dev += stats[3]
        self.last_beta = self._current_beta(train_progress)
        top = dp.exchange_rows(stats[1:2]).max()                    # batch max over ALL ranks (prioritized_replay_history.py:353-354)
        weight = (raw / top * (float(rows) / float(B))).to(torch.float32)
        self.last_sample = {"slot": slot, "env": env, "start": start, "weight": weight, "stats": stats,
                            "loss_start": loss_start, "stratum": stratum, "raw": raw, "shard": shard, "global": True,
                            "rows": rows}
        return self._gather(rows, env, start, weight, loss_start)

    def _per_config(self, cfg):
        cfg.alpha, cfg.beta, cfg.eps = self._alpha, self._beta, self._eps
        cfg.max_weight_factor = self._max_weight_factor
        cfg.overlap = _lib.INT32_MIN if self._overlap is None else int(self._overlap)
        cfg.global_importance_scaling = int(bool(self._global_importance_scaling))
        cfg.acting_priority_init = int(self._acting_priority_init)
        cfg.acting_vf_eps = self._acting_vf_eps
        if self._beta_anneal is False or self._beta_anneal is None:
            cfg.beta_anneal_mode = 0
        elif self._beta_anneal is True:
            cfg.beta_anneal_mode = 1
        else:
            cfg.beta_anneal_mode, cfg.beta_anneal_to = 2, float(self._beta_anneal)

    def _current_beta(self, train_progress):
        """prioritized_replay_history.py:287-288 (general/utils.py:85-103)."""
        from rltime_amd.general.utils import anneal_value
        return anneal_value(self._beta, train_progress or 0.0, self._beta_anneal, 1.0)

    def _draw_host_rng(self, mbatch):
        """prioritized_replay_history.py:238 — random.random() once per stratum,
        consumed even when the call then returns None (:284 precedes :295)."""
        return np.array([random.random() for _ in range(mbatch)], dtype=np.float64)

    def update_losses(self, indices, losses):
        """prioritized_replay_history.py:243-279.  ``indices`` (M, 2) int64 and
        ``losses`` (M,) float32; device tensors stay on the device (no host
        round trip), numpy arrays are uploaded."""
        if self._h is None:
            return
        if not isinstance(indices, torch.Tensor):
            indices = torch.from_numpy(np.ascontiguousarray(indices, dtype=np.int64))
        if not isinstance(losses, torch.Tensor):
            losses = torch.from_numpy(np.ascontiguousarray(losses, dtype=np.float32))
        indices = indices.to(self.device, torch.int64).reshape(-1, 2).contiguous()
        losses = losses.detach().to(self.device, torch.float32).reshape(-1).contiguous()
        assert indices.shape[0] == losses.shape[0]
        check(lib.mirl_replay_update_losses(
            self._h, losses.shape[0], _ptr(indices), _ptr(losses), _stream()),
            "mirl_replay_update_losses")
        s = torch.cuda.current_stream()
        indices.record_stream(s)
        losses.record_stream(s)

    # -- test hooks ----------------------------------------------------------------
    def tree_nodes(self):
        cap = self.stats()["tree_capacity"]
        v = np.zeros(2 * cap, dtype=np.float64)
        k = np.zeros(2 * cap, dtype=np.uint8)
        m = np.zeros(2 * cap, dtype=np.float64)
        check(lib.mirl_replay_tree_nodes(self._h, _lib.np_ptr(v), _lib.np_ptr(k), _lib.np_ptr(m)))
        return v, k, m

    def free_slots(self):
        n = C.c_int64()
        check(lib.mirl_replay_free_slots(self._h, None, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int32)
        check(lib.mirl_replay_free_slots(self._h, _lib.np_ptr(out), C.byref(n)))
        return out[:n.value]

    def slot_table(self):
        n = self.stats()["n_slots"]
        e = np.zeros(n, dtype=np.int32)
        b = np.zeros(n, dtype=np.int64)
        check(lib.mirl_replay_slot_table(self._h, _lib.np_ptr(e), _lib.np_ptr(b)))
        return e, b

    def env_meta(self):
        f = np.zeros(self._num_envs, dtype=np.int64)
        c = np.zeros(self._num_envs, dtype=np.int64)
        check(lib.mirl_replay_env_meta(self._h, _lib.np_ptr(f), _lib.np_ptr(c)))
        return f, c
