"""Synchronous vectorised actor (reference rltime/acting/actor.py:10-149).

Two output modes:
  * device=False: the reference behaviour — per-env sample dicts (numpy);
  * device=True : DeviceSamples — policy forward, epsilon-greedy, env step and
    replay ingest all stay on the GPU with no host round trip per step
    (SURVEY.md section 8(f) item 1)."""
import numpy as np
import torch

from .acting_interface import ActingInterface, DeviceSamples
from rltime_amd.general.type_registry import get_registered_type
from rltime_amd.general.utils import deep_apply, quiet_gc


class GraphedStep:
    """One acting vector step minus the environment, as two HIP graphs that share
    static buffers (reference order of operations, actor.py:108-128):

      graph A  (after env.step):  make_input_state (recurrent-state reset on
               `dones`) -> pack the stored next_state -> policy forward ->
               epsilon-greedy -> actions for the next env step
      graph B  (first step of a get_samples call): only the forward +
               epsilon-greedy on graph A's state, so that the first action after a
               learner update is chosen with the CURRENT weights like the
               reference's loop does.

    At E=256 the eager version is ~60 launch-bound kernels and ~0.5 ms of host
    work per step; a replay is two small input copies and one graph launch.  The
    recurrent carry lives in static buffers the captured kernels read and write,
    so it survives the learner's forwards re-binding `layer.last_state`."""

    def __init__(self, actor, obs, dones):
        pol, self.actor = actor._policy, actor
        dev = pol.device()
        self.obs, self.dones = obs.clone(), dones.clone()
        self.eps = torch.zeros((), dtype=torch.float64, device=dev)
        self.rec = [layer for layer in pol.model.layers if layer.is_recurrent()]
        self.carry = [tuple(t.clone() for t in layer.last_state) for layer in self.rec]
        saved = [tuple(t.clone() for t in pair) for pair in self.carry]

        def head():
            for layer, pair in zip(self.rec, self.carry):
                layer.last_state = pair
            states = pol.make_input_state(self.obs, self.dones)
            return states, _pack_state(states)

        def tail(states):
            actions, qvalues = actor._select(states, self.eps)
            for layer, (h, c) in zip(self.rec, self.carry):
                h.copy_(layer.last_state[0])
                c.copy_(layer.last_state[1])
            return actions, qvalues

        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):          # warm-up outside capture (MIOpen find, hipBLASLt workspaces)
            for _ in range(3):
                states, _ = head()
                tail(states)
        cur.wait_stream(side)
        self.example = deep_apply(states, lambda x: x[0].cpu().numpy())
        for pair, keep in zip(self.carry, saved):            # the warm-ups advanced the carry
            pair[0].copy_(keep[0])
            pair[1].copy_(keep[1])
        self.graph_a = torch.cuda.CUDAGraph()
        # thread_local: RCCL's watchdog thread polls events concurrently in multi-GPU
        # runs; it must not invalidate this thread's capture
        with quiet_gc(), torch.cuda.graph(self.graph_a, capture_error_mode="thread_local"):
            self.states, self.fields = head()
            self.next_actions, self.next_q = tail(self.states)
        self.graph_b = torch.cuda.CUDAGraph()
        with quiet_gc(), torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), capture_error_mode="thread_local"):
            self.b_actions, self.b_q = tail(self.states)

    def after_env_step(self, obs, dones, eps):
        self.obs.copy_(obs)
        self.dones.copy_(dones)
        self.eps.fill_(eps)
        self.graph_a.replay()
        fields = {k: v.clone() for k, v in self.fields.items() if k != "frames"}
        fields["frames"] = obs                    # the env's own tensor; the static copy is overwritten next step
        return fields, self.next_actions.clone(), self.next_q.clone()

    def reselect(self, eps):
        """Re-run forward + exploration on the current state with the current weights."""
        self.eps.fill_(eps)
        self.graph_b.replay()
        return self.b_actions.clone(), self.b_q.clone()


class Actor(ActingInterface):
    def __init__(self, vec_env, exploration_config=None, base_env_id=0, total_env_ids=None, device=False,
                 use_graph=False):
        self._vec_env = vec_env
        self._num_envs = vec_env.num_envs
        self._base_env_id = base_env_id
        self._env_ids = list(range(base_env_id, base_env_id + self._num_envs))
        total = total_env_ids or self._num_envs
        if exploration_config:
            cls = get_registered_type("exploration", exploration_config["type"])
            self._exploration = cls(**exploration_config.get("args"), total_actors=total)
        else:
            self._exploration = None
        self._policy = None
        self._progress = 0
        self._device_mode = device
        self._use_graph = use_graph and device
        self._graphed = None
        self._fast = None              # acting/fast_step.FastActingStep when the policy shape allows it
        self._sink = None              # device replay the fast path ingests into directly (set_sink)
        self.clip_rewards = False      # sign-clip rewards before they are stored (policy_trainer.py:252-254)
        self._tracker = None
        super().__init__(vec_env.observation_space, vec_env.action_space)

    def get_env_count(self):
        return self._num_envs

    def update_state(self, progress, policy_state=None):
        self._progress = progress
        if policy_state is not None:
            self._policy.load_state(policy_state)

    def set_actor_policy(self, actor_policy):
        """actor.py:78-89: reset the envs and build the first input state."""
        self._policy = actor_policy
        obs = self._vec_env.reset()
        self._reset_obs = obs
        self._fast = None
        self.last_state = actor_policy.make_input_state(obs, np.array([True] * self._num_envs))

    def set_sink(self, history, clip_rewards=False):
        """Let the device actor write its vector steps straight into `history` (a device
        replay with update_batch): get_samples then returns an already-ingested summary
        (fast_step.IngestedSamples) and History.update is a no-op for it."""
        new = history if hasattr(history, "update_batch") else None
        if new is not getattr(self, "_sink", None) and self._fast:
            self._fast.forget_rollouts()     # captured rollouts hold the previous sink's device structures
        self._sink = new
        self.clip_rewards = bool(clip_rewards)

    def close(self):
        self._vec_env.close()

    # -- resume (training/resume.py) -------------------------------------------
    def get_state(self):
        """Recurrent carry, last input state and env generator: what the next
        get_samples call continues from."""
        rec = [layer for layer in self._policy.model.layers if layer.is_recurrent()]
        clone = lambda t: deep_apply(t, lambda x: x.detach().clone().cpu() if isinstance(x, torch.Tensor) else x)  # noqa: E731
        if self._graphed is not None:
            carry = [tuple(t.detach().clone().cpu() for t in pair) for pair in self._graphed.carry]
            last = clone(self._graphed.states)
        else:
            carry = [None if layer.last_state is None else tuple(t.detach().clone().cpu() for t in layer.last_state)
                     for layer in rec]
            last = clone(self.last_state)
        env = self._vec_env.get_state() if hasattr(self._vec_env, "get_state") else None
        return {"progress": self._progress, "carry": carry, "last_state": last, "env": env,
                "fast": self._fast.get_state() if self._fast else None,
                "tracker": self._tracker.get_state() if self._tracker is not None else None}

    def set_state(self, state):
        dev = self._policy.device()
        rec = [layer for layer in self._policy.model.layers if layer.is_recurrent()]
        for layer, pair in zip(rec, state["carry"]):
            layer.last_state = None if pair is None else tuple(t.to(dev) for t in pair)
        self.last_state = deep_apply(state["last_state"], lambda x: x.to(dev) if isinstance(x, torch.Tensor) else x)
        self._progress = state["progress"]
        self._graphed = None                   # the acting graph is re-captured from this state
        self._fast_restore = state.get("fast")
        self._tracker_restore = state.get("tracker")
        if self._fast and self._fast_restore is not None:
            self._fast.set_state(self._fast_restore)
            self._fast_restore = None
        elif self._fast_restore is None and any(pair is not None for pair in state["carry"]):
            # a checkpoint written by the generic device path carries no fused-step buffers: the fused step cannot
            # continue from it (its carry / last observation live in its own static buffers) — this run takes the
            # generic path, which continues from `carry` / `last_state`
            self._fast = False
        if self._tracker is not None and self._tracker_restore is not None:
            self._tracker.set_state(self._tracker_restore)
            self._tracker_restore = None
        if state["env"] is not None and hasattr(self._vec_env, "set_state"):
            self._vec_env.set_state(state["env"])

    def get_samples(self, min_samples):
        """actor.py:97-149."""
        iters = (max(1, min_samples) + self._num_envs - 1) // self._num_envs
        return self._device_steps(iters) if self._device_mode else self._host_steps(iters)

    def _host_steps(self, iters):
        samples = []
        for _ in range(iters):
            pred = self._policy.actor_predict(self.last_state, timesteps=1)
            exp_info = None
            if self._exploration is not None:
                pred["actions"], exp_info = self._exploration.remap_actions(
                    pred["actions"], self._env_ids, self._action_space, self._progress)
            obs, rewards, dones, infos = self._vec_env.step(pred["actions"])
            states = self._policy.make_input_state(obs, np.array(dones))
            host = deep_apply(states, lambda x: x.cpu().numpy() if isinstance(x, torch.Tensor) else x)
            for i in range(self._num_envs):
                info = infos[i]
                if exp_info is not None:
                    info["exploration"] = deep_apply(exp_info, lambda x: x[i])
                samples.append(self._create_sample(
                    deep_apply(pred, lambda x: x[i]), deep_apply(host, lambda x: x[i]),
                    rewards[i], dones[i], info, self._env_ids[i]))
            self.last_state = states
        return samples

    def _act_eager(self, state):
        eps = None
        if self._exploration is not None:
            eps = torch.as_tensor(self._exploration._get_eps(self._progress), dtype=torch.float64,
                                  device=self._policy.device())
        return self._select(state, eps)

    def _select(self, state, eps):
        """Policy forward -> greedy action -> epsilon-greedy remap for one vector step
        (actor.py:108-122), everything after the network in ONE launch
        (csrc/acting.hip k_actor_head) when the policy exposes its raw head outputs.
        `eps`: 0-dim float64 device tensor holding the base epsilon (None = greedy)."""
        import ctypes as C
        pol = self._policy
        expl = self._exploration
        fused = getattr(self, "fused_head", True) and hasattr(pol, "actor_head_raw") and pol.is_cuda() \
            and (expl is None or hasattr(expl, "_device_exponents"))
        if not fused:
            pred = pol.actor_predict(state, timesteps=1, as_numpy=False)
            actions = pred["actions"]
            if expl is not None:
                actions, _ = expl.remap_with_eps_tensor(actions, eps, self._env_ids, self._action_space)
            return actions.to(torch.int32), pred["qvalues"].contiguous()
        from rltime_amd._lib import lib, check
        adv, val, n = pol.actor_head_raw(state, 1)
        adv = adv.contiguous()
        A = adv.shape[1]
        E = adv.shape[0] // n
        dev = adv.device
        actions = torch.empty(E, dtype=torch.int32, device=dev)
        qvalues = torch.empty((E, A), dtype=torch.float32, device=dev)
        u = rnd = expo = None
        eps_min = 0.0
        if expl is not None:
            # same draws, same order as EpsilonGreedyExplorationManager.remap_with_eps_tensor
            u = torch.rand(E, device=dev)
            rnd = torch.randint(0, self._action_space.n, (E,), device=dev)
            expo = expl._device_exponents(self._env_ids, dev)
            eps_min = float(expl.eps_min)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)   # noqa: E731
        if val is not None:
            val = val.contiguous()
        check(lib.mirl_actor_head(E, n, A, p(adv), p(val), val.shape[1] if val is not None else 0,
                                  p(eps) if expl is not None else C.c_void_p(None), p(expo), eps_min, p(u), p(rnd),
                                  p(actions), p(qvalues), C.c_void_p(None),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mirl_actor_head")
        return actions, qvalues

    def _fast_steps(self, iters):
        """The fused vector step (acting/fast_step.py)."""
        from .fast_step import FastActingStep, IngestedSamples
        sink0 = self._sink
        # q-values are only computed when somebody keeps them (DeviceSamples for the caller, or a replay that stores
        # policy outputs / initialises priorities from them)
        need_q = sink0 is None or bool(getattr(sink0, "_keep_policy", True)) or bool(getattr(sink0, "_acting_priority_init", False))
        if self._fast is not None:
            self._fast.set_need_q(need_q)
        if self._fast is None:
            self._fast = FastActingStep(self, self._reset_obs, need_q=need_q)
            if self._tracker is None:
                from .episode_tracker import EpisodeTracker
                self._tracker = EpisodeTracker(self._num_envs, self._action_space.n, self._policy.device())
            self._fast.tracker = self._tracker
            if getattr(self, "_fast_restore", None) is not None:
                self._fast.set_state(self._fast_restore)
                self._fast_restore = None
            if getattr(self, "_tracker_restore", None) is not None:
                self._tracker.set_state(self._tracker_restore)
                self._tracker_restore = None
        fs, sink = self._fast, self._sink
        # weight-derived buffers and the pending action selection are functions of (parameters, epsilon, input state): the
        # selection is idempotent on the carry and redraws the same Philox blocks, so a call that follows another one with NO
        # learner update in between (two acting calls per optimizer step at the benchmark's train_frequency) keeps them
        eps = self._eps()
        stamp = (tuple((p.data_ptr(), p._version) for p in self._policy.parameters()), float(eps), fs.need_q)
        if getattr(fs, "selected_with", None) != stamp:
            fs.refresh()
            fs.set_eps(eps)
            fs.reselect()
            fs.selected_with = stamp
        keep_policy = False
        if sink is not None:
            if sink._h is None:
                pf = self._action_space.n if sink._keep_policy else 0
                sink.configure(fs.example, self._num_envs, self._base_env_id, policy_f32=pf)
                if getattr(sink, "_dedup", False) and fs.trusted_stack and not getattr(sink, "_acting_priority_init", False):
                    sink.prime_stack(self._reset_obs)
            keep_policy = bool(sink._policy_f32)
        out = None
        if fs.can_rollout(iters, sink) and fs.rollout(iters, sink, keep_policy=keep_policy, clip=self.clip_rewards):
            # the whole call — env steps, pre-step kernels, ingests, forwards — went out as ONE captured graph
            self._tracker.flush()
            return IngestedSamples(iters * self._num_envs, self._tracker)
        for _ in range(iters):
            clip = self.clip_rewards and sink is not None
            pre_done = False
            if fs.env_into and fs.env_pre and fs.tracker is not None:
                obs, rewards, dones, stats, pre_done = fs.env_step_pre(clip=clip), None, None, None, True
            elif fs.env_into:
                obs, rewards, dones = fs.env_step()
                stats = None
            else:
                obs, rewards, dones, stats = self._vec_env.step_device(fs.actions)
            if stats is not None:
                raise RuntimeError("the fused acting step keeps the episode statistics on the device; this env returns "
                                   "host episode stats (set actor.fast_step = False for it)")
            fields = fs.step(obs, rewards, dones, sink=sink, keep_policy=keep_policy, clip=clip, pre_done=pre_done)
            if sink is None:
                if out is None:
                    out = DeviceSamples(fs.example, self._num_envs, self._base_env_id)
                out.append(**fields)
        self._tracker.flush()
        if sink is not None:
            return IngestedSamples(iters * self._num_envs, self._tracker)
        out.episode_tracker = self._tracker
        return out

    def _device_steps(self, iters):
        if self._use_graph and getattr(self, "fast_step", True) and self._fast is not False:
            from .fast_step import FastActingStep
            if self._fast is not None or FastActingStep.supports(self):
                try:
                    return self._fast_steps(iters)
                except Exception as e:                    # e.g. graph capture refused: the generic path still works
                    if self._fast is not None and self._fast.graph is not None:
                        raise
                    import logging
                    logging.getLogger().warning("fused acting step unavailable (%s); using the generic device path", e)
                    self._fast = False
            else:
                self._fast = False
        out, pending = None, None
        for it in range(iters):
            # (1)+(2) actor.py:108-122: action selection with the current weights
            if pending is not None:
                actions, qvalues = pending
            elif self._graphed is not None:
                actions, qvalues = self._graphed.reselect(self._eps())
            else:
                actions, qvalues = self._act_eager(self.last_state)
            # (3) actor.py:124: the environments
            obs, rewards, dones, stats = self._vec_env.step_device(actions)
            rewards, dones8 = rewards.to(torch.float32), dones.to(torch.uint8)
            if stats is None:
                # episode reward / length and the action histogram on the device
                # (policy_trainer.py:75-131), one launch per vector step
                if self._tracker is None:
                    from .episode_tracker import EpisodeTracker
                    self._tracker = EpisodeTracker(self._num_envs, self._action_space.n, rewards.device)
                    if getattr(self, "_tracker_restore", None) is not None:
                        self._tracker.set_state(self._tracker_restore)
                        self._tracker_restore = None
                self._tracker.step(rewards, dones8, actions)
            # (4) actor.py:128: next input state (+ the next action when replayed from the graph)
            fields, pending = None, None
            if self._use_graph:
                rec = [layer for layer in self._policy.model.layers if layer.is_recurrent()]
                # GraphedStep re-binds layer.last_state to its static carry and advances
                # it through warm-up forwards and a capture whose kernels never run; if
                # the capture fails the eager fallback must continue from THIS state
                saved = None if self._graphed is not None else \
                    [None if layer.last_state is None else tuple(t.clone() for t in layer.last_state) for layer in rec]
                try:
                    if self._graphed is None:
                        self._graphed = GraphedStep(self, obs, dones)
                    fields, next_actions, next_q = self._graphed.after_env_step(obs, dones, self._eps())
                    pending = (next_actions, next_q)
                    example = self._graphed.example
                except Exception as e:                     # capture unsupported for this model: stay eager
                    import logging
                    logging.getLogger().warning("acting graph capture failed (%s); running eagerly", e)
                    self._use_graph, self._graphed = False, None
                    if saved is not None:
                        for layer, keep in zip(rec, saved):
                            layer.last_state = keep
            if fields is None:
                states = self._policy.make_input_state(obs, dones)
                fields = _pack_state(states)
                example = deep_apply(states, lambda x: x[0].cpu().numpy()) if out is None else None
                self.last_state = states
            if out is None:
                out = DeviceSamples(example, self._num_envs, self._base_env_id)
            fields.update(actions=actions, policy=qvalues, rewards=rewards, dones=dones8, episode_stats=stats)
            out.append(**fields)
        if self._tracker is not None:
            self._tracker.flush()
            out.episode_tracker = self._tracker
        return out

    def _eps(self):
        return self._exploration._get_eps(self._progress) if self._exploration is not None else 0.0


def _to_device_tree(state, device):
    return deep_apply(state, lambda x: x if isinstance(x, torch.Tensor)
                      else torch.as_tensor(x, device=device))


def _pack_state(states):
    """input-state pytree -> the flat per-transition arrays of mirl_ingest."""
    x = states["x"]
    fields = {}
    if isinstance(x, (tuple, list)):
        fields["frames"] = x[0].contiguous()
        fields["extra"] = torch.cat([v.reshape(v.shape[0], -1).float() for v in x[1:]], dim=1).contiguous()
    else:
        fields["frames"] = x.contiguous()
    rec, initials = [], None
    for k, sub in states.items():
        if k == "x" or not sub:
            continue
        for name, v in sub.items():
            if name == "initials":
                initials = v
            else:
                rec.append(v.reshape(v.shape[0], -1).float())
    if rec:
        fields["state"] = torch.cat(rec, dim=1).contiguous()
    if initials is not None:
        fields["initials"] = initials.float().contiguous()
    return fields
