"""On-policy ("online") history for the A2C / PPO plumbing config (reference
rltime/history/online_history.py:4-120 over history.py:71-286).

Host-side Python, like the reference: BASELINE configs[0] (`cartpole_ppo.json`) is a CPU
run that only proves config -> registry -> actor -> history -> trainer -> logger wiring;
nothing here is on the MI355X hot path and nothing here touches the HIP library.

Layout: one `_Track` per env holding the env's pending transitions as parallel lists
(structure of arrays) instead of the reference's list of per-transition dicts.  The lazily
extended n-step record of a transition (history.py:71-108) is three more parallel lists —
running return, steps covered, bootstrap mask; the n-step target state of transition i is
`next_state[i + nstep[i] - 1]`, so no object reference is stored for it.
"""
from rltime_amd.general.utils import deep_apply, deep_stack


class _Track:
    """Pending transitions of ONE env, oldest first."""

    __slots__ = ("state", "next_state", "reward", "done", "policy_output", "ret", "nstep", "mask", "newest_state")

    def __init__(self):
        self.state, self.next_state, self.reward, self.done, self.policy_output = [], [], [], [], []
        self.ret, self.nstep, self.mask = [], [], []
        self.newest_state = None          # next_state of the last transition ever added (survives removal)

    def __len__(self):
        return len(self.reward)

    def add(self, sample):
        """history.py:146-167: the transition starts from the previous one's next_state (an
        env's very first transition from its own), its n-step record starts at one step."""
        nxt = sample["next_state"]
        self.state.append(nxt if self.newest_state is None else self.newest_state)
        self.next_state.append(nxt)
        self.newest_state = nxt
        self.reward.append(sample["reward"])
        self.done.append(sample["done"])
        self.policy_output.append(sample["policy_output"])
        self.ret.append(float(sample["reward"]))
        self.nstep.append(1)
        self.mask.append(1 - sample["done"])

    def extend(self, i, want, discount):
        """history.py:71-108 for transition i: cover up to `want` steps with what is there.
        The return stops growing at an episode end; steps covered and the target state keep
        advancing (consecutive target states for recurrent bootstrapping)."""
        stop = min(i + want, len(self))
        for j in range(i + self.nstep[i], stop):
            if self.mask[i]:
                self.ret[i] += discount(self.nstep[i], self.reward[j], self.policy_output[j])
            self.nstep[i] += 1
            if self.done[j]:
                self.mask[i] = 0.

    def drop(self, count):
        for column in (self.state, self.next_state, self.reward, self.done, self.policy_output,
                       self.ret, self.nstep, self.mask):
            del column[:count]


class OnlineHistoryBuffer:
    def __init__(self, max_delayed_steps=5000, fixed_target=True, **kwargs):
        """online_history.py:21-47; **kwargs are History's (history.py:17-18: nstep_target, nstep_train, prefix_steps=0,
        discount_function=None, state_store=None)."""
        self._base_init(**kwargs)
        self.max_delayed_steps = max_delayed_steps
        self.fixed_target = fixed_target
        self.last_env = None
        self.tracks = {}

    def _base_init(self, nstep_target, nstep_train, prefix_steps=0, discount_function=None, state_store=None):
        assert nstep_target == 1 or discount_function is not None, \
            "History buffer must get a 'discount_function' for nstep_target>1"
        self.nstep_target, self.nstep_train, self.prefix_steps = nstep_target, nstep_train, prefix_steps
        self.discount_function = discount_function
        self.state_store = state_store

    # -- feeding ----------------------------------------------------------------------------
    def update(self, samples):
        """history.py:123-176, then online_history.py:61-73: an env holding more than
        max_delayed_steps loses its oldest transitions (acting outran training)."""
        for sample in samples:
            track = self.tracks.get(sample["env_id"])
            if track is None:
                track = self.tracks[sample["env_id"]] = _Track()
            track.add(sample)
        discarded = 0
        for track in self.tracks.values():
            over = len(track) - self.max_delayed_steps
            if over > 0:
                track.drop(over)
                discarded += over
        return {"discarded_steps": discarded}

    def _sequences_ready(self):
        return sum(len(t) // self.nstep_train for t in self.tracks.values())

    def needed_feed_count(self, mbatch_size, num_envs):
        """online_history.py:75-79: feed one vector step whenever no batch can be formed."""
        return None if self._sequences_ready() >= mbatch_size else num_envs

    # -- training batches --------------------------------------------------------------------
    def _window(self, track, steps):
        """history.py:178-201 on the env's oldest `steps` transitions.  fixed_target: no
        transition's target looks past the end of the window (history.py:187-190; the bound
        only ever shrinks along the window, as in the reference)."""
        want = self.nstep_target
        for i in range(steps):
            if self.fixed_target:
                want = min(want, steps - i)
            track.extend(i, want, self.discount_function)
        return [{"states": track.state[i], "target_states": track.next_state[i + track.nstep[i] - 1],
                 "returns": track.ret[i], "nsteps": track.nstep[i], "target_masks": track.mask[i],
                 "policy_outputs": track.policy_output[i]} for i in range(steps)]

    def get_train_data(self, mbatch_size, train_progress=None):
        """online_history.py:81-120: `mbatch_size` sequences of nstep_train consecutive
        transitions, taken round-robin over the env ids starting after the env served last,
        each removed as soon as it is handed out.  None = feed more."""
        assert self.prefix_steps == 0, "Online history does not support prefix/burnin steps"
        T = self.nstep_train
        if self._sequences_ready() < mbatch_size:
            return None
        ids = sorted(self.tracks)
        # (`not last_env`, online_history.py:101-103: env id 0 restarts the walk like "none served yet")
        at = 0 if not self.last_env else (ids.index(self.last_env) + 1) % len(ids)
        windows = []
        while len(windows) < mbatch_size:
            env = ids[at]
            track = self.tracks[env]
            if len(track) >= T:
                windows.append(self._window(track, T))
                track.drop(T)
                self.last_env = env
            at = (at + 1) % len(ids)
        return self._assemble(windows)

    def _assemble(self, windows):
        """history.py:203-286: time-major (T, B, ...) arrays; scalars through np.stack (so the
        dtypes are NumPy's promotion of the stored Python / NumPy scalars, as in the reference),
        state pytrees leaf by leaf."""
        B, T = len(windows), len(windows[0])
        rows = [w[t] for t in range(T) for w in windows]
        batch = {key: deep_stack([r[key] for r in rows]) for key in rows[0] if key not in ("states", "target_states")}
        for key in ("target_states", "states"):
            batch[key] = deep_stack([r[key] for r in rows])
        batch = deep_apply(batch, lambda x: x.reshape((T, B) + x.shape[1:]))
        batch["extra_data"] = {}
        return batch

    def update_losses(self, indices, losses):
        pass

    def close(self):
        pass
