"""Oracle: replay history buffers (TEST INFRASTRUCTURE ONLY).

A plain Python/numpy restatement of the reference's per-transition record
buffers and their sampling / batch assembly / priority logic:

  * rltime/history/history.py                       (History)
  * rltime/history/replay_history.py                (ReplayHistoryBuffer)
  * rltime/history/prioritized_replay_history.py    (PrioritizedReplayHistoryBuffer)
  * rltime/history/data_structures/cyclic_array.py  (CyclicArray -> _Ring)
  * rltime/general/utils.py:25-101                  (deep_stack, deep_apply, anneal_value)

It keeps the reference's *algorithm class* on purpose (one record per
transition, shared state objects between neighbours, np.stack of thousands of
small arrays) because it doubles as the CPU baseline that bench.py times next
to the HIP path.
"""
import random
from collections import deque

import numpy as np

from .sumtree import SumTree, MinTree


# ----------------------------------------------------------------------------
# pytree helpers (general/utils.py:25-68)
# ----------------------------------------------------------------------------
def tree_stack(items, axis=0):
    """general/utils.py:25-53 (deep_stack): stack a list of equal-structure
    pytrees leaf by leaf."""
    head = items[0]
    if isinstance(head, np.ndarray):
        return np.stack(items, axis=axis)
    if isinstance(head, (list, tuple)):
        return type(head)(
            tree_stack([it[k] for it in items], axis) for k in range(len(head)))
    if isinstance(head, dict):
        return {k: tree_stack([it[k] for it in items], axis) for k in head}
    if head is None:
        return None
    return np.stack(list(items), axis=axis)


def tree_map(x, fn):
    """general/utils.py:56-68 (deep_apply)."""
    if isinstance(x, (list, tuple)):
        return type(x)(tree_map(v, fn) for v in x)
    if isinstance(x, dict):
        return {k: tree_map(v, fn) for k, v in x.items()}
    if x is None:
        return None
    return fn(x)


def anneal(base, progress, mode, default_target=0.0):
    """general/utils.py:85-103 (anneal_value)."""
    assert progress >= 0
    progress = min(progress, 1.0)
    if mode is False or mode is None:
        return base
    target = default_target if mode is True else float(mode)
    return base + (target - base) * progress


class _Ring:
    """Pop-front / push-back / index container (cyclic_array.py:1-72)."""

    def __init__(self):
        self._items = []
        self._head = 0

    def push(self, item):
        self._items.append(item)

    def pop_front(self):
        item = self._items[self._head]
        self._items[self._head] = None
        self._head += 1
        if self._head > 4096 and self._head * 2 > len(self._items):
            del self._items[:self._head]
            self._head = 0
        return item

    def __len__(self):
        return len(self._items) - self._head

    def __getitem__(self, i):
        if isinstance(i, slice):
            lo, hi, _ = i.indices(len(self))
            return self._items[self._head + lo:self._head + hi]
        if i < 0:
            i += len(self)
        assert 0 <= i < len(self)
        return self._items[self._head + i]


# ----------------------------------------------------------------------------
# History base (history.py)
# ----------------------------------------------------------------------------
class OracleHistory:
    """history.py:8-335."""

    def __init__(self, nstep_target, nstep_train, prefix_steps=0,
                 discount_function=None, state_stack=None):
        # history.py:40-59
        assert nstep_target == 1 or discount_function is not None
        self.nstep_target = nstep_target
        self.nstep_train = nstep_train
        self.prefix_steps = prefix_steps
        self.discount_function = discount_function
        # stand-in for StateStore.stack (backend.py:136-153) on the CPU device
        self.state_stack = state_stack or tree_stack
        self.rings = {}        # env id -> _Ring of records   (history.py:52)
        self.last_record = {}  # env id -> newest record      (history.py:55)

    # hooks (history.py:61-69)
    def _on_added(self, rec):
        pass

    def _on_removed(self, rec):
        pass

    def _extend_nstep(self, env, pos, want):
        """history.py:71-108 (_update_nstep): lazily extend the cached n-step
        record of ring position ``pos`` up to ``want`` steps."""
        ring = self.rings[env]
        rec = ring[pos]
        stop = min(pos + want, len(ring))
        for j in range(pos + rec['nstep'], stop):
            nxt = ring[j]
            if rec['target_mask']:                       # history.py:87-90
                rec['return'] += self.discount_function(
                    rec['nstep'], nxt['reward'], nxt['policy_output'])
            rec['nstep'] += 1                            # history.py:98
            rec['target_state'] = nxt['next_state']      # history.py:102
            if nxt['done']:                              # history.py:104-108
                rec['target_mask'] = 0.

    def _drop_oldest(self, env, count):
        """history.py:110-121 (_remove_samples)."""
        ring = self.rings[env]
        assert len(ring) >= count
        for _ in range(count):
            self._on_removed(ring[0])
            ring.pop_front()

    def update(self, new_samples):
        """history.py:123-176."""
        for rec in new_samples:
            rec['return'] = float(rec['reward'])         # history.py:146
            rec['target_mask'] = 1 - rec['done']         # history.py:147
            rec['target_state'] = rec['next_state']      # history.py:154
            rec['nstep'] = 1
            env = rec['env_id']
            prev = self.last_record.get(env)
            # history.py:159-167: state is the previous record's next_state
            # (same object); the very first record of an env reuses its own.
            rec['state'] = rec['next_state'] if prev is None \
                else prev['next_state']
            if env not in self.rings:
                self.rings[env] = _Ring()
            self.rings[env].push(rec)
            self.last_record[env] = rec
            self._on_added(rec)
        return {}

    def _window(self, env, pos, steps, fixed_target=False):
        """history.py:178-201 (_make_sample_range)."""
        assert pos >= 0 and pos + steps <= len(self.rings[env])
        want = self.nstep_target
        out = []
        for i in range(pos, pos + steps):
            if fixed_target:                             # history.py:187-190
                want = min(want, pos + steps - i)
            self._extend_nstep(env, i, want)
            rec = self.rings[env][i]
            out.append({
                "target_states": rec["target_state"],
                "states": rec["state"],
                "returns": rec["return"],
                "nsteps": rec["nstep"],
                "target_masks": rec["target_mask"],
                "policy_outputs": rec["policy_output"],
            })
        return out

    def _assemble(self, windows, extra=None):
        """history.py:203-286 (_make_train_batch): time-major batch."""
        mbatch = len(windows)
        steps = len(windows[0])
        # history.py:228-230: transpose so that time is the outer index
        flat = []
        for column in zip(*windows):
            flat.extend(column)

        batch = {}
        for key in flat[0]:
            if key not in ('states', 'target_states'):   # history.py:235-241
                batch[key] = tree_stack([rec[key] for rec in flat])

        seqlen = self.nstep_train + self.prefix_steps
        if self.nstep_target < seqlen and \
                np.all(batch['nsteps'] == self.nstep_target):
            # history.py:245-265: one stack of L*B states + the last n*B
            # target states; states / target_states are two views of it.
            pile = [rec['states'] for rec in flat]
            assert len(pile) == mbatch * seqlen
            shift = mbatch * self.nstep_target
            pile += [rec['target_states'] for rec in flat[-shift:]]
            stacked = self.state_stack(pile)
            batch['states'] = tree_map(stacked, lambda x: x[:mbatch * seqlen])
            batch['target_states'] = tree_map(stacked, lambda x: x[shift:])
        else:                                            # history.py:266-270
            for key in ('target_states', 'states'):
                batch[key] = self.state_stack([rec[key] for rec in flat])

        # history.py:274-275
        batch = tree_map(
            batch, lambda x: x.reshape((steps, mbatch) + x.shape[1:]))
        if extra is not None:                            # history.py:279-284
            assert len(extra) == mbatch
            batch['extra_data'] = tree_stack(extra, axis=1)
        else:
            batch['extra_data'] = {}
        return batch


# ----------------------------------------------------------------------------
# Online (on-policy) history (online_history.py) — BASELINE configs[0] plumbing
# ----------------------------------------------------------------------------
class OracleOnline(OracleHistory):
    """online_history.py:4-120."""

    def __init__(self, max_delayed_steps=5000, fixed_target=True, **kw):
        super().__init__(**kw)
        self.max_delayed_steps = max_delayed_steps
        self.fixed_target = fixed_target
        self.last_env = None

    def _enough(self, mbatch):
        """online_history.py:49-59."""
        return sum(int(len(r) / self.nstep_train) for r in self.rings.values()) >= mbatch

    def update(self, new_samples):
        """online_history.py:61-73."""
        out = super().update(new_samples)
        lost = 0
        for env, ring in self.rings.items():
            if len(ring) > self.max_delayed_steps:
                extra = len(ring) - self.max_delayed_steps
                self._drop_oldest(env, extra)
                lost += extra
        out['discarded_steps'] = lost
        return out

    def needed_feed_count(self, mbatch_size, num_envs):
        """online_history.py:75-79."""
        return None if self._enough(mbatch_size) else num_envs

    def get_train_data(self, mbatch_size, train_progress=None):
        """online_history.py:81-120."""
        assert self.prefix_steps == 0
        if not self._enough(mbatch_size):
            return None
        ids = sorted(self.rings)
        k = 0 if not self.last_env else (ids.index(self.last_env) + 1) % len(ids)   # :101-103
        windows = []
        while len(windows) < mbatch_size:
            env = ids[k]
            if len(self.rings[env]) >= self.nstep_train:
                windows.append(self._window(env, 0, self.nstep_train, self.fixed_target))
                self._drop_oldest(env, self.nstep_train)
                self.last_env = env
            k = (k + 1) % len(ids)
        return self._assemble(windows)


def make_gae_discount(gamma, lam):
    """a2c.py:48-66: k-th term of a truncated GAE(lambda) return."""
    def discount(nstep, reward, policy_output):
        v = policy_output['values']
        return (gamma ** nstep) * (lam ** (nstep - 1)) * (v + lam * (reward - v))
    return discount


# ----------------------------------------------------------------------------
# Uniform replay (replay_history.py)
# ----------------------------------------------------------------------------
class OracleReplay(OracleHistory):
    """replay_history.py:6-184."""

    def __init__(self, size, train_frequency, avoid_episode_crossing=False,
                 **kw):
        super().__init__(**kw)
        self.size = size
        self.train_frequency = train_frequency
        self.avoid_episode_crossing = avoid_episode_crossing
        self.fifo = deque()     # global insertion order (replay_history.py:55)
        self.train_quota = 0

    def needed_feed_count(self, mbatch_size, num_envs):
        """replay_history.py:62-75."""
        if not self.train_frequency:
            return 0
        if self.train_quota > 0:
            return None
        return max(int(-self.train_quota / self.train_frequency), num_envs)

    def _on_added(self, rec):
        """replay_history.py:77-91: global-FIFO eviction, then quota."""
        if len(self.fifo) >= self.size:
            assert len(self.fifo) == self.size
            victim = self.fifo.popleft()
            env = victim['env_id']
            assert self.rings[env][0] is victim
            self._drop_oldest(env, 1)
        self.fifo.append(rec)
        if self.train_frequency:
            self.train_quota += self.train_frequency

    def _shift_window(self, env, start, steps):
        """replay_history.py:142-171 (_refine_sample_range)."""
        if not self.avoid_episode_crossing:
            return start
        ring = self.rings[env]
        for k in range(steps - 1):
            if ring[start + k]['done']:
                if k < steps / 2:
                    start = max(start - (steps - k - 1), 0)
                else:
                    start = min(start + k + 1, len(ring) - steps)
                break
        return start

    def _draw(self, mbatch_size, train_progress):
        """replay_history.py:93-140 (_get_train_data)."""
        seqlen = self.nstep_train + self.prefix_steps
        avail = {}
        total = 0
        for env, ring in self.rings.items():
            n = len(ring) - (seqlen + self.nstep_target - 1)  # :104
            if n > 0:
                avail[env] = n
                total += n
        if total < mbatch_size:
            assert len(self.fifo) < self.size
            return None
        picks = np.random.choice(total, mbatch_size)          # :118
        self.last_picks = [int(p) for p in picks]
        self.last_windows = []
        windows = []
        for pick in picks:
            win = None
            for env, n in avail.items():                      # :124-134
                if pick < n:
                    start = self._shift_window(env, pick, seqlen)
                    self.last_windows.append((env, int(start)))
                    win = self._window(env, start, seqlen)
                    break
                pick -= n
            assert win
            windows.append(win)
        return self._assemble(windows)

    def get_train_data(self, mbatch_size, train_progress=None):
        """replay_history.py:173-184."""
        if self.train_frequency:
            self.train_quota -= mbatch_size * self.nstep_train
            assert self.train_quota < 100 * mbatch_size * self.nstep_train
            assert self.train_quota > -100 * mbatch_size * self.nstep_train
        return self._draw(mbatch_size, train_progress)

    def update_losses(self, indices, losses):
        """history.py:332-335: no-op for uniform replay."""


# ----------------------------------------------------------------------------
# Prioritized sequence replay (prioritized_replay_history.py)
# ----------------------------------------------------------------------------
class OraclePrioritizedReplay(OracleReplay):
    """prioritized_replay_history.py:10-356."""

    def __init__(self, alpha=0.6, beta=0.4, beta_anneal=False, eps=1e-6,
                 overlap=None, max_weight_factor=0.9,
                 global_importance_scaling=False, **kw):
        super().__init__(**kw)
        self.alpha = alpha
        self.beta = beta
        self.beta_anneal = beta_anneal
        self.eps = eps
        self.max_weight_factor = max_weight_factor
        self.global_importance_scaling = global_importance_scaling
        if overlap is None:                                   # :97-103
            overlap = int(self.nstep_train / 2)
        elif overlap < 0:
            overlap = self.nstep_train + overlap
            assert overlap >= 0
        assert overlap < self.nstep_train
        self.overlap = overlap
        self.gap = self.nstep_train - overlap                 # :105

        self.n_slots = int(self.size / self.gap)              # :109
        cap = 1
        while cap < self.n_slots:
            cap *= 2
        self.tree = SumTree(cap)
        self.min_tree = MinTree(cap) if global_importance_scaling else None
        self.initial_loss = 1.0                               # :120
        self.free_slots = deque(range(self.n_slots))          # :123-125
        self.slot_record = [None] * self.n_slots              # :129
        self.env_first_offset = {}                            # :134

    def _on_added(self, rec):
        """prioritized_replay_history.py:136-172 (_sample_added)."""
        super()._on_added(rec)
        assert len(self.free_slots) > 0
        rec['loss'] = self.initial_loss
        env = rec['env_id']
        first = self.env_first_offset.setdefault(env, 0)
        offset = first + len(self.rings[env]) - 1
        rec['env_buffer_offset'] = offset
        base = offset - self.nstep_train + 1 - self.nstep_target + 1  # :155
        if base % self.gap == 0 and base >= first + self.prefix_steps:
            slot = self.free_slots.popleft()
            head = self.rings[env][base - first]
            head['prioritization_index'] = slot
            self.slot_record[slot] = head
            self._reprioritize(slot)

    def _reprioritize(self, slot):
        """prioritized_replay_history.py:174-208 (_recalc_weighted_priority)."""
        head = self.slot_record[slot]
        assert head['prioritization_index'] == slot
        base = head['env_buffer_offset']
        assert base % self.gap == 0
        env = head['env_id']
        pos = base - self.env_first_offset[env]
        span = self.rings[env][pos:pos + self.nstep_train]
        assert span[0] is head
        if self.nstep_train == 1:
            mixed = span[0]['loss']
        else:
            losses = [rec['loss'] for rec in span]
            mixed = self.max_weight_factor * np.max(losses) + \
                (1 - self.max_weight_factor) * np.mean(losses)
        priority = mixed ** self.alpha
        head['weighted_priority'] = priority
        self.tree.set_leaf(slot, priority)
        if self.min_tree is not None:
            self.min_tree.set_leaf(slot, priority)

    def _on_removed(self, rec):
        """prioritized_replay_history.py:210-230 (_sample_removed)."""
        env = rec['env_id']
        probe = self.rings[env][self.prefix_steps]
        if probe['env_buffer_offset'] % self.gap == 0 and \
                'prioritization_index' in probe:
            slot = probe['prioritization_index']
            probe['prioritization_index'] = None
            self.tree.set_leaf(slot, 0)
            if self.min_tree is not None:
                self.min_tree.set_leaf(slot, np.inf)
            self.free_slots.append(slot)
            self.slot_record[slot] = None
        assert rec['env_buffer_offset'] == self.env_first_offset[env]
        self.env_first_offset[env] += 1

    def _stratified(self, count):
        """prioritized_replay_history.py:232-241 (_sample_proportional)."""
        total = self.tree.total()
        seg = total / count
        picks = []
        self.last_uniforms = []
        for i in range(count):
            u = random.random()
            self.last_uniforms.append(u)
            picks.append(self.tree.descend(u * seg + i * seg))
        return picks

    def update_losses(self, indices, losses):
        """prioritized_replay_history.py:243-279."""
        touched = {}
        T = self.nstep_train
        for (env, offset), loss in zip(indices, losses):
            first = self.env_first_offset[env]
            if offset < first:                       # evicted meanwhile :254
                continue
            ring = self.rings[env]
            ring[offset - first]['loss'] = abs(loss) + self.eps      # :263
            base = offset - (offset % self.gap)
            while base + T > offset and base >= first:            # :268-274
                slot = ring[base - first].get('prioritization_index', None)
                if slot is not None:
                    touched[slot] = True
                base -= self.gap
        for slot in touched:
            self._reprioritize(slot)

    def _draw(self, mbatch_size, train_progress):
        """prioritized_replay_history.py:281-356 (_get_train_data)."""
        slots = self._stratified(mbatch_size)
        beta = anneal(self.beta, train_progress, self.beta_anneal, 1.0)
        active = len(self.slot_record) - len(self.free_slots)         # :291
        seqlen = self.prefix_steps + self.nstep_train
        if active < mbatch_size:
            assert len(self.fifo) < self.size
            return None
        self.last_slots = list(slots)
        self.last_windows = []
        windows, extra = [], []
        p_sum = self.tree.total()
        for slot in slots:
            head = self.slot_record[slot]
            assert head is not None
            env = head['env_id']
            base = head['env_buffer_offset']
            pos = base - self.env_first_offset[env] - self.prefix_steps
            assert pos >= 0
            pos = self._shift_window(env, pos, seqlen)
            self.last_windows.append((env, int(pos)))
            windows.append(self._window(env, pos, seqlen))
            weight = ((self.tree.leaf(slot) / p_sum) * active) ** (-beta)  # :327
            where = [(-1, -1)] * self.prefix_steps + \
                [(env, off) for off in range(base, base + self.nstep_train)]
            extra.append({
                'importance_weights': np.array([weight] * seqlen),
                'loss_indices': np.array(where),
            })
        batch = self._assemble(windows, extra)
        if self.global_importance_scaling:                            # :349-354
            p_min = self.min_tree.total() / p_sum
            top = (p_min * active) ** (-beta)
        else:
            top = np.max(batch['extra_data']['importance_weights'])
        batch['extra_data']['importance_weights'] /= top
        return batch


def make_discount(gamma):
    """multi_step_trainer.py:70-74 (_get_discount_function)."""
    def discount(nstep, reward, policy_output):
        return (gamma ** nstep) * reward
    return discount


# ----------------------------------------------------------------------------
# Acting-time priority initialisation — an EXTENSION, not reference behaviour.
# The reference lists it as missing (prioritized_replay_history.py:33-36): new
# samples enter at the constant maximum loss 1.0 and keep it until the learner
# first trains on them.  This restates, on the reference's record structures, what
# rltime_amd's device replay does when `acting_priority_init` is set
# (include/mirl.h, mirl_replay_config), so that the HIP path has a CPU checker.
# ----------------------------------------------------------------------------
class OracleActingPriorityReplay(OraclePrioritizedReplay):
    """After every History.update call, each new transition j of an env makes
    the TD error of transition t = j - n computable from STORED data only:

        R, mask  = the n-step return / target mask of t exactly as _update_nstep
                   accumulates them (history.py:71-108; float64, gamma**k Python floats)
        v        = max_a Q(s_j)[a]            (q-values stored with transition j: the
                                               state of j is the target state of t)
        y        = h(float32(R) + float32(gamma**n) * h^-1(v) * mask)   (torch_trainer.py:124-147, float32)
        delta    = Q(s_t)[a_t] - y

    and writes it through update_losses (same abs + eps, same fan-out to the
    overlapped sequences, prioritized_replay_history.py:243-279)."""

    def __init__(self, gamma, acting_priority_vf_eps=None, **kw):
        super().__init__(**kw)
        self.gamma = gamma
        self.vf_eps = acting_priority_vf_eps

    def _h(self, x):
        e = np.float32(self.vf_eps)
        return np.float32(np.sign(x) * (np.sqrt(np.abs(x) + np.float32(1)) - np.float32(1)) + e * x)

    def _h_inv(self, y):
        eps = float(self.vf_eps)
        a = abs(float(y))
        x = a / eps - (1.0 / (2.0 * (eps * eps))) * np.sqrt(4.0 * eps * a + (2.0 * eps + 1.0) * (2.0 * eps + 1.0)) + \
            (2.0 * eps + 1.0) / (2.0 * (eps * eps))
        return np.float32(x * float(np.sign(y)))

    def update(self, new_samples):
        new_samples = list(new_samples)
        out = super().update(new_samples)
        n = self.nstep_target
        idx, losses = [], []
        for rec in new_samples:
            env = rec['env_id']
            first = self.env_first_offset[env]
            ring = self.rings[env]
            j = rec['env_buffer_offset']
            t = j - n
            if t < first:
                continue
            old = ring[t - first]
            ret = float(old['reward'])
            mask = 0 if old['done'] else 1
            for k in range(1, n):
                nxt = ring[t + k - first]
                if mask:
                    ret = ret + (self.gamma ** k) * float(nxt['reward'])
                if nxt['done']:
                    mask = 0
            v = np.float32(np.max(np.asarray(rec['policy_output']['qvalues'], dtype=np.float32)))
            if self.vf_eps:
                v = self._h_inv(v)
            y = np.float32(ret) + (np.float32(self.gamma ** n) * v) * np.float32(mask)
            if self.vf_eps:
                y = self._h(y)
            chosen = np.float32(np.asarray(old['policy_output']['qvalues'], dtype=np.float32)[int(old['policy_output']['actions'])])
            idx.append((env, t))
            losses.append(np.float32(chosen - y))
        if idx:
            self.update_losses(np.array(idx, dtype=np.int64), np.array(losses, dtype=np.float32))
        return out
