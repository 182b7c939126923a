#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the development container.

Imports the *unmodified* reference from /root/reference (through the import
shims in oracle/ref_shims, because `gym`/`cv2` are not installed), drives it
and the oracle restatement with identical seeded inputs, asserts they agree
bit-for-bit, and writes the reference's outputs as small fixtures into this
directory.  The fixtures are data only (inputs are re-derived from
tests/golden/streams.py seeds); nothing of the reference travels.

    python tests/golden/generate.py
"""
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle", "ref_shims"),
                "/root/reference"]

from tests.golden.streams import (  # noqa: E402
    StreamSpec, vector_steps, as_reference_samples, scalar_kind)
from oracle import replay as orc  # noqa: E402
from oracle import qmath  # noqa: E402
from oracle.sumtree import SumTree  # noqa: E402

from rltime.history.replay_history import ReplayHistoryBuffer  # noqa: E402
from rltime.history.prioritized_replay_history import (  # noqa: E402
    PrioritizedReplayHistoryBuffer)
from rltime.history.data_structures.segment_tree import (  # noqa: E402
    SumSegmentTree)
from rltime.general.backend import StateStore  # noqa: E402
from rltime.training.torch.dqn import DQN  # noqa: E402
from rltime.training.torch.iqn import IQN  # noqa: E402
from rltime.general.value_log import ValueLog  # noqa: E402


# --------------------------------------------------------------------------
def deep_equal(a, b, path=""):
    if isinstance(a, dict):
        assert isinstance(b, dict) and a.keys() == b.keys(), path
        for k in a:
            deep_equal(a[k], b[k], path + "/" + str(k))
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            deep_equal(x, y, path + "/%d" % i)
    else:
        a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        b = b.numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
        assert a.dtype == b.dtype, (path, a.dtype, b.dtype)
        assert a.shape == b.shape, (path, a.shape, b.shape)
        assert np.array_equal(a, b), path


def flatten(prefix, tree, out):
    if isinstance(tree, dict):
        for k, v in tree.items():
            flatten(prefix + "." + k if prefix else k, v, out)
    elif isinstance(tree, (tuple, list)):
        for i, v in enumerate(tree):
            flatten("%s.%d" % (prefix, i), v, out)
    else:
        out[prefix] = tree.numpy() if isinstance(tree, torch.Tensor) \
            else np.asarray(tree)


def tree_dump(tree_nodes, capacity):
    leaves = tree_nodes[capacity:2 * capacity]
    return (np.array([float(v) for v in leaves], dtype=np.float64),
            np.array([scalar_kind(v) for v in leaves], dtype=np.uint8))


# --------------------------------------------------------------------------
# Replay scenarios
# --------------------------------------------------------------------------
SCENARIOS = {
    "uniform_t1": dict(
        mode="uniform",
        spec=dict(seed=11, num_envs=3, frame_shape=(2, 3, 3), lstm_units=0,
                  n_actions=4, done_prob=0.1),
        hist=dict(size=40, train_frequency=4, nstep_target=1, nstep_train=1,
                  prefix_steps=0),
        gamma=0.99,
        script=[("feed", 8), ("draw", 8, 101, None), ("feed", 9),
                ("draw", 8, 102, None), ("feed", 4), ("draw", 8, 103, None)]),
    "uniform_seq": dict(
        mode="uniform",
        spec=dict(seed=12, num_envs=4, frame_shape=(2, 3, 3), lstm_units=3,
                  n_actions=5, done_prob=0.15),
        hist=dict(size=64, train_frequency=4, nstep_target=3, nstep_train=4,
                  prefix_steps=2),
        gamma=0.9,
        script=[("feed", 3), ("draw", 6, 200, None), ("feed", 9),
                ("draw", 6, 201, None), ("feed", 10), ("draw", 6, 202, None),
                ("draw", 6, 203, None)]),
    "uniform_seq_noxing": dict(
        mode="uniform",
        spec=dict(seed=13, num_envs=2, frame_shape=(1, 2, 2), lstm_units=0,
                  n_actions=3, done_prob=0.2),
        hist=dict(size=48, train_frequency=4, nstep_target=2, nstep_train=5,
                  prefix_steps=0, avoid_episode_crossing=True),
        gamma=0.95,
        script=[("feed", 20), ("draw", 10, 300, None), ("feed", 7),
                ("draw", 10, 301, None)]),
    "per_t1": dict(
        mode="per",
        spec=dict(seed=14, num_envs=4, frame_shape=(2, 3, 3), lstm_units=0,
                  n_actions=4, done_prob=0.1),
        hist=dict(size=32, train_frequency=4, nstep_target=3, nstep_train=1,
                  prefix_steps=0, alpha=0.6, beta=0.4, beta_anneal=True),
        gamma=0.99,
        script=[("feed", 2), ("draw", 8, 400, 0.0), ("feed", 4),
                ("draw", 8, 401, 0.1), ("losses", 501),
                ("draw", 8, 402, 0.2), ("losses", 502), ("feed", 5),
                ("draw", 8, 403, 0.5), ("losses", 503), ("feed", 3),
                ("draw", 8, 404, 0.9), ("losses", 504),
                ("draw", 8, 405, 1.0)]),
    "per_seq": dict(
        mode="per",
        spec=dict(seed=15, num_envs=4, frame_shape=(2, 3, 3), lstm_units=3,
                  n_actions=6, done_prob=0.12),
        hist=dict(size=64, train_frequency=4, nstep_target=2, nstep_train=4,
                  prefix_steps=2, alpha=0.9, beta=0.6, max_weight_factor=0.9),
        gamma=0.99,
        script=[("feed", 6), ("draw", 5, 600, 0.0), ("feed", 6),
                ("draw", 5, 601, 0.05), ("losses", 701),
                ("draw", 5, 602, 0.1), ("losses", 702), ("feed", 7),
                ("draw", 5, 603, 0.3), ("losses", 703), ("feed", 9),
                ("draw", 5, 604, 0.6), ("losses", 704), ("feed", 5),
                ("losses", 705), ("draw", 5, 605, 0.95)]),
    "per_seq_tuple_obs": dict(
        mode="per",
        spec=dict(seed=17, num_envs=3, frame_shape=(1, 4, 4), lstm_units=4,
                  n_actions=4, done_prob=0.1, extra_features=6),
        hist=dict(size=45, train_frequency=4, nstep_target=2, nstep_train=4,
                  prefix_steps=1, alpha=0.8, beta=0.5),
        gamma=0.99,
        script=[("feed", 11), ("draw", 4, 900, 0.1), ("losses", 901),
                ("feed", 9), ("draw", 4, 902, 0.5), ("losses", 903),
                ("draw", 4, 904, 0.7)]),
    "per_seq_noxing": dict(
        mode="per",
        spec=dict(seed=18, num_envs=3, frame_shape=(1, 4, 4), lstm_units=0,
                  n_actions=3, done_prob=0.18, extra_features=8),
        hist=dict(size=90, train_frequency=4, nstep_target=2, nstep_train=6,
                  prefix_steps=0, alpha=0.6, beta=0.4, overlap=3,
                  avoid_episode_crossing=True),
        gamma=0.9,
        script=[("feed", 20), ("draw", 6, 950, 0.2), ("losses", 951),
                ("feed", 15), ("draw", 6, 952, 0.6), ("losses", 953),
                ("feed", 4), ("draw", 6, 954, 0.9)]),
    "per_seq_full_overlap": dict(
        mode="per",
        spec=dict(seed=19, num_envs=2, frame_shape=(1, 3, 3), lstm_units=2,
                  n_actions=3, done_prob=0.1, env_base=32),
        hist=dict(size=40, train_frequency=4, nstep_target=1, nstep_train=3,
                  prefix_steps=1, alpha=0.5, beta=0.7, overlap=-1,
                  max_weight_factor=0.5),
        gamma=0.9,
        script=[("feed", 9), ("draw", 5, 970, 0.2), ("losses", 971),
                ("feed", 14), ("draw", 5, 972, 0.6), ("losses", 973),
                ("draw", 5, 974, 0.9)]),
    "per_seq_global": dict(
        mode="per",
        spec=dict(seed=16, num_envs=3, frame_shape=(1, 2, 2), lstm_units=2,
                  n_actions=3, done_prob=0.1),
        hist=dict(size=60, train_frequency=4, nstep_target=2, nstep_train=6,
                  prefix_steps=3, alpha=0.7, beta=0.5, overlap=4,
                  max_weight_factor=0.8, global_importance_scaling=True,
                  beta_anneal=0.9),
        gamma=0.97,
        script=[("feed", 17), ("draw", 4, 800, 0.3), ("losses", 801),
                ("feed", 10), ("draw", 4, 802, 0.6), ("losses", 803),
                ("feed", 6), ("draw", 4, 804, 0.8)]),
}


def run_replay_scenario(name, cfg):
    spec = StreamSpec(**cfg["spec"])
    gamma = cfg["gamma"]
    per = cfg["mode"] == "per"

    ref_cls = PrioritizedReplayHistoryBuffer if per else ReplayHistoryBuffer
    orc_cls = orc.OraclePrioritizedReplay if per else orc.OracleReplay

    def ref_discount(nstep, reward, policy_output):   # multi_step_trainer.py:72-73
        return (gamma ** nstep) * reward

    ref = ref_cls(**cfg["hist"], discount_function=ref_discount,
                  state_store=StateStore("cpu"))
    mine = orc_cls(**cfg["hist"], discount_function=orc.make_discount(gamma))

    out = {}
    step_no = 0
    rnd = 0
    last_ref_batch = None
    for op in cfg["script"]:
        if op[0] == "feed":
            for step in vector_steps(spec, op[1], start_step=step_no):
                ref.update(as_reference_samples(spec, step))
                mine.update(as_reference_samples(spec, step))
            step_no += op[1]
            assert ref.train_quota == mine.train_quota
            continue
        tag = "r%d" % rnd
        rnd += 1
        if op[0] == "draw":
            _, B, seed, progress = op
            random.seed(seed); np.random.seed(seed)
            feed_ref = ref.needed_feed_count(B, spec.num_envs)
            got_ref = ref.get_train_data(B, train_progress=progress)
            random.seed(seed); np.random.seed(seed)
            feed_mine = mine.needed_feed_count(B, spec.num_envs)
            got_mine = mine.get_train_data(B, train_progress=progress)
            assert feed_ref == feed_mine
            out[tag + ".op"] = np.array("draw")
            out[tag + ".seed"] = np.array(seed)
            out[tag + ".mbatch"] = np.array(B)
            out[tag + ".progress"] = np.array(
                -1.0 if progress is None else progress)
            out[tag + ".steps_fed"] = np.array(step_no)
            out[tag + ".feed_count"] = np.array(
                -1 if feed_ref is None else feed_ref)
            out[tag + ".quota_after"] = np.array(ref.train_quota)
            out[tag + ".is_none"] = np.array(got_ref is None)
            assert (got_ref is None) == (got_mine is None)
            if got_ref is not None:
                deep_equal(got_ref, got_mine, tag)
                flat = {}
                flatten("", got_ref, flat)
                for k, v in flat.items():
                    out[tag + ".batch." + k] = v
                out[tag + ".windows"] = np.array(
                    mine.last_windows, dtype=np.int64)
                if per:
                    out[tag + ".uniforms"] = np.array(mine.last_uniforms)
                    out[tag + ".slots"] = np.array(
                        mine.last_slots, dtype=np.int64)
                else:
                    out[tag + ".picks"] = np.array(
                        mine.last_picks, dtype=np.int64)
                last_ref_batch = got_ref
        elif op[0] == "losses":
            seed = op[1]
            P = cfg["hist"]["prefix_steps"]
            idx = last_ref_batch["extra_data"]["loss_indices"][P:]
            idx = idx.reshape(-1, 2)
            rng = np.random.RandomState(seed)
            losses = (rng.randn(idx.shape[0]) * 0.7).astype(np.float32)
            ref.update_losses(idx, losses)
            mine.update_losses(idx, losses)
            out[tag + ".op"] = np.array("losses")
            out[tag + ".indices"] = idx
            out[tag + ".losses"] = losses
            out[tag + ".steps_fed"] = np.array(step_no)
        if per:
            # full priority state after every op
            cap = ref._it_sum._capacity
            rv, rk = tree_dump(ref._it_sum._value, cap)
            mv, mk = tree_dump(mine.tree.nodes, cap)
            assert np.array_equal(rv, mv) and np.array_equal(rk, mk), tag
            all_ref = [float(v) for v in ref._it_sum._value[1:]]
            all_mine = [float(v) for v in mine.tree.nodes[1:]]
            assert all_ref == all_mine
            assert list(ref._free_indexes) == list(mine.free_slots)
            out[tag + ".leaf_val"] = rv
            out[tag + ".leaf_kind"] = rk
            out[tag + ".node_val"] = np.array(
                [float(v) for v in ref._it_sum._value], dtype=np.float64)
            out[tag + ".node_kind"] = np.array(
                [scalar_kind(v) for v in ref._it_sum._value], dtype=np.uint8)
            out[tag + ".free_slots"] = np.array(
                list(ref._free_indexes), dtype=np.int64)
            slot_env = np.full(len(ref._index_data), -1, dtype=np.int64)
            slot_base = np.full(len(ref._index_data), -1, dtype=np.int64)
            for s, rec in enumerate(ref._index_data):
                if rec is not None:
                    slot_env[s] = rec["env_id"]
                    slot_base[s] = rec["env_buffer_offset"]
            out[tag + ".slot_env"] = slot_env
            out[tag + ".slot_base"] = slot_base
            envs = sorted(ref._env_sample_offsets)
            out[tag + ".env_first"] = np.array(
                [ref._env_sample_offsets[e] for e in envs], dtype=np.int64)
            if ref._it_min is not None:
                out[tag + ".min_val"] = np.array(
                    [float(v) for v in ref._it_min._value[cap:2 * cap]])
    out["rounds"] = np.array(rnd)
    out["config"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, "replay_%s.npz" % name), **out)
    print("replay scenario %-20s rounds=%d  keys=%d" % (name, rnd, len(out)))


# --------------------------------------------------------------------------
# Bare tree: leaves + uniforms -> indices in the three precision regimes
# --------------------------------------------------------------------------
def run_tree_cases():
    out = {}
    rng = np.random.RandomState(77)
    cases = {}
    # (a) nstep_train==1 regime: Python 1.0 leaves and float32 leaves
    cap = 64
    vals = []
    for i in range(cap):
        r = rng.rand()
        if i >= 50:
            vals.append(0.0)                 # never-used leaf (neutral)
        elif r < 0.3:
            vals.append(1.0)                 # fresh sample: 1.0 ** alpha
        elif r < 0.4:
            vals.append(0)                   # evicted leaf (python int)
        else:
            vals.append(np.float32(abs(rng.randn()) + 1e-6) ** 0.6)
    cases["f32_regime"] = (cap, vals)
    # (b) nothing updated yet: all weak Python floats
    cases["weak_regime"] = (32, [1.0] * 20 + [0.0] * 12)
    # (c) nstep_train>1 regime: float32 / float64 leaves mixed
    cap = 128
    vals = []
    for i in range(cap):
        r = rng.rand()
        if i >= 100:
            vals.append(0.0)
        elif r < 0.1:
            vals.append(0)
        elif r < 0.5:
            vals.append(np.float64(abs(rng.randn()) + 0.1) ** 0.9)
        else:
            vals.append(np.float32(abs(rng.randn()) + 0.1) ** 0.9)
    cases["mixed_regime"] = (cap, vals)
    # (d) a large f32 tree (2^14) to exercise deep descents
    cap = 1 << 14
    vals = [np.float32(abs(v) + 1e-6) ** 0.6 for v in rng.randn(cap)]
    cases["f32_large"] = (cap, vals)

    for name, (cap, vals) in cases.items():
        ref = SumSegmentTree(cap)
        mine = SumTree(cap)
        for i, v in enumerate(vals):
            ref[i] = v
            mine.set_leaf(i, v)
        for B in (8, 32):
            random.seed(1234 + B)
            us = [random.random() for _ in range(B)]
            total = ref.sum()
            seg = total / B
            idx_ref, idx_mine = [], []
            for i, u in enumerate(us):
                idx_ref.append(ref.find_prefixsum_idx(u * seg + i * seg))
                idx_mine.append(mine.descend(u * seg + i * seg))
            assert idx_ref == idx_mine
            out["%s.B%d.uniforms" % (name, B)] = np.array(us)
            out["%s.B%d.index" % (name, B)] = np.array(idx_ref, dtype=np.int64)
        out[name + ".capacity"] = np.array(cap)
        out[name + ".leaf_val"] = np.array([float(v) for v in vals])
        out[name + ".leaf_kind"] = np.array(
            [scalar_kind(v) for v in vals], dtype=np.uint8)
        out[name + ".node_val"] = np.array([float(v) for v in ref._value])
        out[name + ".node_kind"] = np.array(
            [scalar_kind(v) for v in ref._value], dtype=np.uint8)
    out["cases"] = np.array(json.dumps(sorted(cases)))
    np.savez_compressed(os.path.join(HERE, "tree_cases.npz"), **out)
    print("tree cases: %s" % ", ".join(sorted(cases)))


# --------------------------------------------------------------------------
# Target / loss arithmetic through the reference trainer classes with stub
# policies
# --------------------------------------------------------------------------
class _StubPolicy:
    """Returns canned tensors from predict(); records nothing else."""

    def __init__(self, outputs):
        self.outputs = list(outputs)
        self.calls = 0

    def predict(self, x, timesteps):
        r = self.outputs[self.calls % len(self.outputs)]
        self.calls += 1
        return r

    def make_tensor(self, x, non_blocking=False):
        from rltime.models.torch.utils import make_tensor
        return make_tensor(x, "cpu")

    def zero_grad(self):
        pass


class _NullHistory:
    def __init__(self):
        self.got = None

    def update_losses(self, indices, losses):
        self.got = (np.array(indices), np.array(losses))


def make_trainer(cls, gamma, vf_eps, double_q, kappa=1.0, batch_mode="mean",
                 time_mode=None, loss_mode="huber"):
    t = cls.__new__(cls)
    t.gamma = gamma
    t.vf_scale_epsilon = vf_eps
    t.double_q = double_q
    t.loss_mode = loss_mode
    t.huber_kappa = kappa
    t.loss_aggregation = t._get_aggregator(batch_mode)
    t.loss_timestep_aggregation = \
        t._get_aggregator(time_mode) if time_mode else None
    t.value_log = ValueLog()
    t.history_buffer = _NullHistory()
    return t


def run_qmath_cases():
    out = {}
    g = torch.Generator().manual_seed(5)
    # --- value rescaling grid (torch_trainer.py:46-78)
    grid = torch.cat([torch.linspace(-300, 300, 241),
                      torch.tensor([-1e-3, 0.0, 1e-3, 1e4, -1e4])])
    for eps in (1e-3, 1e-2):
        t = make_trainer(DQN, 0.99, eps, False)
        out["vf.eps%g.x" % eps] = grid.numpy()
        out["vf.eps%g.scale" % eps] = t._vf_scale(grid).numpy()
        out["vf.eps%g.unscale" % eps] = t._vf_unscale(grid).numpy()
        assert torch.equal(t._vf_scale(grid), qmath.vf_scale(grid, eps))
        assert torch.equal(t._vf_unscale(grid), qmath.vf_unscale(grid, eps))

    T, B, A, N = 3, 4, 5, 8
    M = T * B
    returns = (torch.randn(M, generator=g).double() * 2).numpy()     # f64 like history
    nsteps = torch.randint(1, 4, (M,), generator=g).numpy()           # i64
    masks = (torch.rand(M, generator=g) > 0.25).long().numpy()        # i64
    out["tg.returns"] = returns
    out["tg.nsteps"] = nsteps
    out["tg.masks"] = masks
    q_t = torch.randn(M, A, generator=g) * 3
    q_o = torch.randn(M, A, generator=g) * 3
    z_t = torch.randn(M, N, A, generator=g) * 3
    z_o = torch.randn(M, N, A, generator=g) * 3
    out["tg.q_target"] = q_t.numpy(); out["tg.q_online"] = q_o.numpy()
    out["tg.z_target"] = z_t.numpy(); out["tg.z_online"] = z_o.numpy()
    f32 = lambda a: torch.from_numpy(np.asarray(a).astype("float32"))  # noqa: E731
    for vf_eps in (None, 1e-3):
        for dq in (False, True):
            tag = "tg.vf%s.dq%d" % ("none" if vf_eps is None else "1e-3", dq)
            # DQN
            t = make_trainer(DQN, 0.97, vf_eps, dq)
            t.target_policy = _StubPolicy([q_t])
            t.policy = _StubPolicy([q_o])
            y = t.calc_target_values(returns, {}, masks, nsteps, 1)
            mine = qmath.nstep_target(
                qmath.dqn_bootstrap(q_t, q_o if dq else q_t),
                f32(returns), f32(masks), f32(nsteps), 0.97, vf_eps)
            assert torch.equal(y, mine), tag
            out[tag + ".dqn"] = y.numpy()
            # IQN: target net forward, then selection net forward (iqn.py:18,32)
            z_sel = torch.randn(M, N, A, generator=g) * 3
            t = make_trainer(IQN, 0.97, vf_eps, dq)
            if dq:
                t.target_policy = _StubPolicy([(z_t, None)])
                t.policy = _StubPolicy([(z_sel, None)])
            else:
                t.target_policy = _StubPolicy([(z_t, None), (z_sel, None)])
                t.policy = _StubPolicy([(z_o, None)])
            y = t.calc_target_values(returns, {}, masks, nsteps, 1)
            mine = qmath.nstep_target(
                qmath.iqn_bootstrap(z_t, z_sel),
                f32(returns), f32(masks), f32(nsteps), 0.97, vf_eps)
            assert torch.equal(y, mine), tag
            out[tag + ".z_select"] = z_sel.numpy()
            out[tag + ".iqn"] = y.numpy()

    # --- losses (dqn.py:132-172, iqn.py:54-129)
    actions = torch.randint(0, A, (M,), generator=g).numpy()             # i64
    weights = torch.rand(M, generator=g).double().numpy()                 # f64
    loss_idx = np.stack([np.arange(M) % B, np.arange(M) + 100], 1)
    y_dqn = torch.randn(M, generator=g) * 2
    y_iqn = torch.randn(M, N, generator=g) * 2
    taus = torch.rand(M * N, generator=g)
    out["ls.actions"] = actions; out["ls.weights"] = weights
    out["ls.q"] = q_o.numpy(); out["ls.z"] = z_o.numpy()
    out["ls.y_dqn"] = y_dqn.numpy(); out["ls.y_iqn"] = y_iqn.numpy()
    out["ls.taus"] = taus.numpy()
    out["ls.timesteps"] = np.array(T)
    combos = [("mean", None), ("sum", None), ("mean", "mean"),
              ("sum", "mean"), ("mean", "sum")]
    for bm, tm in combos:
        for use_w in (False, True):
            for kappa in (1.0, 0.5):
                tag = "ls.%s.%s.w%d.k%g" % (bm, tm, use_w, kappa)
                extra = {"loss_indices": loss_idx}
                if use_w:
                    extra["importance_weights"] = weights
                w_t = f32(weights) if use_w else None
                for mode in ("huber", "mse"):
                    q = q_o.clone().requires_grad_(True)
                    t = make_trainer(DQN, 0.97, None, False, kappa, bm, tm,
                                     loss_mode=mode)
                    t.policy = _StubPolicy([q])
                    t._compute_grads({}, y_dqn, {"actions": actions}, extra, T)
                    q2 = q_o.clone().requires_grad_(True)
                    l2, rep2 = qmath.dqn_loss(
                        q2, torch.from_numpy(actions), y_dqn, w_t, kappa, mode,
                        T, bm, tm)
                    l2.backward()
                    assert torch.equal(q.grad, q2.grad), tag
                    assert np.array_equal(t.history_buffer.got[1], rep2.numpy())
                    out[tag + ".dqn_%s.loss" % mode] = l2.detach().numpy()
                    out[tag + ".dqn_%s.grad" % mode] = q.grad.numpy()
                    out[tag + ".dqn_%s.report" % mode] = t.history_buffer.got[1]
                z = z_o.clone().requires_grad_(True)
                t = make_trainer(IQN, 0.97, None, False, kappa, bm, tm)
                t.policy = _StubPolicy([(z, taus)])
                t._compute_grads({}, y_iqn, {"actions": actions}, extra, T)
                z2 = z_o.clone().requires_grad_(True)
                l2, rep2 = qmath.iqn_loss(
                    z2, taus, torch.from_numpy(actions), y_iqn, w_t, kappa,
                    T, bm, tm)
                l2.backward()
                assert torch.equal(z.grad, z2.grad), tag
                assert np.array_equal(t.history_buffer.got[1], rep2.numpy())
                out[tag + ".iqn.loss"] = l2.detach().numpy()
                out[tag + ".iqn.grad"] = z.grad.numpy()
                out[tag + ".iqn.report"] = t.history_buffer.got[1]
    np.savez_compressed(os.path.join(HERE, "qmath_cases.npz"), **out)
    print("qmath cases: %d arrays" % len(out))


# --------------------------------------------------------------------------
# Network boundary: reference policies (tiny shapes) -> outputs for given weights,
# inputs and torch seed.  Pins the restated PyTorch models/policies
# (rltime_amd.models / rltime_amd.policies), including checkpoint compatibility
# (same parameter names) and the burn-in state substitution.
# --------------------------------------------------------------------------
def run_model_cases():
    import io
    import gym
    from rltime.policies.torch.iqn import IQNPolicy
    from rltime.policies.torch.dqn import DQNPolicy
    from rltime.training.torch.iqn import IQN as RefIQN
    out = {}
    cnn = {"type": "cnn", "args": {"layers": [{"filters": 4, "kernel": 4, "stride": 2},
                                              {"filters": 6, "kernel": 3, "stride": 1}]}}
    lstm = {"type": "lstm", "args": {"num_units": 8}}
    fc = {"type": "fc", "args": {"fc_size": 16}}
    cases = {
        "iqn_lstm": (IQNPolicy, [cnn, lstm, fc], dict(dueling=True, embedding_dim=8, num_sampling_quantiles=4), True),
        "dqn_ff": (DQNPolicy, [cnn, fc], dict(dueling=True), False),
        "iqn_ff": (IQNPolicy, [cnn, fc], dict(dueling=False, embedding_dim=8, num_sampling_quantiles=4), False),
    }
    T, B, A = 5, 3, 4
    obs_space = gym.spaces.Box(0, 255, (2, 12, 12), dtype=np.uint8)
    for name, (cls, layers, pargs, recurrent) in cases.items():
        torch.manual_seed(42)
        pol = cls.create(model_config={"type": "sequential", "args": {"layer_configs": layers}},
                         observation_space=obs_space, action_space=gym.spaces.Discrete(A), cuda=False, **pargs)
        f = io.BytesIO()
        torch.save(pol.state_dict(), f)
        out[name + ".state_dict"] = np.frombuffer(f.getvalue(), dtype=np.uint8)
        g = torch.Generator().manual_seed(9)
        x = torch.randint(0, 256, (T * B, 2, 12, 12), generator=g, dtype=torch.uint8).numpy()
        state = {"x": x, "layer0_state": {}, "layer%d_state" % (len(layers) - 1): {}}
        if recurrent:
            initials = (torch.rand(T * B, generator=g) < 0.3).float().numpy()
            state["layer1_state"] = {"hx": torch.randn(T * B, 8, generator=g).numpy(),
                                     "cx": torch.randn(T * B, 8, generator=g).numpy(),
                                     "initials": initials}
            out[name + ".hx"], out[name + ".cx"] = state["layer1_state"]["hx"], state["layer1_state"]["cx"]
            out[name + ".initials"] = initials
        else:
            state["layer1_state"] = {}
        out[name + ".x"] = x
        torch.manual_seed(77)
        pred = pol.predict(state, T if recurrent else 1)
        if isinstance(pred, tuple):
            out[name + ".pred"], out[name + ".taus"] = pred[0].detach().numpy(), pred[1].numpy()
        else:
            out[name + ".pred"] = pred.detach().numpy()
        torch.manual_seed(78)
        act = pol.actor_predict(state, T if recurrent else 1)
        out[name + ".act_actions"], out[name + ".act_qvalues"] = act["actions"], act["qvalues"]
        if recurrent:
            out[name + ".last_hx"] = pol.model.layers[1].last_state[0].numpy()
            # burn-in substitution (multi_step_trainer.py:90-131) on (T,B) shaped data
            tr = RefIQN.__new__(RefIQN)
            tr.policy = tr.target_policy = pol
            tr.value_log = ValueLog()
            shaped = {"states": {
                "x": torch.from_numpy(x).view(T, B, 2, 12, 12),
                "layer0_state": {}, "layer2_state": {},
                "layer1_state": {k: torch.from_numpy(v.copy()).view((T, B) + v.shape[1:])
                                 for k, v in state["layer1_state"].items()}},
                "returns": torch.arange(T * B, dtype=torch.float32).view(T, B)}
            torch.manual_seed(79)
            res = tr._burn_in(shaped, 2, do_target_states=False)
            out[name + ".burn.hx"] = res["states"]["layer1_state"]["hx"].numpy()
            out[name + ".burn.cx"] = res["states"]["layer1_state"]["cx"].numpy()
            out[name + ".burn.returns"] = res["returns"].numpy()
    np.savez_compressed(os.path.join(HERE, "model_cases.npz"), **out)
    print("model cases: %d arrays" % len(out))


# --------------------------------------------------------------------------
# Pin of oracle/network64.py (the float64 anchor of tests/test_network_ab_gpu.py): the reference's own
# SequentialModel + IQNPolicy (float32, CPU) and Net64 on the same weights, frames, stored state, initials
# and quantile fractions — outputs and the gradients of a fixed linear functional of them.
# --------------------------------------------------------------------------
def run_network64_pin():
    import io
    import gym
    from rltime.policies.torch.iqn import IQNPolicy
    from oracle.network64 import Net64
    layers = [{"type": "cnn", "args": {"layers": [{"filters": 8, "kernel": 4, "stride": 2}, {"filters": 8, "kernel": 3, "stride": 1}]}},
              {"type": "lstm", "args": {"num_units": 16}}, {"type": "fc", "args": {"fc_size": 32}}]
    T, B, A, N = 6, 4, 5, 4
    torch.manual_seed(142)
    pol = IQNPolicy.create(model_config={"type": "sequential", "args": {"layer_configs": layers}},
                           observation_space=gym.spaces.Box(0, 255, (3, 16, 16), dtype=np.uint8),
                           action_space=gym.spaces.Discrete(A), cuda=False, dueling=True, embedding_dim=8, num_sampling_quantiles=N)
    f = io.BytesIO()
    torch.save(pol.state_dict(), f)
    g = torch.Generator().manual_seed(19)
    x = torch.randint(0, 256, (T, B, 3, 16, 16), generator=g, dtype=torch.uint8)
    hx, cx = torch.randn(T, B, 16, generator=g) * 0.5, torch.randn(T, B, 16, generator=g) * 0.5
    initials = (torch.rand(T, B, generator=g) < 0.3).float()
    weight = torch.randn(T * B, N, A, generator=g)
    state = {"x": x.reshape(T * B, 3, 16, 16).numpy(), "layer0_state": {}, "layer2_state": {},
             "layer1_state": {"hx": hx.reshape(T * B, 16).numpy(), "cx": cx.reshape(T * B, 16).numpy(),
                              "initials": initials.reshape(T * B).numpy()}}
    torch.manual_seed(177)
    z, taus = pol.predict(state, T)                                  # the reference's float32 forward (iqn.py:15-52 consumes this)
    pol.zero_grad()
    (z * weight).sum().backward()
    ref_grads = {k: v.grad.detach().clone() for k, v in pol.named_parameters()}
    net = Net64(pol.state_dict(), [2, 1], N, "cpu")
    z64 = net.predict(x, hx[0], cx[0], initials, taus, T)
    names = list(net.params())
    g64 = dict(zip(names, torch.autograd.grad((z64 * weight.double()).sum(), [net.p[k] for k in names])))
    worst = float((z64.float() - z).abs().max() / z.abs().max())
    assert worst <= 1e-5, worst
    for k, gr in ref_grads.items():
        rel = float((g64[k].float() - gr).abs().max() / (gr.abs().max() + 1e-12))
        assert rel <= 1e-4, (k, rel)
        worst = max(worst, rel)
    out = {"state_dict": np.frombuffer(f.getvalue(), dtype=np.uint8), "x": x.numpy(), "hx": hx.numpy(), "cx": cx.numpy(),
           "initials": initials.numpy(), "taus": taus.numpy(), "weight": weight.numpy(), "z": z.detach().numpy()}
    for k, gr in ref_grads.items():
        out["grad." + k] = gr.numpy()
    np.savez_compressed(os.path.join(HERE, "network64_pin.npz"), **out)
    print("network64 pin: reference float32 vs Net64 float64, worst relative deviation %.2e" % worst)


# --------------------------------------------------------------------------
# End-to-end pin: the reference's own training loop (DQN + LSTM + prioritized
# sequence replay + burn-in + double-Q, CPU) on a scripted actor stream ->
# per-learner-step loss / grad-norm series.  The GPU test replays the same
# stream through rltime_amd's loop and compares the series.
# --------------------------------------------------------------------------
E2E = dict(
    spec=dict(seed=31, num_envs=8, frame_shape=(2, 12, 12), lstm_units=8, n_actions=4, done_prob=0.08),
    model={"type": "sequential", "args": {"layer_configs": [
        {"type": "cnn", "args": {"layers": [{"filters": 4, "kernel": 4, "stride": 2},
                                            {"filters": 6, "kernel": 3, "stride": 1}]}},
        {"type": "lstm", "args": {"num_units": 8}},
        {"type": "fc", "args": {"fc_size": 16}}]}},
    policy_args={"dueling": True, "cuda": False},
    train=dict(total_steps=8 * 48, log_freq=10 ** 9, target_update_freq=64, clip_rewards=True,
               double_q=True, huber_kappa=1.0, clip_grad=10.0, adam_epsilon=1e-5, gamma=0.99,
               nstep_train=4, burn_in_timesteps=2, nstep_target=2, mbatch_size=4, lr=1e-3,
               rnn_bootstrap=True, warmup_steps=0,
               history_mode={"type": "prioritized_replay",
                             "args": {"size": 240, "train_frequency": 4, "alpha": 0.7, "beta": 0.5}}),
    seed=5)


def run_e2e_case():
    import gym
    from rltime.acting.acting_interface import ActingInterface
    from rltime.training.torch.dqn import DQN as RefDQN
    spec = StreamSpec(**E2E["spec"])

    class ScriptedActor(ActingInterface):
        def __init__(self):
            super().__init__(gym.spaces.Box(0, 255, spec.frame_shape, dtype=np.uint8),
                             gym.spaces.Discrete(spec.n_actions))
            self.t = 0

        def get_env_count(self):
            return spec.num_envs

        def set_actor_policy(self, p):
            pass

        def update_state(self, progress, policy_state=None):
            pass

        def close(self):
            pass

        def get_samples(self, min_samples):
            iters = (max(1, min_samples) + spec.num_envs - 1) // spec.num_envs
            out = []
            for step in vector_steps(spec, iters, start_step=self.t):
                out.extend(as_reference_samples(spec, step, empty_layers=(0, 2)))
            self.t += iters
            return out

    class Quiet:
        def log_result(self, *a, **k):
            pass

        def save_checkpoint(self, *a, **k):
            pass

    random.seed(E2E["seed"]); np.random.seed(E2E["seed"]); torch.manual_seed(E2E["seed"])
    tr = RefDQN(logger=Quiet(), actors=ScriptedActor(), model_config=E2E["model"],
                policy_args=E2E["policy_args"])
    series = {"qloss": [], "grad_norm": []}
    orig = tr.value_log.log

    def tap(key, value, *a, **k):
        if key in series and k.get("group") == "train":
            series[key].append(float(value))
        return orig(key, value, *a, **k)
    tr.value_log.log = tap
    init = {}
    real_init = tr.init_policies

    def init_and_snapshot():
        real_init()
        import io
        f = io.BytesIO()
        torch.save(tr.policy.state_dict(), f)
        init["online"] = np.frombuffer(f.getvalue(), dtype=np.uint8)
        f = io.BytesIO()
        torch.save(tr.target_policy.state_dict(), f)
        init["target"] = np.frombuffer(f.getvalue(), dtype=np.uint8)
    tr.init_policies = init_and_snapshot
    import copy
    tr.train(**copy.deepcopy(E2E["train"]))
    out = {"config": np.array(json.dumps(E2E)), "qloss": np.array(series["qloss"]),
           "grad_norm": np.array(series["grad_norm"]),
           "init_online": init["online"], "init_target": init["target"]}
    np.savez_compressed(os.path.join(HERE, "e2e_dqn_lstm_per.npz"), **out)
    print("e2e case: %d learner steps, qloss[0..3]=%s" % (len(series["qloss"]), series["qloss"][:4]))


# --------------------------------------------------------------------------
# End-to-end pin of the HEADLINE algorithm: recurrent IQN (dueling, double-Q,
# rnn_bootstrap, burn-in) on prioritized sequence replay, trained by the unmodified
# reference on CPU.  IQN draws its quantile fractions with torch.rand on the policy
# device (policies/torch/iqn.py:76), so a GPU run cannot reproduce the stream; the
# fixture therefore records every tau tensor the reference drew, in call order, and
# the test replays them through IQNPolicy's tau hook.
# --------------------------------------------------------------------------
E2E_IQN = dict(
    spec=E2E["spec"], model=E2E["model"],
    policy_args={"dueling": True, "cuda": False, "embedding_dim": 8, "num_sampling_quantiles": 4},
    train=dict(E2E["train"], vf_scale_epsilon=None),
    seed=6)


# The same algorithm on a model whose layer shapes are the ones the round-3 HIP kernels take, so that the GPU test
# can force EVERY hand-written product into this reference-pinned run (MIRL_GEMM3_MIN_WORK=0, MIRL_CONV3_MIN_WORK=0):
# (4,36,36) uint8 frames -> conv 32@8x8/4 (csrc/conv_in.hip, bf16 pipe) -> 64@4x4/2 -> 64@3x3/1 (csrc/conv3.hip,
# conv_mid.hip data gradient) -> LSTM 512 (csrc/lstm_seq.hip persistent sweeps: H in {128,256,512}, B % 16 == 0)
# -> quantile layer 64 -> 512 (gemm3 NT with the feature product in its epilogue) -> FC 128 | value-hidden 128
# (gemm3 NT / NN / TN).  LSTM 512 because the persistent BACKWARD sweep exists for H = 512 only.  The reference trains
# it on CPU like the small case; its initial weights are a seeded function of the parameter shapes.
E2E_IQN_WIDE = dict(
    spec=dict(seed=47, num_envs=8, frame_shape=(4, 36, 36), lstm_units=512, n_actions=6, done_prob=0.06),
    model={"type": "sequential", "args": {"layer_configs": [
        {"type": "cnn", "args": {"layers": [{"filters": 32, "kernel": 8, "stride": 4},
                                            {"filters": 64, "kernel": 4, "stride": 2},
                                            {"filters": 64, "kernel": 3, "stride": 1}]}},
        {"type": "lstm", "args": {"num_units": 512}},
        {"type": "fc", "args": {"fc_size": 128}}]}},
    policy_args={"dueling": True, "cuda": False, "embedding_dim": 64, "num_sampling_quantiles": 8},
    init_seeds=(101, 202),      # online / target weights = tests/golden/streams.py seeded_weights (no blobs in the fixture)
    train=dict(total_steps=8 * 200, log_freq=10 ** 9, target_update_freq=256, clip_rewards=True,
               double_q=True, huber_kappa=1.0, clip_grad=10.0, adam_epsilon=1e-5, gamma=0.99,
               nstep_train=6, burn_in_timesteps=4, nstep_target=2, mbatch_size=16, lr=1e-3,
               rnn_bootstrap=True, warmup_steps=0, vf_scale_epsilon=None,
               history_mode={"type": "prioritized_replay",
                             "args": {"size": 640, "train_frequency": 4, "alpha": 0.7, "beta": 0.5}}),
    seed=8)


def run_e2e_iqn_case(E2E_IQN=E2E_IQN, fname="e2e_iqn_lstm_per.npz"):
    import copy
    import gym
    import io
    from rltime.acting.acting_interface import ActingInterface
    from rltime.training.torch.iqn import IQN as RefIQN
    spec = StreamSpec(**E2E_IQN["spec"])

    class ScriptedActor(ActingInterface):
        def __init__(self):
            super().__init__(gym.spaces.Box(0, 255, spec.frame_shape, dtype=np.uint8),
                             gym.spaces.Discrete(spec.n_actions))
            self.t = 0

        def get_env_count(self):
            return spec.num_envs

        def set_actor_policy(self, p):
            pass

        def update_state(self, progress, policy_state=None):
            pass

        def close(self):
            pass

        def get_samples(self, min_samples):
            iters = (max(1, min_samples) + spec.num_envs - 1) // spec.num_envs
            out = []
            for step in vector_steps(spec, iters, start_step=self.t):
                out.extend(as_reference_samples(spec, step, empty_layers=(0, 2)))
            self.t += iters
            return out

    class Quiet:
        def log_result(self, *a, **k):
            pass

        def save_checkpoint(self, *a, **k):
            pass

    random.seed(E2E_IQN["seed"]); np.random.seed(E2E_IQN["seed"]); torch.manual_seed(E2E_IQN["seed"])
    tr = RefIQN(logger=Quiet(), actors=ScriptedActor(), model_config=E2E_IQN["model"],
                policy_args=E2E_IQN["policy_args"])
    series = {"qloss": [], "grad_norm": []}
    orig = tr.value_log.log

    def tap(key, value, *a, **k):
        if key in series and k.get("group") == "train":
            series[key].append(float(value))
        return orig(key, value, *a, **k)
    tr.value_log.log = tap
    init = {}
    real_init = tr.init_policies
    taus, real_rand = [], torch.rand

    def init_and_snapshot():
        real_init()
        seeds = E2E_IQN.get("init_seeds")
        for i, (key, pol) in enumerate((("online", tr.policy), ("target", tr.target_policy))):
            if seeds is not None:
                from tests.golden.streams import seeded_weights
                pol.load_state_dict(seeded_weights(pol.state_dict(), seeds[i]))
                init[key] = np.zeros(0, dtype=np.uint8)
                continue
            f = io.BytesIO()
            torch.save(pol.state_dict(), f)
            init[key] = np.frombuffer(f.getvalue(), dtype=np.uint8)

        def logged_rand(*a, **k):            # only the IQN quantile draws call torch.rand from here on
            out = real_rand(*a, **k)
            taus.append(out.detach().cpu().numpy().copy())
            return out
        torch.rand = logged_rand
    tr.init_policies = init_and_snapshot
    # double-Q near-ties: per learner step, the smallest gap between the best and the second-best action of the
    # selection scores (mean over quantiles, training/torch/iqn.py:36-45) over the batch rows, next to the scores'
    # magnitude.  A float32 re-implementation can resolve a row whose gap is within rounding the other way; the GPU
    # test exempts exactly those steps from its strict gradient-norm bar (tests/test_e2e_gpu.py).
    gaps = {"min": [], "scale": [], "second": []}
    real_boot = tr._get_bootstrap_target_value
    real_argmax = torch.Tensor.argmax

    def boot_with_gap(target_states, timesteps):
        seen = []

        def argmax_tap(self, *a, **k):
            if self.dim() == 2 and k.get("keepdim"):
                seen.append(self.detach().clone())
            return real_argmax(self, *a, **k)
        torch.Tensor.argmax = argmax_tap
        try:
            out = real_boot(target_states, timesteps)
        finally:
            torch.Tensor.argmax = real_argmax
        assert len(seen) == 1, len(seen)
        top = seen[0].topk(2, dim=1).values
        gap = (top[:, 0] - top[:, 1]).sort().values
        gaps["min"].append(float(gap[0]))
        gaps["second"].append(float(gap[1]))
        gaps["scale"].append(float(seen[0].abs().max()))
        return out
    tr._get_bootstrap_target_value = boot_with_gap
    try:
        tr.train(**copy.deepcopy(E2E_IQN["train"]))
    finally:
        torch.rand = real_rand
    assert all(t.ndim == 1 for t in taus)
    assert len(gaps["min"]) == len(series["qloss"])
    out = {"config": np.array(json.dumps(E2E_IQN)), "qloss": np.array(series["qloss"]),
           "grad_norm": np.array(series["grad_norm"]),
           "init_online": init["online"], "init_target": init["target"],
           "tau_sizes": np.array([len(t) for t in taus], dtype=np.int64),
           "taus": np.concatenate(taus).astype(np.float32),
           "sel_gap_min": np.array(gaps["min"]), "sel_gap_second": np.array(gaps["second"]), "sel_scale": np.array(gaps["scale"])}
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print("e2e IQN case " + fname + ": %d learner steps, %d tau draws (%d values), qloss[0..3]=%s" % (
        len(series["qloss"]), len(taus), out["taus"].size, series["qloss"][:4]))


# --------------------------------------------------------------------------
# Schedules: epsilon-greedy (exploration/epsilon_greedy.py:64-99) and the linear
# anneal used for beta / LR (general/utils.py:85-103)
# --------------------------------------------------------------------------
EXPLORE_CASES = {
    "decay": dict(eps_start=1.0, eps_final=0.01, exploration_fraction=0.1),
    "apex": dict(eps_start=0.4, eps_final=0.01, eps_min=0.01, per_actor_exponent_factor=7,
                 exploration_fraction=0.5),
    "a3c": dict(eps_final=[0.1, 0.01, 0.5], eps_prob=[0.4, 0.3, 0.3], exploration_fraction=0.3),
}


def run_schedule_cases():
    import gym
    from rltime.exploration.epsilon_greedy import EpsilonGreedyExplorationManager as RefEps
    from rltime.general.utils import anneal_value
    out = {"cases": np.array(json.dumps(EXPLORE_CASES))}
    space = gym.spaces.Discrete(6)
    E = 16
    for name, kw in EXPLORE_CASES.items():
        m = RefEps(**kw, total_actors=E)
        for pi, progress in enumerate((0.0, 0.03, 0.25, 0.5, 0.9, 1.0)):
            np.random.seed(100 + pi)
            acts, info = m.remap_actions(np.arange(E) % 6, list(range(E)), space, progress)
            out["%s.p%d.actions" % (name, pi)] = np.array(acts)
            out["%s.p%d.eps" % (name, pi)] = info["eps"]
    grid = np.array([0.0, 0.1, 0.5, 0.99, 1.0, 1.7])
    out["anneal.progress"] = grid
    out["anneal.true"] = np.array([anneal_value(0.4, p, True, 1.0) for p in grid])
    out["anneal.to"] = np.array([anneal_value(3e-4, p, 1e-5) for p in grid])
    out["anneal.off"] = np.array([anneal_value(0.6, p, False) for p in grid])
    np.savez_compressed(os.path.join(HERE, "schedule_cases.npz"), **out)
    print("schedule cases: %d arrays" % len(out))


# --------------------------------------------------------------------------
# Config loader semantics (general/config.py:32-117): '@json', '@python', '_'
# comment keys, '**' shallow merge, '***' deep merge, nested key paths.
# --------------------------------------------------------------------------
def run_config_cases():
    from rltime.general.config import load_config
    cfg_dir = os.path.join(HERE, "configs")
    out = {}
    for name in sorted(os.listdir(cfg_dir)):
        if not name.endswith(".json"):
            continue
        loaded = load_config(os.path.join(cfg_dir, name))

        def enc(o):
            if callable(o):
                return "<python:%s.%s>" % (o.__module__.split(".", 1)[-1], o.__name__)
            raise TypeError(o)
        out[name] = json.loads(json.dumps(loaded, default=enc))
    with open(os.path.join(HERE, "config_cases.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("config cases: %s" % ", ".join(out))


# --------------------------------------------------------------------------
# Plugin-surface conformance: the parameter names of the reference's plugin
# classes / methods, taken from the imported reference with
# inspect.signature.  tests/test_abi.py holds rltime_amd's mirror classes to them
# (same names in the same order; the mirror may only APPEND optional parameters).
# --------------------------------------------------------------------------
SIGNATURE_TARGETS = {
    "history.History": ("rltime.history.history", "History",
                        ["__init__", "update", "needed_feed_count", "get_train_data", "update_losses"]),
    "history.ReplayHistoryBuffer": ("rltime.history.replay_history", "ReplayHistoryBuffer", ["__init__"]),
    "history.PrioritizedReplayHistoryBuffer": ("rltime.history.prioritized_replay_history",
                                               "PrioritizedReplayHistoryBuffer", ["__init__", "update_losses"]),
    "policies.Policy": ("rltime.policies.policy", "Policy",
                        ["actor_predict", "get_creator", "get_state", "is_recurrent", "make_input_state",
                         "get_state_store"]),
    "policies.TorchPolicy": ("rltime.policies.torch.torch_policy", "TorchPolicy",
                             ["create", "copy_from", "get_grad_norm", "make_tensor", "get_creator", "load_state"]),
    "policies.DQNPolicy": ("rltime.policies.torch.dqn", "DQNPolicy", ["__init__", "predict", "actor_predict"]),
    "policies.IQNPolicy": ("rltime.policies.torch.iqn", "IQNPolicy", ["__init__"]),
    "training.PolicyTrainer": ("rltime.training.policy_trainer", "PolicyTrainer",
                               ["__init__", "create_policy", "sample_actors", "train"]),
    "training.MultiStepTrainer": ("rltime.training.multi_step_trainer", "MultiStepTrainer",
                                  ["calc_target_values", "train_init", "set_lr", "train_batch", "_burn_in", "_train"]),
    "training.TorchTrainer": ("rltime.training.torch.torch_trainer", "TorchTrainer",
                              ["_train", "_compute_grads", "_get_bootstrap_target_value"]),
    "training.DQN": ("rltime.training.torch.dqn", "DQN", ["_train"]),
    "acting.ActingInterface": ("rltime.acting.acting_interface", "ActingInterface",
                               ["get_spaces", "get_samples", "get_env_count", "set_actor_policy", "update_state",
                                "close", "_create_sample"]),
    "acting.Actor": ("rltime.acting.actor", "Actor", ["get_samples", "update_state", "set_actor_policy"]),
    "models.SequentialModel": ("rltime.models.torch.sequential", "SequentialModel",
                               ["make_input_state", "forward", "set_layer_preprocessor"]),
    # BASELINE configs[0] (cartpole_ppo.json): the on-policy plumbing chain
    "history.OnlineHistoryBuffer": ("rltime.history.online_history", "OnlineHistoryBuffer",
                                    ["__init__", "update", "needed_feed_count", "get_train_data"]),
    "policies.ActorCriticPolicy": ("rltime.policies.torch.actor_critic", "ActorCriticPolicy",
                                   ["__init__", "actor_predict", "get_state_value", "evaluate_actions",
                                    "get_dist_and_state_value"]),
    "training.A2C": ("rltime.training.torch.a2c", "A2C",
                     ["_train", "create_policy", "_get_discount_function", "_discount_bootstrap_target_value",
                      "_get_bootstrap_target_value", "_calc_action_gain", "_compute_grads"]),
    "training.PPO": ("rltime.training.torch.ppo", "PPO", ["_train", "_calc_action_gain"]),
}


def run_signature_cases():
    import importlib
    import inspect
    out = {}
    for key, (module, cls_name, methods) in SIGNATURE_TARGETS.items():
        cls = getattr(importlib.import_module(module), cls_name)
        table = {}
        for m in methods:
            params = []
            for name, p in inspect.signature(getattr(cls, m)).parameters.items():
                kind = {p.VAR_POSITIONAL: "*", p.VAR_KEYWORD: "**"}.get(p.kind, "")
                params.append([kind + name, p.default is not p.empty])
            table[m] = params
        out[key] = table
    with open(os.path.join(HERE, "signatures.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("signature cases: %d classes" % len(out))


# --------------------------------------------------------------------------
# On-policy plumbing (BASELINE configs[0], cartpole_ppo.json): OnlineHistoryBuffer batches with the GAE discount of
# a2c.py:48-66, targets through calc_target_values and the PPO loss / gradients through PPO._compute_grads of the
# unmodified reference on a seeded-weight 2x16 MLP ActorCriticPolicy
# --------------------------------------------------------------------------
def run_online_ppo_case():
    import gym
    from rltime.history.online_history import OnlineHistoryBuffer
    from rltime.training.torch.ppo import PPO
    from rltime.policies.torch.actor_critic import ActorCriticPolicy
    from tests.golden.streams import ONLINE_CASE as case, online_vector_steps, seeded_weights
    T, gamma, lam = case["nstep_train"], case["gamma"], case["advlam"]
    tr = PPO.__new__(PPO)
    tr.gamma, tr.advlam, tr.vf_coef, tr.adv_norm = gamma, lam, case["vf_coef"], True
    tr.entropy_factor, tr.entropy_anneal = case["entropy_factor"], None
    tr._clip_value, tr._clip_anneal = case["clip_value"], None
    tr.vf_scale_epsilon = None
    tr.steps, tr.total_steps = 0, 1
    tr.value_log = ValueLog()
    policy = ActorCriticPolicy.create(
        model_config=case["model"], observation_space=gym.spaces.Box(-10, 10, (case["obs_dim"],), np.float32),
        action_space=gym.spaces.Discrete(case["n_actions"]), cuda=False)
    policy.load_state_dict(seeded_weights(policy.state_dict(), case["weights_seed"]))
    tr.policy = tr.target_policy = policy
    ref = OnlineHistoryBuffer(nstep_target=T, nstep_train=T, discount_function=tr._get_discount_function(gamma),
                              state_store=StateStore("cpu"))
    mine = orc.OracleOnline(nstep_target=T, nstep_train=T, discount_function=orc.make_gae_discount(gamma, lam))
    out, step_no, rnd = {}, 0, 0
    for op in case["script"]:
        if op[0] == "feed":
            for a, b in zip(online_vector_steps(case, op[1], step_no), online_vector_steps(case, op[1], step_no)):
                assert ref.update(a) == mine.update(b)
            step_no += op[1]
            continue
        tag = "r%d" % rnd
        rnd += 1
        B = op[1]
        assert ref.needed_feed_count(B, case["num_envs"]) == mine.needed_feed_count(B, case["num_envs"])
        out[tag + ".feed_count"] = np.array(-1 if ref.needed_feed_count(B, case["num_envs"]) is None else case["num_envs"])
        got, got_mine = ref.get_train_data(B), mine.get_train_data(B)
        out[tag + ".is_none"] = np.array(got is None)
        assert (got is None) == (got_mine is None)
        if got is None:
            continue
        deep_equal(got, got_mine, tag)
        assert ref.last_env == mine.last_env
        flat = {}
        flatten("", got, flat)
        for k, v in flat.items():
            out[tag + ".batch." + k] = v
        # multi_step_trainer.py:290-320: flatten (T, B) -> (T*B), targets, one training pass over the whole batch
        f = lambda x: x.reshape((x.shape[0] * x.shape[1],) + tuple(x.shape[2:]))                # noqa: E731
        data = {k: orc.tree_map(v, f) for k, v in got.items() if k != "extra_data"}
        targets = tr.calc_target_values(data["returns"], data["target_states"], data["target_masks"], data["nsteps"], 1)
        policy.zero_grad()
        tr._compute_grads(data["states"], targets, data["policy_outputs"], {}, 1)
        out[tag + ".targets"] = targets.detach().numpy()
        for name, prm in policy.named_parameters():
            out[tag + ".grad." + name] = prm.grad.detach().numpy().copy()
        log = tr.value_log.get()["train"]
        for key in ("value_loss", "policy_loss", "policy_entropy", "state_value_mean"):
            out[tag + ".log." + key] = np.array(log[key])
        # the oracle's arithmetic (oracle/qmath.py) on the same tensors
        with torch.no_grad():
            boot = policy.get_state_value(policy.make_tensor(data["target_states"]), 1)
            mk = policy.make_tensor
            y = mk(data["returns"]) + qmath.gae_bootstrap_discount(boot, mk(data["nsteps"]), gamma, lam) * mk(data["target_masks"])
        assert torch.equal(y, targets)
        lp, vals, ent = policy.evaluate_actions(policy.make_tensor(data["states"]), 1, data["policy_outputs"]["actions"])
        total, vloss, gain = qmath.actor_critic_loss(
            lp, vals, ent, y, mk(data["policy_outputs"]["values"]), mk(data["policy_outputs"]["action_log_probs"]),
            case["vf_coef"], case["entropy_factor"], True, case["clip_value"])
        assert abs(float(vloss) - float(log["value_loss"])) <= 1e-6 * abs(float(vloss))
        assert abs(-float(gain) - float(log["policy_loss"])) <= 1e-6
    out["rounds"] = np.array(rnd)
    np.savez_compressed(os.path.join(HERE, "online_ppo.npz"), **out)
    print("online / PPO plumbing case: %d draws, %d keys" % (rnd, len(out)))


if __name__ == "__main__":
    torch.manual_seed(0)
    run_tree_cases()
    for name, cfg in SCENARIOS.items():
        run_replay_scenario(name, cfg)
    run_qmath_cases()
    run_model_cases()
    run_network64_pin()
    run_e2e_case()
    run_e2e_iqn_case()
    run_e2e_iqn_case(E2E_IQN_WIDE, "e2e_iqn_lstm_per_wide.npz")
    run_schedule_cases()
    run_config_cases()
    run_signature_cases()
    run_online_ppo_case()
    print("golden fixtures written to", HERE)
