"""CPU: the host half of librltime_hip — ring / FIFO / free-list bookkeeping
(csrc/book.hpp) and the NumPy-promotion arithmetic model (csrc/np_emul.h) —
against the golden vectors of the unmodified reference.  No GPU, no kernels."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from rltime_amd import _lib
from rltime_amd._lib import lib, check, np_ptr
from tests import scenario

GOLDEN = scenario.GOLDEN


def make_cfg(cfg):
    h = cfg["hist"]
    per = cfg["mode"] == "per"
    ba = h.get("beta_anneal", False)
    return _lib.ReplayConfig(
        size=h["size"], num_envs=cfg["spec"]["num_envs"], env_base=cfg["spec"].get("env_base", 0), frame_bytes=16,
        nstep_train=h["nstep_train"], prefix_steps=h["prefix_steps"],
        nstep_target=h["nstep_target"], gamma=cfg["gamma"],
        mode=_lib.MODE_PER if per else _lib.MODE_UNIFORM,
        train_frequency=h["train_frequency"],
        avoid_episode_crossing=int(h.get("avoid_episode_crossing", False)),
        overlap=_lib.INT32_MIN if h.get("overlap") is None else h["overlap"],
        alpha=h.get("alpha", 0.6), beta=h.get("beta", 0.4), eps=1e-6,
        max_weight_factor=h.get("max_weight_factor", 0.9),
        beta_anneal_mode=0 if ba is False else (1 if ba is True else 2),
        beta_anneal_to=1.0 if isinstance(ba, bool) else float(ba),
        global_importance_scaling=int(h.get("global_importance_scaling", False)))


class Book:
    def __init__(self, cfg):
        self.h = C.c_void_p()
        self.cfg = make_cfg(cfg)
        check(lib.mirl_book_create(C.byref(self.cfg), C.byref(self.h)))
        self.E = cfg["spec"]["num_envs"]

    def __del__(self):
        lib.mirl_book_destroy(self.h)

    def stats(self):
        v = [C.c_int64() for _ in range(5)]
        check(lib.mirl_book_stats(self.h, *[C.byref(x) for x in v]))
        return [x.value for x in v]


@pytest.mark.parametrize("name", scenario.SCENARIOS)
def test_bookkeeping_follows_reference(name):
    gold, cfg = scenario.load(name)
    b = Book(cfg)
    E = b.E
    per = cfg["mode"] == "per"
    rnd = 0
    for op in cfg["script"]:
        if op[0] == "feed":
            for _ in range(op[1]):
                check(lib.mirl_book_ingest(b.h, E, None))
            continue
        tag = "r%d" % rnd
        rnd += 1
        if op[0] == "draw":
            B = op[1]
            out = C.c_int64()
            check(lib.mirl_book_needed_feed_count(b.h, B, E, C.byref(out)))
            assert out.value == int(gold[tag + ".feed_count"]), tag
            check(lib.mirl_book_charge_quota(b.h, B))
            total, active, quota, cap, n_slots = b.stats()
            assert quota == int(gold[tag + ".quota_after"]), tag
            if not per and not bool(gold[tag + ".is_none"]) and \
                    not cfg["hist"].get("avoid_episode_crossing"):
                picks = np.ascontiguousarray(gold[tag + ".picks"], dtype=np.int64)
                env = np.zeros(B, dtype=np.int32)
                start = np.zeros(B, dtype=np.int64)
                check(lib.mirl_book_uniform_map(b.h, B, np_ptr(picks), np_ptr(env), np_ptr(start)))
                first = np.zeros(E, dtype=np.int64)
                check(lib.mirl_book_env_meta(b.h, np_ptr(first), None))
                win = gold[tag + ".windows"]          # (env, ring position)
                assert np.array_equal(env + cfg["spec"].get("env_base", 0), win[:, 0]), tag
                assert np.array_equal(start - first[env], win[:, 1]), tag
        if per:
            total, active, quota, cap, n_slots = b.stats()
            n = C.c_int64()
            free = np.zeros(n_slots + 1, dtype=np.int32)
            check(lib.mirl_book_free_slots(b.h, np_ptr(free), C.byref(n)))
            assert np.array_equal(free[:n.value], gold[tag + ".free_slots"]), tag
            se = np.zeros(n_slots, dtype=np.int32)
            sb = np.zeros(n_slots, dtype=np.int64)
            check(lib.mirl_book_slot_table(b.h, np_ptr(se), np_ptr(sb)))
            se = se.astype(np.int64)
            se[se >= 0] += cfg["spec"].get("env_base", 0)          # the book keeps local env indices
            assert np.array_equal(se, gold[tag + ".slot_env"]), tag
            assert np.array_equal(sb, gold[tag + ".slot_base"]), tag
            first = np.zeros(E, dtype=np.int64)
            check(lib.mirl_book_env_meta(b.h, np_ptr(first), None))
            assert np.array_equal(first, gold[tag + ".env_first"]), tag
            assert cap * 2 == len(gold[tag + ".node_val"])
            assert active == n_slots - n.value


def _emul_tree(leaf_val, leaf_kind):
    cap = len(leaf_val)
    nv = np.zeros(2 * cap, dtype=np.float64)
    nk = np.zeros(2 * cap, dtype=np.uint8)
    lv = np.ascontiguousarray(leaf_val, dtype=np.float64)
    lk = np.ascontiguousarray(leaf_kind, dtype=np.uint8)
    check(lib.mirl_emul_build_tree(cap, np_ptr(lv), np_ptr(lk), np_ptr(nv), np_ptr(nk)))
    return nv, nk


def test_promotion_model_tree_cases():
    """Heap built by np_emul.h == the reference's list-of-scalars heap (values
    AND scalar kinds), and stratified descents land on the same leaves."""
    d = np.load(os.path.join(GOLDEN, "tree_cases.npz"))
    for case in json.loads(str(d["cases"])):
        nv, nk = _emul_tree(d[case + ".leaf_val"], d[case + ".leaf_kind"])
        assert np.array_equal(nv[1:], d[case + ".node_val"][1:]), case
        assert np.array_equal(nk[1:], d[case + ".node_kind"][1:]), case
        cap = int(d[case + ".capacity"])
        for B in (8, 32):
            us = np.ascontiguousarray(d["%s.B%d.uniforms" % (case, B)])
            idx = np.zeros(B, dtype=np.int64)
            check(lib.mirl_emul_find(cap, np_ptr(nv), np_ptr(nk), B, np_ptr(us), np_ptr(idx)))
            assert np.array_equal(idx, d["%s.B%d.index" % (case, B)]), (case, B)


@pytest.mark.parametrize("name", ["per_t1", "per_seq", "per_seq_global"])
def test_promotion_model_on_live_trees(name):
    """Every intermediate tree of the PER scenarios: rebuild from the golden
    leaves, compare all inner nodes, and re-run the recorded draws."""
    gold, cfg = scenario.load(name)
    for r in range(int(gold["rounds"])):
        tag = "r%d" % r
        nv, nk = _emul_tree(gold[tag + ".leaf_val"], gold[tag + ".leaf_kind"])
        assert np.array_equal(nv[1:], gold[tag + ".node_val"][1:]), tag
        assert np.array_equal(nk[1:], gold[tag + ".node_kind"][1:]), tag
    # draws use the tree as it was BEFORE the draw == state after the previous op
    prev = None
    for r in range(int(gold["rounds"])):
        tag = "r%d" % r
        if str(gold[tag + ".op"]) == "draw" and (tag + ".slots") in gold.files:
            # a draw does not modify the tree, so the post-op state is the pre-op state
            nv, nk = _emul_tree(gold[tag + ".leaf_val"], gold[tag + ".leaf_kind"])
            us = np.ascontiguousarray(gold[tag + ".uniforms"])
            B = len(us)
            idx = np.zeros(B, dtype=np.int64)
            check(lib.mirl_emul_find(len(nv) // 2, np_ptr(nv), np_ptr(nk), B, np_ptr(us), np_ptr(idx)))
            assert np.array_equal(idx, gold[tag + ".slots"]), tag
        prev = tag
    assert prev is not None


def _np_priority(slots, T, alpha, mwf):
    """prioritized_replay_history.py:188-203 evaluated with NumPy scalars."""
    vals = [1.0 if s < 0 else np.float32(s) for s in slots]
    if T == 1:
        mixed = vals[0]
    else:
        mixed = mwf * np.max(vals) + (1 - mwf) * np.mean(vals)
    return mixed ** alpha


@pytest.mark.parametrize("T", [1, 2, 4, 7, 8, 9, 20, 80, 128, 129, 200, 513])
def test_sequence_priority_matches_numpy(T):
    """np.max / np.mean pairwise summation / scalar promotion: kinds exact,
    values within 1 ulp (pow is libm on the host here and ocml on the GPU)."""
    from tests.golden.streams import scalar_kind
    rng = np.random.RandomState(T)
    for trial in range(60):
        slots = (np.abs(rng.randn(T)) + 1e-6).astype(np.float32)
        if trial % 3 == 1:
            slots[rng.randint(0, T, size=max(1, T // 5))] = -1.0   # never-updated steps
        if trial % 3 == 2:
            slots[:] = -1.0
        alpha, mwf = (0.6, 0.9) if trial % 2 else (0.9, 0.8)
        want = _np_priority(slots, T, alpha, mwf)
        v, k = C.c_double(), C.c_uint8()
        check(lib.mirl_emul_seq_priority(T, alpha, mwf, np_ptr(slots), C.byref(v), C.byref(k)))
        assert k.value == scalar_kind(want), (T, trial)
        rtol = 1.3e-7 if k.value == 1 else 4.5e-16
        assert abs(v.value - float(want)) <= rtol * float(want), (T, trial, v.value, float(want))


def test_global_sampling_rows_bound_every_stratum_fits():
    """Exact global sampling sizes a rank's padded batch from the exchanged shard totals
    (history/replay_history.py:global_sampling_rows): for ANY table of shard masses and ANY
    uniforms, the strata whose mass point falls into a shard's cumulative range number at most
    `bound` — so no stratum can be dropped, whatever the imbalance."""
    from rltime_amd.history.replay_history import global_sampling_rows
    rs = np.random.RandomState(0)
    for trial in range(400):
        R = int(rs.choice([2, 3, 4, 8]))
        B = int(rs.choice([4, 8, 64, 512]))
        P = np.abs(rs.randn(R)) ** rs.choice([1, 3]) + 1e-9
        if trial % 5 == 0:
            P[rs.randint(R)] *= 1000.0                      # one shard owns almost all the mass
        Bg, Pg = B * R, P.sum()
        seg = Pg / Bg
        u = rs.rand(Bg)
        if trial % 7 == 0:
            u[:] = rs.choice([0.0, 1.0 - 1e-12])            # all points at a stratum edge
        mass = (u + np.arange(Bg)) * seg
        edges = np.concatenate([[0.0], np.cumsum(P)])
        edges[-1] = np.inf
        total = 0
        for r in range(R):
            mine = int(((mass >= edges[r]) & (mass < edges[r + 1])).sum())
            rows, bound = global_sampling_rows(B, R, P[r] / Pg)
            assert mine <= bound <= rows, (trial, r, mine, bound, rows)
            assert rows % max(4, B // 8) == 0
            total += mine
        assert total == Bg


def test_bench_overlap_rule_and_strong_shares():
    """bench.py: the per-rank config of the strong-scaling line (global B = 512, 256 envs, 1M transitions
    split over the ranks) and the acting-overlap rule (on when a rank trains <= 8192 rows per step)."""
    import argparse
    import bench
    base = dict(config="iqn_lstm", mbatch=None, nstep_train=None, burn_in=None, nstep_target=None, envs=None,
                replay_size=1000000, train_arg=[], frame_dedup=False, no_acting=False, overlap_acting="auto")
    for world, want_b, want_e, want_overlap in ((1, 512, 256, False), (2, 256, 128, False), (4, 128, 64, False), (8, 64, 32, True)):
        cfg = bench.build_config(argparse.Namespace(**base), world - 1, world, "strong")
        ta = cfg["training"]["args"]
        assert ta["mbatch_size"] == want_b and cfg["acting"]["actor_envs"] == want_e
        assert ta["history_mode"]["args"]["size"] == 1000000 // world
        assert cfg["acting"]["env_base"] == (world - 1) * want_e and bool(ta["overlap_acting"]) == want_overlap
    weak = bench.build_config(argparse.Namespace(**base), 3, 8, "weak")
    assert weak["training"]["args"]["mbatch_size"] == 512 and not weak["training"]["args"]["overlap_acting"]
    forced = bench.build_config(argparse.Namespace(**dict(base, overlap_acting="on")), 0, 1, "strong")
    assert forced["training"]["args"]["overlap_acting"]
