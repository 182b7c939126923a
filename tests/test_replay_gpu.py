"""GPU parity: the HIP replay shard (through the C-ABI, via the reference-API
classes in rltime_amd.history) against the golden vectors of the unmodified
reference and against the oracle on larger seeded streams.

Bars: bit-exact for frames / recurrent state / actions / nsteps / masks /
returns / loss indices / sampled windows / tree kinds / free-list order;
importance weights rtol 2e-6; leaf priorities <= 1 ulp (pow)."""
import ctypes as C
import random

import numpy as np
import pytest
import torch

from tests import scenario
from tests.golden.streams import StreamSpec, vector_steps, as_reference_samples

pytestmark = pytest.mark.gpu


def _make(cfg, gamma):
    from rltime_amd.history import ReplayHistoryBuffer, PrioritizedReplayHistoryBuffer
    cls = PrioritizedReplayHistoryBuffer if cfg["mode"] == "per" else ReplayHistoryBuffer
    return cls(**cfg["hist"], gamma=gamma)


def _per_state(buf):
    from rltime_amd.history import PrioritizedReplayHistoryBuffer
    if not isinstance(buf, PrioritizedReplayHistoryBuffer) or buf._h is None:
        return None
    v, k, _ = buf.tree_nodes()
    cap = len(v) // 2
    se, sb = buf.slot_table()
    first, _ = buf.env_meta()
    se = se.astype(np.int64)
    se[se >= 0] += buf._env_base
    return {"leaf_val": v[cap:], "leaf_kind": k[cap:], "leaf_exact": False,
            "free_slots": buf.free_slots().astype(np.int64),
            "slot_env": se, "slot_base": sb, "env_first": first}


def _check_windows(tag, buf, gold, last_batch):
    if str(gold[tag + ".op"]) != "draw" or bool(gold[tag + ".is_none"]):
        return
    win = gold[tag + ".windows"]
    env = buf.last_sample["env"].cpu().numpy().astype(np.int64) + buf._env_base
    start = buf.last_sample["start"].cpu().numpy()
    first = np.zeros(buf._num_envs, dtype=np.int64)
    from rltime_amd._lib import lib, check, np_ptr
    check(lib.mirl_replay_env_meta(buf._h, np_ptr(first), None))
    assert np.array_equal(env, win[:, 0]), tag
    assert np.array_equal(start - first[env - buf._env_base], win[:, 1]), tag
    if (tag + ".slots") in gold.files:
        assert np.array_equal(buf.last_sample["slot"].cpu().numpy(), gold[tag + ".slots"]), tag


@pytest.mark.parametrize("feed", sorted(scenario.FEEDS))
@pytest.mark.parametrize("name", scenario.SCENARIOS)
def test_golden_scenarios(name, feed):
    """Every reference-generated scenario through each of the three hand-overs the build has: the reference's per-env
    dict lists (actor.py:132-145 -> history.py:123-176), the device actor's DeviceSamples, and the fused rollout's
    planned ingest (mirl_replay_ingest_plan + mirl_replay_ingest_planned) — same golden batches, windows, tree kinds
    and free-list order after every operation."""
    scenario.run(name, _make, exact_dtypes=False, per_state=_per_state,
                 on_round=_check_windows, feeder=scenario.FEEDS[feed])


def test_tree_cases_on_device():
    """Injected leaves (value + scalar kind) -> every inner node and every
    stratified index bit-identical to the reference's list-of-scalars tree."""
    import json, os
    from rltime_amd.history import PrioritizedReplayHistoryBuffer
    from rltime_amd._lib import lib, check, np_ptr
    d = np.load(os.path.join(scenario.GOLDEN, "tree_cases.npz"))
    for case in json.loads(str(d["cases"])):
        cap = int(d[case + ".capacity"])
        buf = PrioritizedReplayHistoryBuffer(
            size=cap, train_frequency=4, nstep_target=1, nstep_train=1, gamma=0.99)
        buf.configure({"x": np.zeros((16,), np.uint8)}, num_envs=1)
        lv = np.ascontiguousarray(d[case + ".leaf_val"])
        lk = np.ascontiguousarray(d[case + ".leaf_kind"])
        check(lib.mirl_replay_tree_set_leaves(buf._h, cap, np_ptr(lv), np_ptr(lk), None))
        v, k, _ = buf.tree_nodes()
        assert np.array_equal(v[1:], d[case + ".node_val"][1:]), case
        assert np.array_equal(k[1:], d[case + ".node_kind"][1:]), case
        for B in (8, 32):
            us = np.ascontiguousarray(d["%s.B%d.uniforms" % (case, B)])
            idx = np.zeros(B, dtype=np.int64)
            check(lib.mirl_replay_tree_find(buf._h, B, np_ptr(us), np_ptr(idx), None))
            assert np.array_equal(idx, d["%s.B%d.index" % (case, B)]), (case, B)
        buf.close()


def _run_pair(seed, E, steps_script, hist, gamma, per, spec_kw, B, dev_kw=None):
    """Same seeded stream through the oracle and the HIP buffer."""
    from oracle import replay as orc
    from rltime_amd.history import ReplayHistoryBuffer, PrioritizedReplayHistoryBuffer
    spec = StreamSpec(seed=seed, num_envs=E, **spec_kw)
    o_cls = orc.OraclePrioritizedReplay if per else orc.OracleReplay
    d_cls = PrioritizedReplayHistoryBuffer if per else ReplayHistoryBuffer
    ora = o_cls(**hist, discount_function=orc.make_discount(gamma))
    dev = d_cls(**hist, gamma=gamma, **(dev_kw or {}))
    step_no = 0
    mism = {"leaf": 0, "leaves": 0, "draws": 0}
    for op in steps_script:
        if op[0] == "feed":
            for st in vector_steps(spec, op[1], start_step=step_no):
                ora.update(as_reference_samples(spec, st))
                dev.update(as_reference_samples(spec, st))
            step_no += op[1]
            continue
        s = op[1]
        random.seed(s); np.random.seed(s)
        a = ora.get_train_data(B, train_progress=op[2])
        random.seed(s); np.random.seed(s)
        b = dev.get_train_data(B, train_progress=op[2])
        assert (a is None) == (b is None)
        if a is None:
            continue
        mism["draws"] += 1
        fa = scenario.flatten("", a, {})
        fb = {k: scenario.to_numpy(v) for k, v in scenario.flatten("", b, {}).items()}
        assert set(fa) == set(fb)
        for key, w in fa.items():
            g = fb[key]
            if key.endswith("importance_weights"):
                np.testing.assert_allclose(g, w.astype(np.float32), rtol=scenario.WEIGHT_RTOL, err_msg=key)
            elif key.endswith("actions") or key.endswith("loss_indices"):
                assert np.array_equal(g, w), key
            else:
                assert np.array_equal(g, scenario.make_tensor_dtype(w)), key
        if per:
            P = hist["prefix_steps"]
            idx = a["extra_data"]["loss_indices"][P:].reshape(-1, 2)
            rng = np.random.RandomState(s + 7)
            losses = (rng.randn(idx.shape[0]) * 0.8).astype(np.float32)
            ora.update_losses(idx, losses)
            # device path takes device tensors straight from the batch
            dev.update_losses(b["extra_data"]["loss_indices"][P:].reshape(-1, 2),
                              torch.from_numpy(losses).cuda())
            v, k, _ = dev.tree_nodes()
            cap = ora.tree.capacity
            want = np.array([float(x) for x in ora.tree.nodes[cap:]])
            from tests.golden.streams import scalar_kind
            wk = np.array([scalar_kind(x) for x in ora.tree.nodes[cap:]], dtype=np.uint8)
            assert np.array_equal(k[cap:], wk)
            scenario.check_leaf_values(v[cap:], want, wk, "seed%d" % s, False)
            mism["leaf"] += int(np.sum(v[cap:] != want))
            mism["leaves"] += len(want)
            assert np.array_equal(dev.free_slots(), np.array(list(ora.free_slots)))
    dev.close()
    return mism


def test_uniform_stream_vs_oracle_atari_frames():
    """84x84x4 frames, T=1 DQN shape, ring wrap and eviction."""
    _run_pair(21, 8, [("feed", 40), ("draw", 1, None), ("feed", 200), ("draw", 2, None),
                      ("feed", 77), ("draw", 3, None)],
              dict(size=1024, train_frequency=0, nstep_target=1, nstep_train=1, prefix_steps=0),
              0.99, False, dict(frame_shape=(4, 84, 84), n_actions=6, done_prob=0.01), 64)


def test_per_sequences_vs_oracle_recurrent():
    """R2D2-shaped: T=16, burn-in 8, n=2, overlap 8, LSTM state 2x32, eviction;
    losses fed back 6 times.  Sampled windows must stay identical throughout."""
    m = _run_pair(22, 6, [("feed", 60), ("draw", 11, 0.0), ("feed", 30), ("draw", 12, 0.2),
                          ("draw", 13, 0.3), ("feed", 120), ("draw", 14, 0.5), ("feed", 45),
                          ("draw", 15, 0.8), ("draw", 16, 1.0)],
                  dict(size=900, train_frequency=4, nstep_target=2, nstep_train=16, prefix_steps=8,
                       alpha=0.9, beta=0.6, max_weight_factor=0.9),
                  0.997, True, dict(frame_shape=(1, 20, 20), lstm_units=32, n_actions=6, done_prob=0.03), 16)
    assert m["draws"] == 6
    print("PER leaves differing from the reference by 1 ulp: %d / %d" % (m["leaf"], m["leaves"]))


@pytest.mark.parametrize("T,P,n", [(9, 3, 2), (20, 5, 3), (80, 40, 2), (128, 4, 1)])
def test_wave_per_sequence_priorities_vs_oracle(T, P, n):
    """k_recalc_flagged_wave (8 <= T <= 128: one wave per sequence, NumPy's eight
    interleaved accumulators kept in lanes 0..7, shuffle-combined in NumPy's
    association) against the reference restatement: leaf kinds exact, values
    <= 1 ulp, for sequence lengths with and without an n % 8 tail, fresh (Python
    float 1.0 -> float64 path) and fully-updated (float32 path) sequences."""
    E = 4
    per_env = T + P + n + 3 * (T - T // 2) + 8
    m = _run_pair(40 + T, E, [("feed", per_env), ("draw", 31, 0.0), ("draw", 32, 0.1), ("feed", T // 2 + 3),
                              ("draw", 33, 0.4), ("draw", 34, 0.6), ("feed", T), ("draw", 35, 0.9)],
                  dict(size=E * (per_env + 2 * T), train_frequency=0, nstep_target=n, nstep_train=T, prefix_steps=P,
                       alpha=0.9, beta=0.6, max_weight_factor=0.9),
                  0.99, True, dict(frame_shape=(1, 8, 8), lstm_units=4, n_actions=4, done_prob=0.03), 4)
    assert m["draws"] == 5


@pytest.mark.parametrize("T,P,n,vf", [(1, 0, 3, None), (8, 4, 2, 1e-3), (1, 0, 1, 1e-2), (16, 0, 5, None)])
def test_acting_time_priority_initialisation_vs_oracle(T, P, n, vf):
    """acting_priority_init (an extension: the reference lists it as missing,
    prioritized_replay_history.py:33-36) against its restatement on the reference's
    record structures (oracle.replay.OracleActingPriorityReplay): after every vector
    step the tree holds the same leaf kinds, leaf values within 1e-6 (one float32
    TD per transition: sqrt / pow roundings), the same free list; and no leaf input
    is left at the constant 1.0 once its n-step target exists."""
    from oracle import replay as orc
    from rltime_amd.history import PrioritizedReplayHistoryBuffer
    from tests.golden.streams import scalar_kind
    E, gamma = 6, 0.97
    spec = StreamSpec(seed=70 + T + n, num_envs=E, frame_shape=(1, 8, 8), lstm_units=4 if T > 1 else 0,
                      n_actions=5, done_prob=0.08)
    hist = dict(size=E * 40, train_frequency=0, nstep_target=n, nstep_train=T, prefix_steps=P,
                alpha=0.7, beta=0.5, max_weight_factor=0.9)
    ora = orc.OracleActingPriorityReplay(gamma=gamma, acting_priority_vf_eps=vf, **hist,
                                         discount_function=orc.make_discount(gamma))
    dev = PrioritizedReplayHistoryBuffer(**hist, gamma=gamma, acting_priority_init=True, acting_priority_vf_eps=vf)
    worst = 0.0
    for i, st in enumerate(vector_steps(spec, 90)):
        ora.update(as_reference_samples(spec, st))
        dev.update(as_reference_samples(spec, st))
        if i % 7 and i < 80:
            continue
        v, k, _ = dev.tree_nodes()
        cap = ora.tree.capacity
        want = np.array([float(x) for x in ora.tree.nodes[cap:]])
        wk = np.array([scalar_kind(x) for x in ora.tree.nodes[cap:]], dtype=np.uint8)
        assert np.array_equal(k[cap:], wk), i
        np.testing.assert_allclose(v[cap:], want, rtol=1e-6, atol=0, err_msg="step %d" % i)
        live = want > 0
        if live.any():
            worst = max(worst, float(np.max(np.abs(v[cap:][live] - want[live]) / want[live])))
        assert np.array_equal(dev.free_slots(), np.array(list(ora.free_slots)))
    # sampling + a learner-side update_losses on top still agree (stratified indices included)
    random.seed(5)
    a = ora.get_train_data(4, 0.3)
    random.seed(5)
    b = dev.get_train_data(4, 0.3)
    assert np.array_equal(scenario.to_numpy(b["extra_data"]["loss_indices"]), a["extra_data"]["loss_indices"])
    print("acting-time priorities: max rel deviation of leaf values %.2e" % worst)
    dev.close()


def test_frame_dedup_uniform_atari_frames_vs_oracle():
    """frame_stack_dedup: the shard keeps ONE 84x84 plane per transition; the oracle
    (reference algorithm) is fed and stores the whole (4,84,84) stacks of a
    stack-consistent stream (resets included).  Every gathered batch must be
    bit-identical, through ring wrap and eviction (size 1024 over 8 envs)."""
    _run_pair(51, 8, [("feed", 40), ("draw", 1, None), ("feed", 200), ("draw", 2, None),
                      ("feed", 77), ("draw", 3, None), ("feed", 130), ("draw", 4, None)],
              dict(size=1024, train_frequency=0, nstep_target=1, nstep_train=1, prefix_steps=0),
              0.99, False, dict(frame_shape=(4, 84, 84), n_actions=6, done_prob=0.02, stacked=True), 64,
              dev_kw=dict(frame_stack_dedup=True))


def test_frame_dedup_resets_within_an_envs_first_steps():
    """Episode ends inside an env's first three transitions: their stacks reach back
    to planes no transition stored (virtual predecessors in ring slots -1, -2, -3);
    a reset's zero fill must not overwrite them."""
    _run_pair(53, 12, [("feed", 6), ("draw", 1, None), ("feed", 9), ("draw", 2, None), ("feed", 40), ("draw", 3, None)],
              dict(size=600, train_frequency=0, nstep_target=2, nstep_train=2, prefix_steps=0),
              0.99, False, dict(frame_shape=(4, 8, 8), n_actions=4, done_prob=0.35, stacked=True), 24,
              dev_kw=dict(frame_stack_dedup=True))


@pytest.mark.parametrize("noxing", [False, True])
def test_frame_dedup_per_sequences_vs_oracle(noxing):
    """De-duplicated storage under prioritized sequence replay: overlapped
    (states / target_states views of one block) and separate (avoid_episode_crossing)
    layouts, recurrent state, eviction; frames, n-step scalars, indices, priorities
    all as with whole-stack storage."""
    m = _run_pair(52, 6, [("feed", 60), ("draw", 11, 0.0), ("feed", 30), ("draw", 12, 0.2),
                          ("draw", 13, 0.3), ("feed", 120), ("draw", 14, 0.5), ("feed", 45),
                          ("draw", 15, 0.8), ("draw", 16, 1.0)],
                  dict(size=900, train_frequency=4, nstep_target=2, nstep_train=16, prefix_steps=8,
                       alpha=0.9, beta=0.6, max_weight_factor=0.9, avoid_episode_crossing=noxing),
                  0.997, True, dict(frame_shape=(4, 8, 8), lstm_units=32, n_actions=6, done_prob=0.05, stacked=True), 16,
                  dev_kw=dict(frame_stack_dedup=True))
    assert m["draws"] == 6


def test_frame_dedup_rejects_frames_that_break_the_stack_contract():
    from rltime_amd import _lib
    from rltime_amd.history import ReplayHistoryBuffer
    spec = StreamSpec(seed=9, num_envs=4, frame_shape=(4, 8, 8), n_actions=4, done_prob=0.0, stacked=False)
    buf = ReplayHistoryBuffer(size=64, train_frequency=0, nstep_target=1, nstep_train=1, gamma=0.9, frame_stack_dedup=True)
    for st in vector_steps(spec, 6):                    # unrelated frames every step: not a shifted stack
        buf.update(as_reference_samples(spec, st))
    np.random.seed(0)
    torch.cuda.synchronize()                              # the flag is written by the (asynchronous) ingest kernels
    with pytest.raises(_lib.MirlError, match="shift contract"):
        buf.get_train_data(2)
    buf.close()
    with pytest.raises(_lib.MirlError, match="multiple of 16"):
        b2 = ReplayHistoryBuffer(size=64, train_frequency=0, nstep_target=1, nstep_train=1, gamma=0.9, frame_stack_dedup=True)
        s2 = StreamSpec(seed=9, num_envs=2, frame_shape=(4, 3, 3), stacked=True)
        b2.update(as_reference_samples(s2, next(vector_steps(s2, 1))))


def _philox_u53(seed, call, lane):
    """Philox4x32-10 (Salmon et al., SC'11) keyed (seed lo, seed hi), counter
    (lane, call lo, call hi, tag) -> 53-bit uniform: an independent restatement of
    what k_per_sample draws in device-RNG mode (csrc/replay.hip)."""
    M0, M1, MASK = 0xD2511F53, 0xCD9E8D57, 0xFFFFFFFF
    c = [lane & MASK, call & MASK, (call >> 32) & MASK, 0x52544D45]
    k0, k1 = seed & MASK, (seed >> 32) & MASK
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0, k1 = (k0 + 0x9E3779B9) & MASK, (k1 + 0xBB67AE85) & MASK
    return ((c[0] >> 5) * 67108864.0 + (c[1] >> 6)) / 9007199254740992.0


def test_device_rng_path_is_the_reference_sampler_on_philox_uniforms(monkeypatch):
    """bench.py samples with device_rng=True (Philox drawn inside k_per_sample), every
    other parity test with the reference's host streams.  Pin the device-RNG path too:
    recompute the Philox uniforms on the host, hand them to the ORACLE's stratified
    sampler in place of random.random(), and demand the same tree indices, windows,
    importance weights and gathered batch."""
    from oracle import replay as orc
    from rltime_amd.history import PrioritizedReplayHistoryBuffer
    E, B, gamma = 6, 16, 0.997
    spec = StreamSpec(seed=61, num_envs=E, frame_shape=(1, 8, 8), lstm_units=4, n_actions=4, done_prob=0.04)
    hist = dict(size=900, train_frequency=4, nstep_target=2, nstep_train=16, prefix_steps=8,
                alpha=0.9, beta=0.6, max_weight_factor=0.9)
    ora = orc.OraclePrioritizedReplay(**hist, discount_function=orc.make_discount(gamma))
    dev = PrioritizedReplayHistoryBuffer(**hist, gamma=gamma, device_rng=True)
    step_no, calls, draws = 0, 0, 0
    for feed, progress in ((70, 0.0), (25, 0.3), (0, 0.4), (90, 0.7), (40, 1.0)):
        for st in vector_steps(spec, feed, start_step=step_no):
            ora.update(as_reference_samples(spec, st))
            dev.update(as_reference_samples(spec, st))
        step_no += feed
        b = dev.get_train_data(B, train_progress=progress)
        calls += 1
        uniforms = iter([_philox_u53(dev._seed, calls, i) for i in range(B)])
        monkeypatch.setattr(random, "random", lambda: next(uniforms))
        a = ora.get_train_data(B, train_progress=progress)
        monkeypatch.undo()
        assert (a is None) == (b is None)
        if a is None:
            continue
        draws += 1
        assert np.array_equal(dev.last_sample["slot"].cpu().numpy(), np.array(ora.last_slots))
        fa = scenario.flatten("", a, {})
        fb = {k: scenario.to_numpy(v) for k, v in scenario.flatten("", b, {}).items()}
        for key, w in fa.items():
            if key.endswith("importance_weights"):
                np.testing.assert_allclose(fb[key], w.astype(np.float32), rtol=scenario.WEIGHT_RTOL, err_msg=key)
            elif key.endswith("actions") or key.endswith("loss_indices"):
                assert np.array_equal(fb[key], w), key
            else:
                assert np.array_equal(fb[key], scenario.make_tensor_dtype(w)), key
        P = hist["prefix_steps"]
        idx = a["extra_data"]["loss_indices"][P:].reshape(-1, 2)
        losses = (np.random.RandomState(draws).randn(idx.shape[0]) * 0.7).astype(np.float32)
        ora.update_losses(idx, losses)
        dev.update_losses(idx, losses)
    assert draws >= 4
    dev.close()


def test_per_t1_vs_oracle_rainbow_shape():
    """Rainbow-shaped: T=1, n=3, f32 tree regime, beta anneal, many updates."""
    script = [("feed", 50)]
    for i in range(12):
        script += [("draw", 100 + i, i / 12.0), ("feed", 9)]
    _run_pair(23, 16, script,
              dict(size=2048, train_frequency=4, nstep_target=3, nstep_train=1, prefix_steps=0,
                   alpha=0.6, beta=0.4, beta_anneal=True),
              0.99, True, dict(frame_shape=(4, 21, 21), n_actions=6, done_prob=0.02), 128)


def test_duplicate_loss_rows_last_wins():
    """update_losses walks its rows in order (prioritized_replay_history.py:252):
    a transition reported twice keeps the LAST loss."""
    from oracle import replay as orc
    from rltime_amd.history import PrioritizedReplayHistoryBuffer
    spec = StreamSpec(seed=5, num_envs=2, frame_shape=(1, 4, 4), done_prob=0.0)
    hist = dict(size=64, train_frequency=4, nstep_target=1, nstep_train=4, prefix_steps=0, alpha=0.7)
    ora = orc.OraclePrioritizedReplay(**hist, discount_function=orc.make_discount(0.9))
    dev = PrioritizedReplayHistoryBuffer(**hist, gamma=0.9)
    for st in vector_steps(spec, 20):
        ora.update(as_reference_samples(spec, st)); dev.update(as_reference_samples(spec, st))
    idx = np.array([[0, 4], [0, 5], [0, 4], [1, 8], [0, 5], [0, 4], [-1, -1], [1, 9]], dtype=np.int64)
    losses = np.array([0.5, -0.25, 2.0, 0.1, 3.5, -0.75, 9.0, 0.3], dtype=np.float32)
    keep = idx[:, 0] >= 0
    ora.update_losses(idx[keep], losses[keep])
    dev.update_losses(idx, losses)
    out = np.zeros(8, dtype=np.float32)
    from rltime_amd._lib import lib, check, np_ptr
    check(lib.mirl_replay_losses_peek(dev._h, 0, 2, 8, np_ptr(out)))
    want = [ora.rings[0][o]['loss'] for o in range(2, 10)]
    want = np.array([-1.0 if isinstance(w, float) else w for w in want], dtype=np.float32)
    assert np.array_equal(out, want)
    dev.close()


def test_no_gpu_fallback_marker():
    """The product never imports the oracle."""
    import rltime_amd.history.replay_history as m
    src = open(m.__file__).read()
    assert "oracle" not in src


def test_full_size_properties_config_d():
    """BASELINE configs[3] at full size (1M transitions, B=512, T=80, burn-in 40,
    n=2, 84x84x4): size-independent properties instead of an oracle run —
      * every gathered frame row carries the identity of (env, start+t-1);
      * stratified sum-tree indices are non-decreasing, land on active slots
        and their windows lie inside the live ring;
      * importance weights are in (0, 1] with max exactly 1;
      * the root equals the sum of the leaves (1e-9) and update_losses is
        idempotent (same losses twice -> identical tree)."""
    from rltime_amd.history import PrioritizedReplayHistoryBuffer
    E, T, P, n, B, H = 256, 80, 40, 2, 512, 512
    buf = PrioritizedReplayHistoryBuffer(
        size=1000000, train_frequency=4, nstep_target=n, nstep_train=T, prefix_steps=P,
        alpha=0.9, beta=0.6, gamma=0.997, device_rng=True, keep_policy_outputs=False)
    ex = {"x": np.zeros((4, 84, 84), np.uint8),
          "layer1_state": {"hx": np.zeros(H, np.float32), "cx": np.zeros(H, np.float32), "initials": np.float32(0)}}
    buf.configure(ex, num_envs=E)
    dev = buf.device
    steps = 1000000 // E + 300                   # wraps the rings, evicts ~76k transitions
    env_ids = torch.arange(E, device=dev, dtype=torch.int32)
    state = torch.zeros(E, 2 * H, device=dev)
    zeros_f = torch.zeros(E, device=dev)
    act = torch.zeros(E, dtype=torch.int32, device=dev)
    done = torch.zeros(E, dtype=torch.uint8, device=dev)
    frames = torch.zeros((E, 4 * 84 * 84), dtype=torch.uint8, device=dev)
    for s in range(steps):
        # identity in the first 8 bytes: env (2 bytes), offset (4 bytes), checksum-ish
        frames[:, 0] = (env_ids & 0xFF).to(torch.uint8); frames[:, 1] = (env_ids >> 8).to(torch.uint8)
        for k in range(4):
            frames[:, 2 + k] = (s >> (8 * k)) & 0xFF
        buf.update_batch(frames.view(E, 4, 84, 84), act, zeros_f + float(s % 3) - 1.0, done, state=state, initials=zeros_f)
    st = buf.stats()
    assert st["total_items"] == 1000000
    buf.train_quota = 0          # the fill accrued size*train_frequency of quota (replay_history.py:179 guard)
    batch = buf.get_train_data(B, 0.5)
    smp = buf.last_sample
    slot = smp["slot"].cpu().numpy(); env = smp["env"].cpu().numpy(); start = smp["start"].cpu().numpy()
    assert np.all(np.diff(slot) >= 0)                                   # stratified => sorted
    se, sb = buf.slot_table()
    assert np.all(se[slot] == env) and np.all(sb[slot] - P == start)    # active slots, right windows
    first, count = buf.env_meta()
    assert np.all(start >= first[env]) and np.all(start + T + P + n - 1 <= count[env])
    x = batch["states"]["x"]                                            # (L, B, 4, 84, 84)
    tx = batch["target_states"]["x"]
    L = T + P
    assert x.shape == (L, B, 4, 84, 84) and tx.data_ptr() == x.data_ptr() + n * B * 4 * 84 * 84
    head = torch.cat([x, tx[-n:]])[:, :, 0].reshape(L + n, B, -1)[:, :, :6].cpu().numpy().astype(np.int64)
    got_env = head[..., 0] + (head[..., 1] << 8)
    got_off = head[..., 2] + (head[..., 3] << 8) + (head[..., 4] << 16) + (head[..., 5] << 24)
    want_off = start[None, :] + np.arange(L + n)[:, None] - 1          # state of transition o is next_state of o-1
    assert np.array_equal(got_env, np.broadcast_to(env[None, :], got_env.shape))
    assert np.array_equal(got_off, want_off)
    w = batch["extra_data"]["importance_weights"].cpu().numpy()
    assert w.max() == 1.0 and w.min() > 0.0 and np.all(w[0] == w[-1])
    li = batch["extra_data"]["loss_indices"].cpu().numpy()
    assert np.all(li[:P] == -1) and np.array_equal(li[P:, :, 1], start[None, :] + P + np.arange(T)[:, None])
    # tree invariants + idempotent update
    idx = batch["extra_data"]["loss_indices"][P:].reshape(-1, 2)
    losses = torch.rand(idx.shape[0], device=dev)
    buf.update_losses(idx, losses)
    v1, k1, _ = buf.tree_nodes()
    cap = len(v1) // 2
    assert abs(v1[1] - v1[cap:].sum()) <= 2e-6 * v1[1]   # f32-kind nodes round like np.float32 adds
    buf.update_losses(idx, losses)
    v2, k2, _ = buf.tree_nodes()
    assert np.array_equal(v1, v2) and np.array_equal(k1, k2)
    assert st["active_sequences"] == np.count_nonzero(v1[cap:])
    buf.close()


def test_ragged_vector_steps_vs_oracle():
    """Calls that carry only a subset of the envs, in shuffled env order (each env
    at most once per call): per-env rings advance unevenly, the global FIFO
    eviction and the PER activation order must still follow the reference."""
    from oracle import replay as orc
    from rltime_amd.history import PrioritizedReplayHistoryBuffer
    spec = StreamSpec(seed=51, num_envs=5, frame_shape=(2, 5, 7), lstm_units=4, n_actions=3, done_prob=0.1)
    hist = dict(size=120, train_frequency=0, nstep_target=2, nstep_train=4, prefix_steps=2, alpha=0.7, beta=0.4)
    ora = orc.OraclePrioritizedReplay(**hist, discount_function=orc.make_discount(0.95))
    dev = PrioritizedReplayHistoryBuffer(**hist, gamma=0.95, num_envs=5, env_base=0, env_ring_slack=80)
    rng = np.random.RandomState(9)
    per_env_step = [0] * 5
    for call in range(140):
        envs = [e for e in rng.permutation(5) if rng.rand() < 0.7] or [int(rng.randint(5))]
        chunk = []
        for e in envs:
            st = next(vector_steps(spec, 1, start_step=per_env_step[e]))
            chunk.append(as_reference_samples(spec, st)[e])
            per_env_step[e] += 1
        ora.update([dict(s) for s in chunk])
        dev.update([dict(s) for s in chunk])
        if call % 20 == 19:
            random.seed(call)
            a = ora.get_train_data(4, 0.3)
            random.seed(call)
            b = dev.get_train_data(4, 0.3)
            assert (a is None) == (b is None)
            if a is not None:
                fb = {k: scenario.to_numpy(v) for k, v in scenario.flatten("", b, {}).items()}
                for k, w in scenario.flatten("", a, {}).items():
                    if k.endswith("importance_weights"):
                        np.testing.assert_allclose(fb[k], w.astype(np.float32), rtol=scenario.WEIGHT_RTOL)
                    else:
                        assert np.array_equal(fb[k], scenario.make_tensor_dtype(w) if not k.endswith(("actions", "loss_indices")) else w), k
            assert np.array_equal(dev.free_slots(), np.array(list(ora.free_slots)))
    dev.close()


def test_large_batch_strata_on_device():
    """B = 3000 > the 1024 lanes of the sampling workgroup: every stratum still
    gets its own query (lanes loop), indices stay sorted, weights max is 1."""
    from rltime_amd.history import PrioritizedReplayHistoryBuffer
    buf = PrioritizedReplayHistoryBuffer(size=40000, train_frequency=0, nstep_target=1, nstep_train=1,
                                         gamma=0.99, device_rng=True, keep_policy_outputs=False)
    E = 50
    buf.configure({"x": np.zeros((32,), np.uint8)}, num_envs=E)
    dev = buf.device
    fr = torch.zeros((E, 32), dtype=torch.uint8, device=dev)
    z = torch.zeros(E, device=dev)
    for _ in range(900):
        buf.update_batch(fr, z.int(), z, z.to(torch.uint8))
    B = 3000
    batch = buf.get_train_data(B, 0.5)
    idx = batch["extra_data"]["loss_indices"].reshape(-1, 2)
    buf.update_losses(idx, torch.rand(B, device=dev) * 3)
    batch = buf.get_train_data(B, 0.5)
    slot = buf.last_sample["slot"].cpu().numpy()
    assert np.all(np.diff(slot) >= 0) and len(np.unique(slot)) > B // 2
    w = batch["extra_data"]["importance_weights"].cpu().numpy()
    assert w.max() == 1.0 and w.min() > 0
    buf.close()


@pytest.mark.parametrize("mode", ["dqn_uniform_config_b", "rainbow_per_config_c"])
def test_full_size_properties_t1_configs(mode):
    """BASELINE configs[1] (DQN, uniform, B=256) and configs[2] (Rainbow-style,
    PER with a 2^20-leaf tree, n=3, B=512) at the full 1M-transition size."""
    from rltime_amd.history import ReplayHistoryBuffer, PrioritizedReplayHistoryBuffer
    E = 32
    per = mode.startswith("rainbow")
    n = 3 if per else 1
    B = 512 if per else 256
    kw = dict(size=1000000, train_frequency=8, nstep_target=n, nstep_train=1, gamma=0.99,
              device_rng=True, keep_policy_outputs=False)
    buf = PrioritizedReplayHistoryBuffer(alpha=0.6, beta=0.4, beta_anneal=True, **kw) if per \
        else ReplayHistoryBuffer(**kw)
    buf.configure({"x": np.zeros((4, 84, 84), np.uint8)}, num_envs=E)
    dev = buf.device
    env_ids = torch.arange(E, device=dev, dtype=torch.int32)
    frames = torch.zeros((E, 4 * 84 * 84), dtype=torch.uint8, device=dev)
    act = torch.zeros(E, dtype=torch.int32, device=dev)
    done = torch.zeros(E, dtype=torch.uint8, device=dev)
    rew = torch.ones(E, device=dev)
    steps = 1000000 // E + 50
    for s in range(steps):
        frames[:, 0] = env_ids.to(torch.uint8)
        for k in range(4):
            frames[:, 1 + k] = (s >> (8 * k)) & 0xFF
        buf.update_batch(frames.view(E, 4, 84, 84), act, rew, done)
    assert buf.stats()["total_items"] == 1000000
    buf.train_quota = 0
    if per:
        assert buf.stats()["tree_capacity"] == 1 << 20
    batch = buf.get_train_data(B, 0.25)
    env = buf.last_sample["env"].cpu().numpy(); start = buf.last_sample["start"].cpu().numpy()
    x = batch["states"]["x"].reshape(1, B, -1)[0, :, :5].cpu().numpy().astype(np.int64)
    tx = batch["target_states"]["x"].reshape(1, B, -1)[0, :, :5].cpu().numpy().astype(np.int64)
    off = lambda h: h[:, 1] + (h[:, 2] << 8) + (h[:, 3] << 16) + (h[:, 4] << 24)   # noqa: E731
    assert np.array_equal(x[:, 0], env) and np.array_equal(off(x), start - 1)           # state = next_state of o-1
    assert np.array_equal(tx[:, 0], env) and np.array_equal(off(tx), start + n - 1)     # n-step target state
    ret = batch["returns"].cpu().numpy()[0]
    want = sum(0.99 ** k for k in range(n))
    np.testing.assert_allclose(ret, np.float32(want), rtol=1e-7)
    assert np.all(batch["nsteps"].cpu().numpy() == n) and np.all(batch["target_masks"].cpu().numpy() == 1)
    if per:
        slot = buf.last_sample["slot"].cpu().numpy()
        assert np.all(np.diff(slot) >= 0)
        idx = batch["extra_data"]["loss_indices"].reshape(-1, 2)
        buf.update_losses(idx, torch.randn(B, device=dev))
        v, k, _ = buf.tree_nodes()
        cap = len(v) // 2
        assert abs(v[1] - v[cap:].sum()) <= 2e-6 * v[1]
        assert np.all(k[cap:][np.unique(slot)] == 1)          # updated leaves are np.float32-kind
    buf.close()


def test_error_behaviour_mirrors_reference_asserts():
    """The reference aborts with asserts; the C-ABI returns an error code and a
    message (never aborts), surfaced as MirlError by the mirror."""
    from rltime_amd._lib import MirlError
    from rltime_amd.history import ReplayHistoryBuffer, PrioritizedReplayHistoryBuffer
    # prioritized_replay_history.py:102  "Overlap must be < nstep_train"
    with pytest.raises(AssertionError):
        PrioritizedReplayHistoryBuffer(size=64, train_frequency=4, nstep_target=1, nstep_train=4, overlap=4, gamma=0.9)
    # history.py:43-44: n-step > 1 needs a discount
    with pytest.raises((AssertionError, ValueError)):
        ReplayHistoryBuffer(size=64, train_frequency=4, nstep_target=3, nstep_train=1)
    # replay_history.py:113: buffer full but no batch can ever be formed
    spec = StreamSpec(seed=1, num_envs=4, frame_shape=(1, 4, 4), done_prob=0.0)
    buf = ReplayHistoryBuffer(size=16, train_frequency=0, nstep_target=2, nstep_train=4, gamma=0.9)
    for st in vector_steps(spec, 6):
        buf.update(as_reference_samples(spec, st))
    np.random.seed(0)
    with pytest.raises(MirlError, match="replay_history.py:113"):
        buf.get_train_data(2)
    buf.close()
    # replay_history.py:179-181: train/act ratio drift
    buf = ReplayHistoryBuffer(size=64, train_frequency=4, nstep_target=1, nstep_train=1, gamma=0.9)
    for st in vector_steps(spec, 8):
        buf.update(as_reference_samples(spec, st))
    np.random.seed(0)
    with pytest.raises(MirlError, match="179-181"):
        for _ in range(200):
            buf.get_train_data(4)
    buf.close()
    # an env id outside the shard
    buf = ReplayHistoryBuffer(size=64, train_frequency=0, nstep_target=1, nstep_train=1, gamma=0.9, num_envs=4, env_base=0)
    bad = as_reference_samples(spec, next(vector_steps(spec, 1)))
    bad[2]["env_id"] = 9
    with pytest.raises(MirlError, match="outside this shard"):
        buf.update(bad)
    buf.close()
    # get_train_data before anything was fed -> None (feed more)
    buf = PrioritizedReplayHistoryBuffer(size=64, train_frequency=4, nstep_target=1, nstep_train=1, gamma=0.9)
    assert buf.get_train_data(4, 0.0) is None
    assert buf.needed_feed_count(4, 8) == 8


def test_snapshot_resume_round_trip(tmp_path):
    """save -> fresh buffer -> load -> continue: the resumed shard produces the
    same batches, priorities, free list and tree as the uninterrupted one, and a
    snapshot of a different configuration is refused."""
    from rltime_amd._lib import MirlError
    from rltime_amd.history import PrioritizedReplayHistoryBuffer
    spec = StreamSpec(seed=61, num_envs=5, frame_shape=(4, 12, 12), lstm_units=8, n_actions=4, done_prob=0.06)
    hist = dict(size=260, train_frequency=0, nstep_target=2, nstep_train=6, prefix_steps=3, alpha=0.8, beta=0.5,
                global_importance_scaling=True)
    a = PrioritizedReplayHistoryBuffer(**hist, gamma=0.99)
    steps = list(vector_steps(spec, 120))

    def advance(buf, lo, hi, seed0):
        outs = []
        for i in range(lo, hi):
            buf.update(as_reference_samples(spec, steps[i]))
            if i % 10 == 9:
                random.seed(seed0 + i)
                b = buf.get_train_data(6, 0.4)
                if b is None:
                    continue
                idx = b["extra_data"]["loss_indices"][3:].reshape(-1, 2)
                g = torch.Generator().manual_seed(i)
                buf.update_losses(idx, torch.rand(idx.shape[0], generator=g).cuda())
                outs.append({k: scenario.to_numpy(v) for k, v in scenario.flatten("", b, {}).items()})
        return outs

    advance(a, 0, 70, 1000)
    path = tmp_path / "shard.snap"
    a.save(path)
    b = PrioritizedReplayHistoryBuffer(**hist, gamma=0.99)
    example = as_reference_samples(spec, steps[0])[0]["next_state"]
    b.load(path, example_state=example, num_envs=5, policy_f32=spec.n_actions)
    oa = advance(a, 70, 120, 2000)
    ob = advance(b, 70, 120, 2000)
    assert len(oa) == len(ob) and len(oa) >= 4
    for x, y in zip(oa, ob):
        assert set(x) == set(y)
        for k in x:
            assert np.array_equal(x[k], y[k]), k
    va, ka, ma = a.tree_nodes(); vb, kb, mb = b.tree_nodes()
    assert np.array_equal(va, vb) and np.array_equal(ka, kb) and np.array_equal(ma, mb)
    assert np.array_equal(a.free_slots(), b.free_slots()) and a.stats() == b.stats()
    c = PrioritizedReplayHistoryBuffer(**dict(hist, nstep_train=4), gamma=0.99)
    with pytest.raises(MirlError, match="does not match"):
        c.load(path, example_state=example, num_envs=5, policy_f32=spec.n_actions)
    a.close(); b.close(); c.close()


def test_two_recurrent_layers_vs_oracle():
    """Input-state pytrees with two recurrent layers of different widths (e.g. a
    CNN -> LSTM -> FC -> LSTM model): each layer's hx/cx/initials come back in
    their own sub-dict, bit-exact."""
    from oracle import replay as orc
    from rltime_amd.history import ReplayHistoryBuffer
    E, T, P, n = 3, 5, 2, 2
    hist = dict(size=150, train_frequency=0, nstep_target=n, nstep_train=T, prefix_steps=P)
    ora = orc.OracleReplay(**hist, discount_function=orc.make_discount(0.97))
    dev = ReplayHistoryBuffer(**hist, gamma=0.97)
    rng = np.random.RandomState(4)
    for s in range(70):
        dones = rng.rand(E) < 0.1
        samples = []
        for e in range(E):
            ini = np.float32(dones[e])
            samples.append({
                "policy_output": {"actions": int(rng.randint(3)), "qvalues": rng.randn(3).astype(np.float32)},
                "next_state": {"x": rng.randint(0, 256, (2, 6, 6)).astype(np.uint8), "layer0_state": {},
                               "layer1_state": {"hx": rng.randn(4).astype(np.float32), "cx": rng.randn(4).astype(np.float32), "initials": ini},
                               "layer2_state": {},
                               "layer3_state": {"hx": rng.randn(7).astype(np.float32), "cx": rng.randn(7).astype(np.float32), "initials": ini}},
                "reward": float(rng.randint(-1, 2)), "done": bool(dones[e]), "info": {}, "env_id": e})
        ora.update([dict(x) for x in samples])
        dev.update([dict(x) for x in samples])
    np.random.seed(8)
    a = ora.get_train_data(6)
    np.random.seed(8)
    b = dev.get_train_data(6)
    fb = {k: scenario.to_numpy(v) for k, v in scenario.flatten("", b, {}).items()}
    fa = scenario.flatten("", a, {})
    assert set(fa) == set(fb)
    for k, w in fa.items():
        want = w if k.endswith("actions") else scenario.make_tensor_dtype(w)
        assert np.array_equal(fb[k], want), k
    assert fb["states.layer3_state.hx"].shape == (T + P, 6, 7)
    dev.close()
