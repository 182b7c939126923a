"""GPU: one learner step replayed from a captured HIP graph (training/torch_trainer.py _learner_step_graphed) against the
same step issued launch by launch — THE LOOP of rltime/training/multi_step_trainer.py:245-375 at T = 1 (BASELINE
configs[1] DQN + uniform replay, configs[2] Rainbow-style IQN + prioritized replay)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CNN = {"type": "cnn", "args": {"channels_last": True, "layers": [{"filters": 32, "kernel": 8, "stride": 4}, {"filters": 64, "kernel": 4, "stride": 2},
                                                                 {"filters": 64, "kernel": 3, "stride": 1}]}}
# epsilon = 1 throughout: the actions are the Philox draws of (seed, step, env), independent of the weights — both runs then
# fill the replay with the same transitions, and what is compared is the learner (two EAGER runs of this loop already differ
# in the last bit of a loss after a few steps — atomics in the library's weight-gradient kernels — and a greedy actor turns one
# flipped arg-max into a different data stream)
BASE = {
    "acting": {"actor_envs": 8, "exploration": {"type": "epsilon_greedy", "args": {"eps_start": 1.0, "eps_final": 1.0, "exploration_fraction": 0.5}}},
    "env": "synthetic-atari", "env_args": {"frame_shape": [4, 84, 84], "n_actions": 6, "done_prob": 0.02},
}


def _series(kind, graphed, steps=600):
    from rltime_amd.general.loggers import NullLogger
    from rltime_amd.train import train
    cfg = copy.deepcopy(BASE)
    cfg["model"] = {"type": "sequential", "args": {"layer_configs": [CNN, {"type": "fc", "args": {"fc_size": 64}}]}}
    common = {"clip_rewards": True, "gamma": 0.99, "mbatch_size": 32, "nstep_train": 1, "nstep_target": 3, "lr": 1e-3, "lr_anneal": True,
              "double_q": True, "clip_grad": 10.0, "target_update_freq": 24, "total_steps": steps, "log_freq": 10 ** 9, "warmup_steps": 96,
              "graph_learner_step": graphed}
    if kind == "iqn_lstm":
        # the RECURRENT step (BASELINE configs[3] in small): burn-in prefix passes with the stored-state substitution,
        # rnn_bootstrap target pass, persistent LSTM forward sweeps (H = 128, B = 16), per-step backward, sequence priorities
        cfg["model"]["args"]["layer_configs"] = [CNN, {"type": "lstm", "args": {"num_units": 128}}, {"type": "fc", "args": {"fc_size": 64}}]
        cfg["acting"]["actor_envs"] = 16
        cfg["policy_args"] = {"dueling": True, "embedding_dim": 16, "num_sampling_quantiles": 8}
        cfg["training"] = {"type": "iqn", "args": dict(
            common, mbatch_size=16, nstep_train=8, burn_in_timesteps=4, nstep_target=2, rnn_bootstrap=True, vf_scale_epsilon=1e-3,
            clip_rewards=False, warmup_steps=480, total_steps=max(steps, 1600),
            history_mode={"type": "prioritized_replay", "args": {
                "size": 1600, "train_frequency": 4, "alpha": 0.9, "beta": 0.6, "max_weight_factor": 0.9, "device_rng": True}})}
    elif kind == "dqn":
        cfg["policy_args"] = {"dueling": True}
        cfg["training"] = {"type": "dqn", "args": dict(common, history_mode={"type": "replay", "args": {
            "size": 400, "train_frequency": 8, "device_rng": True}})}
    else:
        cfg["policy_args"] = {"dueling": True, "embedding_dim": 16, "num_sampling_quantiles": 8}
        cfg["training"] = {"type": "iqn", "args": dict(common, history_mode={"type": "prioritized_replay", "args": {
            "size": 400, "train_frequency": 8, "alpha": 0.6, "beta": 0.4, "beta_anneal": True, "device_rng": True}})}
    torch.manual_seed(11)
    np.random.seed(11)
    import random
    random.seed(11)
    from rltime_amd.general.type_registry import get_registered_type
    from rltime_amd.train import create_actors
    actors = create_actors(cfg, torch.device("cuda", 0), device_acting=True, use_graph=True)
    cls = get_registered_type("trainers", cfg["training"]["type"])
    tr = cls(logger=NullLogger(), actors=actors, model_config=cfg["model"], policy_args=cfg.get("policy_args", {}))
    tr.data_parallel = None
    series = {"qloss": [], "grad_norm": []}
    orig = tr.value_log.log

    def tap(key, value, *a, **k):
        if key in series and k.get("group") == "train":
            series[key].append(value.detach().clone() if isinstance(value, torch.Tensor) else torch.tensor(float(value)))
        return orig(key, value, *a, **k)
    tr.value_log.log = tap
    tr.train(**cfg["training"]["args"])
    torch.cuda.synchronize()
    out = {k: torch.stack([t.float().cpu() for t in v]).numpy() for k, v in series.items()}
    out["params"] = [p.detach().cpu().clone() for p in tr.policy.parameters()]
    out["target"] = [p.detach().cpu().clone() for p in tr.target_policy.parameters()]
    out["captured"] = tr._gstep is not None and tr._gstep["graph"] is not None
    out["steps"] = tr.ts_learner_steps if hasattr(tr, "ts_learner_steps") else None
    tree = tr.history_buffer.tree_nodes() if kind in ("iqn", "iqn_lstm") else None
    out["tree"] = tree
    tr.history_buffer.close()
    return out


@pytest.fixture
def deterministic_library(monkeypatch):
    """MIOpen's weight-gradient kernels of these conv shapes add their K-splits with atomics: two EAGER runs of the loop then
    differ in the last bit of a gradient after a few steps, and Adam on this tiny net grows that by ~2.6x per step (1e-7 ->
    1e-2 in a dozen steps; tools/graph_step_deviation.py prints the spreads).  With this library's own weight-gradient
    kernels at every frame count (csrc/conv_wrw.hip: fixed summation order, like all of its kernels; by default they take
    over above a work threshold) every kernel of the step is deterministic, and what the tests below can state is EXACT
    equality."""
    from rltime_amd.models.torch import fused
    monkeypatch.setattr(fused, "_CONV_WRW_MIN_WORK", 0)


def _identical(a, b):
    for key in ("qloss", "grad_norm"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    assert all(torch.equal(x, y) for x, y in zip(a["params"], b["params"]))
    assert all(torch.equal(x, y) for x, y in zip(a["target"], b["target"]))


def test_dqn_learner_step_from_a_graph_is_the_eager_step(deterministic_library):
    """DQN + uniform replay, T = 1: every loss and gradient norm of the run and the final online / target weights are
    BIT-IDENTICAL to the same set-up issued launch by launch (same kernels, same order: the graph only removes the host's
    launch work).  Target syncs, the annealed learning rate and the actor's weight refresh (version counters) all cross the
    replays.  Against the step with the learning rate on the host (a double, not the float32 device word the captured
    update reads): the same first loss, the same first steps to rounding, then statistics."""
    a, b, c = _series("dqn", True), _series("dqn", "no-capture"), _series("dqn", False)
    assert a["captured"] and not b["captured"] and not c["captured"]
    assert len(a["qloss"]) == len(b["qloss"]) == len(c["qloss"]) > 100
    _identical(a, b)
    assert a["qloss"][0] == c["qloss"][0]
    for key in ("qloss", "grad_norm"):
        np.testing.assert_allclose(a[key][:5], c[key][:5], rtol=1e-4, err_msg=key)
        assert np.isfinite(a[key]).all()
        assert abs(a[key][-50:].mean() - c[key][-50:].mean()) <= 0.35 * abs(c[key][-50:].mean()), key
    assert any(not torch.equal(x, y) for x, y in zip(a["target"], a["params"]))          # the target net is not the online net


def test_iqn_prioritized_learner_step_from_a_graph_is_the_eager_step(deterministic_library):
    """Rainbow-style IQN on prioritized replay, T = 1: quantile fractions drawn by torch.rand INSIDE the captured step (the
    generator's offset advances per replay exactly as per eager call: the same fractions), priorities written back after
    every replay: losses, gradient norms, final weights AND the sum tree bit-identical to the eager run."""
    a, b = _series("iqn", True), _series("iqn", "no-capture")
    assert a["captured"] and not b["captured"]
    assert len(a["qloss"]) == len(b["qloss"]) > 100
    _identical(a, b)
    for x, y in zip(a["tree"], b["tree"]):
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))


def test_recurrent_learner_step_from_a_graph_is_the_eager_step(deterministic_library):
    """The recurrent IQN step — burn-in prefix passes of both networks with the stored recurrent state replaced at the
    first trained row (multi_step_trainer.py:90-131), rnn_bootstrap target pass, persistent LSTM sweeps, training pass,
    clip + Adam — captured once and replayed: every loss, gradient norm, the final weights and the priority tree
    bit-identical to the same set-up issued launch by launch."""
    a, b = _series("iqn_lstm", True), _series("iqn_lstm", "no-capture")
    assert a["captured"] and not b["captured"]
    assert len(a["qloss"]) == len(b["qloss"]) > 30
    _identical(a, b)
    for x, y in zip(a["tree"], b["tree"]):
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))


def test_two_launch_adam_and_torch_adam_walk_the_same_trajectory(deterministic_library, monkeypatch):
    """The graphed DQN step with csrc/optim.hip (norm -> clip -> Adam in two launches, bias corrections in float64) against
    the eager step with torch.optim.Adam as the reference uses it (bias corrections on the host in double) + the clip as
    tensor ops: the first steps to 1e-4 (what is left is the learning rate as a float32 device word against a double, and
    the net grows a last-bit difference 2.6x per step), then the statistics of the run.  (A `capturable` torch Adam is NOT the
    yardstick: it evaluates 1 - 0.999^step in float32, 1.3e-5 off at step 1.)"""
    a = _series("dqn", True)
    monkeypatch.setenv("MIRL_CLIP_ADAM", "0")
    b = _series("dqn", False)
    assert a["captured"] and not b["captured"]
    for key in ("qloss", "grad_norm"):
        np.testing.assert_allclose(a[key][:5], b[key][:5], rtol=1e-4, err_msg=key)
        assert abs(a[key][-50:].mean() - b[key][-50:].mean()) <= 0.35 * abs(b[key][-50:].mean()), key
