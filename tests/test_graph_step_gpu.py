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
    if kind == "dqn":
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
    tree = tr.history_buffer.tree_nodes() if kind == "iqn" else None
    out["tree"] = tree
    tr.history_buffer.close()
    return out


HEAD = 16        # learner steps compared tightly: 3 eager + capture + 12 replays, two target syncs (every 6 steps), the annealed lr


def _close_then_sane(a, b, tight, key, HEAD=HEAD):
    """Adam turns last-bit differences into O(lr) differences within tens of steps (two EAGER runs of this loop drift apart
    the same way), so: the first HEAD steps to `tight`, the rest of the run statistically."""
    np.testing.assert_allclose(a[key][:HEAD], b[key][:HEAD], rtol=tight, err_msg=key)
    assert np.isfinite(a[key]).all()
    assert abs(a[key][-50:].mean() - b[key][-50:].mean()) <= 0.35 * abs(b[key][-50:].mean()), key


def test_dqn_learner_step_from_a_graph_is_the_eager_step():
    """DQN + uniform replay, T = 1: losses and gradient norms against the same set-up issued launch by launch (same kernels,
    same order: the graph only removes the host's launch work).  Target syncs, the annealed learning rate and the actor's
    weight refresh (version counters) all cross the replays.  Against the DEFAULT eager step (Adam's bias corrections on the
    host in double instead of on the device in float32): the same first loss, then float32 rounding."""
    a, b, c = _series("dqn", True), _series("dqn", "no-capture"), _series("dqn", False)
    assert a["captured"] and not b["captured"] and not c["captured"]
    assert len(a["qloss"]) == len(b["qloss"]) == len(c["qloss"]) > 100
    for key in ("qloss", "grad_norm"):
        _close_then_sane(a, b, 2e-5, key)
        # the default Adam perturbs EVERY weight in the seventh digit at every step (host double vs device float32 bias
        # corrections): after a dozen steps of this tiny net that is 1 % of a gradient norm — the first steps, then statistics
        _close_then_sane(a, c, 1e-4, key, HEAD=5)
    assert a["qloss"][0] == c["qloss"][0]
    assert any(not torch.equal(x, y) for x, y in zip(a["target"], a["params"]))          # the target net is not the online net


def test_iqn_prioritized_learner_step_from_a_graph_is_the_eager_step():
    """Rainbow-style IQN on prioritized replay, T = 1: quantile fractions drawn by torch.rand INSIDE the captured step (the
    generator's offset advances per replay exactly as per eager call: the same fractions), priorities written back after
    every replay."""
    a, b = _series("iqn", True), _series("iqn", "no-capture")
    assert a["captured"] and not b["captured"]
    assert len(a["qloss"]) == len(b["qloss"]) > 100
    for key in ("qloss", "grad_norm"):
        _close_then_sane(a, b, 5e-5, key)
    va, vb = a["tree"][0], b["tree"][0]
    assert np.isfinite(va).all() and abs(va[1] - vb[1]) <= 0.35 * abs(vb[1])                 # total priority mass
