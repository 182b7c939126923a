"""GPU: THE LOOP end to end through the plugin surface (json-style config ->
registry -> device actor -> device replay -> trainer -> logger), tiny networks,
synthetic env.  Checks the wiring the reference exercises with its cartpole
smoke configs (readme.md:60-63): it runs, trains, syncs the target network,
logs the reference's interval keys, and priorities/weights stay finite."""
import copy
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CNN = {"type": "cnn", "args": {"layers": [{"filters": 8, "kernel": 4, "stride": 2},
                                          {"filters": 8, "kernel": 3, "stride": 1}]}}
BASE = {
    "acting": {"actor_envs": 8, "exploration": {"type": "epsilon_greedy", "args": {
        "eps_start": 1.0, "eps_final": 0.05, "exploration_fraction": 0.5}}},
    "env": "synthetic-atari", "env_args": {"frame_shape": [2, 20, 20], "n_actions": 4, "done_prob": 0.02},
    "policy_args": {},
}


def _run(config, device_acting=True):
    from rltime_amd.general.loggers import NullLogger
    from rltime_amd.train import train
    logger = NullLogger()
    trainer = train(copy.deepcopy(config), logger, device_acting=device_acting)
    assert logger.rows, "no log interval was reached"
    last = logger.rows[-1][2]
    assert last["this_interval"]["steps_trained"] > 0
    assert last["this_interval"]["steps_trained_per_second"] > 0
    assert math.isfinite(last["train"]["qloss"]) and math.isfinite(last["train"]["grad_norm"])
    # phase timings: the reference's wall-clock keys plus HIP-event GPU time per phase
    for phase in ("sample_actors", "history_update", "get_train_data", "calc_target_values", "train"):
        assert phase in last["timings_mean_ms"], phase
        assert last["timings_gpu_mean_ms"][phase] >= 0.0
    for p in trainer.policy.parameters():
        assert torch.isfinite(p).all()
    # episode statistics and the action histogram (policy_trainer.py:75-131) — on the
    # device-resident actor they are accumulated on the GPU (acting/episode_tracker.py)
    assert last["total"]["episodes"] > 0
    assert last["last10"]["episode_length"] >= 1 and math.isfinite(last["last10"]["reward"])
    hist = last["acting"]["actions"]
    assert len(hist) > 1 and abs(sum(hist) - 1.0) < 0.02
    return trainer, last


def test_dqn_uniform_replay_loop():
    cfg = dict(BASE)
    cfg["model"] = {"type": "sequential", "args": {"layer_configs": [CNN, {"type": "fc", "args": {"fc_size": 32}}]}}
    cfg["training"] = {"type": "dqn", "args": {
        "clip_rewards": True, "gamma": 0.99, "mbatch_size": 32, "nstep_train": 1, "nstep_target": 3,
        "lr": 1e-3, "lr_anneal": True, "double_q": True, "clip_grad": 10.0, "target_update_freq": 400,
        "total_steps": 4000, "log_freq": 1000, "warmup_steps": 400,
        "history_mode": {"type": "replay", "args": {"size": 1000, "train_frequency": 8}}}}
    trainer, last = _run(cfg)
    # train_frequency=8: every acted sample is trained ~8 times once warmed up
    assert 4.0 < last["this_interval"]["train_ratio"] < 12.0
    # the target network was synced (target_update_freq crossed) and is not the online net
    assert trainer.target_policy is not trainer.policy


def test_recurrent_iqn_prioritized_loop_with_burn_in():
    cfg = dict(BASE)
    cfg["model"] = {"type": "sequential", "args": {"layer_configs": [
        CNN, {"type": "lstm", "args": {"num_units": 16}}, {"type": "fc", "args": {"fc_size": 32}}]}}
    cfg["policy_args"] = {"dueling": True, "embedding_dim": 8, "num_sampling_quantiles": 4}
    cfg["training"] = {"type": "iqn", "args": {
        "clip_rewards": False, "vf_scale_epsilon": 1e-3, "gamma": 0.99, "mbatch_size": 8, "nstep_train": 8,
        "burn_in_timesteps": 4, "nstep_target": 2, "lr": 1e-3, "double_q": True, "rnn_bootstrap": True,
        "clip_grad": 10.0, "clip_grad_dynamic_alpha": 0.9, "target_update_freq": 500,
        "loss_timestep_aggregation": "mean", "loss_aggregation": "sum",
        "total_steps": 3000, "log_freq": 1000, "warmup_steps": 300,
        "history_mode": {"type": "prioritized_replay", "args": {
            "size": 1200, "train_frequency": 4, "alpha": 0.9, "beta": 0.6, "beta_anneal": True,
            "max_weight_factor": 0.9}}}}
    trainer, last = _run(cfg)
    hist = trainer.history_buffer
    v, k, _ = hist.tree_nodes()
    cap = len(v) // 2
    assert np.isfinite(v).all() and v[1] > 0
    assert abs(v[1] - v[cap:].sum()) <= 2e-6 * v[1]      # f32-kind nodes round like np.float32 adds
    assert hist.stats()["total_items"] == 1200          # full, evicting
    assert 0 < last["train"]["importance_weights"] <= 1.0


def test_reference_style_host_actor_feeds_the_device_replay():
    """device=False: the actor emits the reference's per-env sample dicts
    (acting_interface.py:83-90) and History.update regroups them."""
    cfg = dict(BASE)
    cfg["model"] = {"type": "sequential", "args": {"layer_configs": [
        CNN, {"type": "lstm", "args": {"num_units": 8}}, {"type": "fc", "args": {"fc_size": 16}}]}}
    cfg["training"] = {"type": "dqn", "args": {
        "gamma": 0.99, "mbatch_size": 4, "nstep_train": 4, "nstep_target": 2, "lr": 1e-3,
        "target_update_freq": 200, "total_steps": 800, "log_freq": 400, "warmup_steps": 100,
        "history_mode": {"type": "prioritized_replay", "args": {"size": 300, "train_frequency": 4}}}}
    _run(cfg, device_acting=False)


def test_unknown_training_argument_raises_typeerror():
    """The reference consumes training.args through the _train(**kwargs) chain;
    an unknown key is a TypeError (policy_trainer.py:284-286)."""
    from rltime_amd.general.loggers import NullLogger
    from rltime_amd.train import train
    cfg = dict(BASE)
    cfg["model"] = {"type": "sequential", "args": {"layer_configs": [CNN, {"type": "fc", "args": {"fc_size": 8}}]}}
    cfg["training"] = {"type": "dqn", "args": {
        "gamma": 0.99, "nstep_train": 1, "lr": 1e-3, "total_steps": 100, "no_such_option": 1,
        "history_mode": {"type": "replay", "args": {"size": 100, "train_frequency": 4}}}}
    with pytest.raises(TypeError):
        train(copy.deepcopy(cfg), NullLogger())


def test_graph_captured_acting_matches_eager():
    """Actor(use_graph=True): the acting forward replayed from a HIP graph gives
    the same actions / q-values / recurrent state as the eager forward (DQN+LSTM:
    no RNG inside the forward)."""
    from rltime_amd.acting.actor import Actor
    from rltime_amd.acting.synthetic_env import SyntheticAtariVecEnv
    from rltime_amd.policies.dqn import DQNPolicy
    model = {"type": "sequential", "args": {"layer_configs": [
        dict(CNN, args=dict(CNN["args"], channels_last=True)),
        {"type": "lstm", "args": {"num_units": 16}}, {"type": "fc", "args": {"fc_size": 32}}]}}
    outs = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        env = SyntheticAtariVecEnv(8, frame_shape=(2, 20, 20), n_actions=4, seed=3)
        pol = DQNPolicy.create(model_config=model, observation_space=env.observation_space,
                               action_space=env.action_space, dueling=True)
        actor = Actor(env, device=True, use_graph=use_graph)
        actor.set_actor_policy(pol)
        batch = actor.get_samples(8 * 6)
        assert actor._use_graph == use_graph          # capture must not have fallen back
        outs.append([(s["actions"].cpu(), s["policy"].cpu(), s["state"].cpu()) for s in batch.vector_steps])
    for a, b in zip(*outs):
        assert torch.equal(a[0], b[0])
        assert torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-6)
        assert torch.allclose(a[2], b[2], rtol=1e-5, atol=1e-6)


def test_failed_graph_capture_falls_back_to_the_eager_state(monkeypatch):
    """If HIP-graph capture of the acting step fails, the eager fallback must
    continue from the recurrent state the actor had BEFORE the capture attempt
    (GraphedStep re-binds layer.last_state to a static carry and advances it
    through warm-up forwards): the sample stream must equal pure eager acting."""
    from rltime_amd.acting.actor import Actor
    from rltime_amd.acting.synthetic_env import SyntheticAtariVecEnv
    from rltime_amd.policies.dqn import DQNPolicy
    model = {"type": "sequential", "args": {"layer_configs": [
        CNN, {"type": "lstm", "args": {"num_units": 16}}, {"type": "fc", "args": {"fc_size": 32}}]}}

    def run(use_graph):
        torch.manual_seed(0)
        env = SyntheticAtariVecEnv(8, frame_shape=(2, 20, 20), n_actions=4, seed=3)
        pol = DQNPolicy.create(model_config=model, observation_space=env.observation_space,
                               action_space=env.action_space, dueling=True)
        actor = Actor(env, device=True, use_graph=use_graph)
        actor.set_actor_policy(pol)
        batch = actor.get_samples(8 * 6)
        return actor, [(s["actions"].cpu(), s["policy"].cpu(), s["state"].cpu()) for s in batch.vector_steps]

    _, eager = run(False)

    class Boom(torch.cuda.CUDAGraph):
        def __new__(cls, *a, **k):
            raise RuntimeError("capture unsupported (forced by the test)")
    monkeypatch.setattr(torch.cuda, "CUDAGraph", Boom)
    actor, fallen = run(True)
    assert actor._use_graph is False and actor._graphed is None      # it did fall back
    for a, b in zip(eager, fallen):
        assert torch.equal(a[0], b[0])
        assert torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-6)
        assert torch.allclose(a[2], b[2], rtol=1e-5, atol=1e-6)


def _iqn_lstm_config(**targs):
    cfg = dict(BASE)
    cfg["model"] = {"type": "sequential", "args": {"layer_configs": [
        dict(CNN, args=dict(CNN["args"], channels_last=True)),
        {"type": "lstm", "args": {"num_units": 16}}, {"type": "fc", "args": {"fc_size": 32}}]}}
    cfg["policy_args"] = {"dueling": True, "embedding_dim": 8, "num_sampling_quantiles": 4}
    args = {
        "clip_rewards": False, "vf_scale_epsilon": 1e-3, "gamma": 0.99, "mbatch_size": 8, "nstep_train": 8,
        "burn_in_timesteps": 4, "nstep_target": 2, "lr": 1e-3, "double_q": True, "rnn_bootstrap": True,
        "clip_grad": 10.0, "target_update_freq": 200, "total_steps": 1600, "log_freq": 800, "warmup_steps": 200,
        "history_mode": {"type": "prioritized_replay", "args": {
            "size": 600, "train_frequency": 4, "alpha": 0.9, "beta": 0.6, "max_weight_factor": 0.9}}}
    args.update(targs)
    cfg["training"] = {"type": "iqn", "args": args}
    return cfg


def _loss_series(cfg, use_graph=False):
    import random
    from rltime_amd.general.loggers import NullLogger
    from rltime_amd.train import train
    random.seed(4); np.random.seed(5); torch.manual_seed(6)      # noqa: E702
    series = []

    def hook(trainer):
        orig = trainer.value_log.log

        def tap(key, value, *a, **k):
            if key == "qloss" and k.get("group") == "train":
                series.append(float(value.item()))
            return orig(key, value, *a, **k)
        trainer.value_log.log = tap
    old = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        trainer = train(copy.deepcopy(cfg), NullLogger(), use_graph=use_graph, on_trainer=hook)
    finally:
        torch.backends.cudnn.deterministic = old
    return trainer, series


def test_overlapped_acting_trains_and_matches_its_serialised_schedule():
    """overlap_acting=True runs the acting + ingest of iteration k+1 on a second HIP
    stream against iteration k's training.  The same schedule issued on ONE stream
    (overlap_acting="serial") computes the same thing without any concurrency, so
    the two loss series must agree: a missing event dependency (ingest overwriting
    ring slots under the gather, priority-tree updates racing the ingest, the actor
    reading weights mid-copy) would show up as a difference."""
    t1, two_streams = _loss_series(_iqn_lstm_config(overlap_acting=True))
    t2, one_stream = _loss_series(_iqn_lstm_config(overlap_acting="serial"))
    assert len(two_streams) == len(one_stream) > 50
    assert all(math.isfinite(x) for x in two_streams)
    np.testing.assert_allclose(two_streams, one_stream, rtol=1e-6, atol=1e-9)
    assert t1.steps == t2.steps == 1600
    assert t1.history_buffer.stats() == t2.history_buffer.stats()
    print("overlapped == serialised schedule bit-for-bit: %s" % (two_streams == one_stream))


def test_overlapped_acting_with_graph_replay_runs():
    t, series = _loss_series(_iqn_lstm_config(overlap_acting=True, total_steps=1200, log_freq=600), use_graph=True)
    assert len(series) > 50 and all(math.isfinite(x) for x in series)
    assert t.actors._use_graph and t.actors._graphed is not None


@pytest.mark.parametrize("kind", ["dqn_dueling", "iqn_dueling", "dqn_plain"])
def test_fused_actor_head_matches_policy_postprocessing(kind):
    """k_actor_head (dueling combine + quantile mean + argmax + epsilon-greedy in one
    launch) against the policy's own post-processing followed by
    EpsilonGreedyExplorationManager.remap_with_eps_tensor: same seeds -> same tau /
    exploration draws -> same actions and q-values."""
    from rltime_amd.acting.actor import Actor
    from rltime_amd.acting.synthetic_env import SyntheticAtariVecEnv
    from rltime_amd.policies.dqn import DQNPolicy
    from rltime_amd.policies.iqn import IQNPolicy
    model = {"type": "sequential", "args": {"layer_configs": [
        CNN, {"type": "lstm", "args": {"num_units": 16}}, {"type": "fc", "args": {"fc_size": 32}}]}}
    expl = {"type": "epsilon_greedy", "args": {"eps_start": 0.4, "eps_final": 0.01, "eps_min": 0.01,
                                               "per_actor_exponent_factor": 7, "exploration_fraction": 0.5}}
    outs = []
    for fused in (True, False):
        torch.manual_seed(0)
        np.random.seed(0)
        env = SyntheticAtariVecEnv(16, frame_shape=(2, 20, 20), n_actions=5, seed=3)
        if kind == "iqn_dueling":
            pol = IQNPolicy.create(model_config=model, observation_space=env.observation_space, action_space=env.action_space,
                                   dueling=True, embedding_dim=8, num_sampling_quantiles=4)
        else:
            pol = DQNPolicy.create(model_config=model, observation_space=env.observation_space, action_space=env.action_space,
                                   dueling=(kind == "dqn_dueling"))
        actor = Actor(env, exploration_config=expl, device=True, use_graph=False)
        actor.fused_head = fused
        actor.set_actor_policy(pol)
        actor.update_state(0.2)
        batch = actor.get_samples(16 * 8)
        outs.append([(s["actions"].cpu(), s["policy"].cpu()) for s in batch.vector_steps])
    same = total = 0
    for (a1, q1), (a2, q2) in zip(*outs):
        assert torch.allclose(q1, q2, rtol=1e-5, atol=1e-6)
        same += int((a1 == a2).sum())
        total += a1.numel()
        if not torch.equal(a1, a2):
            break                       # a flipped near-tie changes the env trajectory from here on
    assert same >= total - 1, (same, total)


def test_trainer_with_the_persistent_lstm_sweeps_follows_the_per_step_path(monkeypatch):
    """The whole loop (device actor, prioritized sequence replay, burn-in, double-Q IQN targets, backward,
    Adam) with an LSTM the persistent sweep kernels cover (H = 128, B = 16 sequences): the loss / grad-norm
    series with csrc/lstm_seq.hip in every sweep follows the series of the per-step GEMM + cell path from
    the same seeds (the kernels differ by the rounding of the gate activations, ~1e-7 per step)."""
    import random
    from rltime_amd.general.loggers import NullLogger
    from rltime_amd.general.type_registry import get_registered_type
    from rltime_amd.models.torch import lstm_seq
    from rltime_amd.train import create_actors
    cfg = dict(BASE)
    cfg["acting"] = {"actor_envs": 16, "exploration": BASE["acting"]["exploration"]}
    cfg["model"] = {"type": "sequential", "args": {"layer_configs": [
        CNN, {"type": "lstm", "args": {"num_units": 128}}, {"type": "fc", "args": {"fc_size": 32}}]}}
    cfg["policy_args"] = {"dueling": True, "embedding_dim": 8, "num_sampling_quantiles": 4}
    targs = {"clip_rewards": False, "gamma": 0.99, "mbatch_size": 16, "nstep_train": 8, "burn_in_timesteps": 4,
             "nstep_target": 2, "lr": 5e-4, "double_q": True, "rnn_bootstrap": True, "clip_grad": 10.0,
             "target_update_freq": 320, "total_steps": 10 ** 9, "log_freq": 10 ** 9, "warmup_steps": 0,
             "history_mode": {"type": "prioritized_replay", "args": {
                 "size": 2000, "train_frequency": 4, "alpha": 0.9, "beta": 0.6, "device_rng": True}}}
    series = []
    used = []
    for persistent in (True, False):
        monkeypatch.setattr(lstm_seq, "_PERSISTENT", persistent)
        monkeypatch.setattr(lstm_seq, "_BWD_PERSISTENT_MAX_B", 128)
        random.seed(1); np.random.seed(2); torch.manual_seed(3)           # noqa: E702
        actors = create_actors(copy.deepcopy(cfg), "cuda", device_acting=True)
        trainer = get_registered_type("trainers", "iqn")(logger=NullLogger(), actors=actors, model_config=cfg["model"],
                                                       policy_args=cfg["policy_args"])
        trainer.setup(**copy.deepcopy(targs))
        calls = {"n": 0}
        orig = lstm_seq.lib.mirl_lstm_seq_fwd
        got = []
        tap = trainer.value_log.log

        def log(key, value, *a, _tap=tap, _got=got, **kw):
            if key in ("qloss", "grad_norm") and kw.get("group") == "train":
                _got.append(float(value.item() if hasattr(value, "item") else value))
            return _tap(key, value, *a, **kw)
        trainer.value_log.log = log
        steps = 0
        while steps < 12:
            steps += bool(trainer.loop_iteration())
        series.append(got)
        used.append(lstm_seq.persistent_supported(8, 16, 128))
        trainer.history_buffer.close()
        actors.close()
        del calls, orig
    assert used == [True, False]
    a, b = np.array(series[0]), np.array(series[1])
    assert a.shape == b.shape and a.size == 24
    np.testing.assert_allclose(a, b, rtol=2e-3, atol=1e-5)
