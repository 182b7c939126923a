"""CPU: BASELINE configs[0] — the `cartpole_ppo.json` plumbing run: json config -> registry -> sync actor on the
build's CartPole -> OnlineHistoryBuffer -> PPO (A2C + GAE + clipped surrogate) on an ActorCriticPolicy -> logger.

(i)  parity: tests/golden/online_ppo.npz holds what the UNMODIFIED reference produced on a seeded on-policy stream
     (OnlineHistoryBuffer.get_train_data batches under a2c.py's GAE discount; calc_target_values; PPO._compute_grads
     gradients and logged losses on a seeded-weight MLP) — the mirror classes must reproduce the batches bit for bit and
     the targets / gradients / losses to float32 rounding (1e-5 relative, stated here; same torch on both sides).
(ii) the oracle's restatement (oracle.replay.OracleOnline, oracle.qmath.actor_critic_loss) against the same fixture.
(iii) the shipped config trains: episode reward rises on CartPole with 4 sync envs.
No GPU, no HIP library: this path is host-side Python like the reference's."""
import json
import os

import numpy as np
import torch

from tests import scenario
from tests.golden.streams import ONLINE_CASE as CASE, online_vector_steps, seeded_weights

GOLD = os.path.join(scenario.GOLDEN, "online_ppo.npz")
RTOL = 1e-5


def _flat(x):
    return x.reshape((x.shape[0] * x.shape[1],) + tuple(x.shape[2:]))


def _mirror():
    from rltime_amd.general.value_log import ValueLog
    from rltime_amd.history.online_history import OnlineHistoryBuffer
    from rltime_amd.policies.actor_critic import ActorCriticPolicy
    from rltime_amd.spaces import Box, Discrete
    from rltime_amd.training.ppo import PPO
    tr = PPO.__new__(PPO)
    tr.gamma, tr.advlam, tr.vf_coef, tr.adv_norm = CASE["gamma"], CASE["advlam"], CASE["vf_coef"], True
    tr.entropy_factor, tr.entropy_anneal = CASE["entropy_factor"], None
    tr._clip_value, tr._clip_anneal = CASE["clip_value"], None
    tr.vf_scale_epsilon = None
    tr.steps, tr.total_steps = 0, 1
    tr.value_log = ValueLog()
    policy = ActorCriticPolicy.create(model_config=CASE["model"], observation_space=Box(-10, 10, (CASE["obs_dim"],), np.float32),
                                      action_space=Discrete(CASE["n_actions"]), cuda=False)
    policy.load_state_dict(seeded_weights(policy.state_dict(), CASE["weights_seed"]))
    tr.policy = tr.target_policy = policy
    T = CASE["nstep_train"]
    hist = OnlineHistoryBuffer(nstep_target=T, nstep_train=T, discount_function=tr._get_discount_function(CASE["gamma"]))
    return tr, policy, hist


def _check_batch(gold, tag, batch):
    got = {k: scenario.to_numpy(v) for k, v in scenario.flatten("", batch, {}).items()}
    want = {k[len(tag + ".batch."):]: gold[k] for k in gold.files if k.startswith(tag + ".batch.")}
    assert set(got) == set(want), (sorted(got), sorted(want))
    for key, w in want.items():
        assert got[key].dtype == w.dtype and got[key].shape == w.shape, (tag, key, got[key].dtype, w.dtype)
        assert np.array_equal(got[key], w), (tag, key)


def test_online_history_and_ppo_match_the_reference_fixture():
    gold = np.load(GOLD)
    tr, policy, hist = _mirror()
    step_no, rnd, drawn = 0, 0, 0
    for op in CASE["script"]:
        if op[0] == "feed":
            for samples in online_vector_steps(CASE, op[1], step_no):
                assert hist.update(samples) == {"discarded_steps": 0}
            step_no += op[1]
            continue
        tag = "r%d" % rnd
        rnd += 1
        feed = hist.needed_feed_count(op[1], CASE["num_envs"])
        assert (-1 if feed is None else feed) == int(gold[tag + ".feed_count"]), tag
        batch = hist.get_train_data(op[1])
        assert (batch is None) == bool(gold[tag + ".is_none"]), tag
        if batch is None:
            continue
        drawn += 1
        _check_batch(gold, tag, batch)
        data = {k: scenario_map(v, _flat) for k, v in batch.items() if k != "extra_data"}
        targets = tr.calc_target_values(data["returns"], data["target_states"], data["target_masks"], data["nsteps"], 1)
        np.testing.assert_allclose(targets.numpy(), gold[tag + ".targets"], rtol=RTOL, atol=1e-6)
        policy.zero_grad()
        tr._compute_grads(data["states"], targets, data["policy_outputs"], {}, 1)
        for name, prm in policy.named_parameters():
            want = gold[tag + ".grad." + name]
            np.testing.assert_allclose(prm.grad.numpy(), want, rtol=RTOL, atol=1e-6 * float(np.abs(want).max() + 1e-12), err_msg=name)
        log = tr.value_log.get()["train"]
        for key in ("value_loss", "policy_loss", "policy_entropy", "state_value_mean"):
            np.testing.assert_allclose(log[key], float(gold[tag + ".log." + key]), rtol=RTOL, atol=1e-7, err_msg=key)
    assert rnd == int(gold["rounds"]) and drawn >= 4


def scenario_map(tree, f):
    from rltime_amd.general.utils import deep_apply
    return deep_apply(tree, f)


def test_oracle_online_and_loss_match_the_reference_fixture():
    from oracle import qmath
    from oracle import replay as orc
    gold = np.load(GOLD)
    _, policy, _ = _mirror()          # only a network with the fixture's weights: the arithmetic below is the oracle's
    T = CASE["nstep_train"]
    hist = orc.OracleOnline(nstep_target=T, nstep_train=T, discount_function=orc.make_gae_discount(CASE["gamma"], CASE["advlam"]))
    step_no, rnd = 0, 0
    for op in CASE["script"]:
        if op[0] == "feed":
            for samples in online_vector_steps(CASE, op[1], step_no):
                hist.update(samples)
            step_no += op[1]
            continue
        tag = "r%d" % rnd
        rnd += 1
        batch = hist.get_train_data(op[1])
        assert (batch is None) == bool(gold[tag + ".is_none"]), tag
        if batch is None:
            continue
        _check_batch(gold, tag, batch)
        data = {k: orc.tree_map(v, _flat) for k, v in batch.items() if k != "extra_data"}
        mk = policy.make_tensor
        with torch.no_grad():
            boot = policy.get_state_value(mk(data["target_states"]), 1)
            y = mk(data["returns"]) + qmath.gae_bootstrap_discount(boot, mk(data["nsteps"]), CASE["gamma"], CASE["advlam"]) * mk(data["target_masks"])
        np.testing.assert_allclose(y.numpy(), gold[tag + ".targets"], rtol=RTOL, atol=1e-6)
        lp, vals, ent = policy.evaluate_actions(mk(data["states"]), 1, data["policy_outputs"]["actions"])
        _, vloss, gain = qmath.actor_critic_loss(lp, vals, ent, y, mk(data["policy_outputs"]["values"]),
                                                 mk(data["policy_outputs"]["action_log_probs"]), CASE["vf_coef"],
                                                 CASE["entropy_factor"], True, CASE["clip_value"])
        np.testing.assert_allclose(float(vloss), float(gold[tag + ".log.value_loss"]), rtol=RTOL)
        np.testing.assert_allclose(-float(gain), float(gold[tag + ".log.policy_loss"]), rtol=RTOL, atol=1e-7)


def test_round_robin_restarts_after_env_zero_and_drops_delayed_steps():
    """online_history.py:101-103 tests `not self.last_env`: having served env 0 last restarts the walk at the first env
    (like "nothing served yet"); online_history.py:61-73: an env over max_delayed_steps loses its oldest transitions."""
    from rltime_amd.history.online_history import OnlineHistoryBuffer
    hist = OnlineHistoryBuffer(max_delayed_steps=6, nstep_target=2, nstep_train=2, discount_function=lambda n, r, po: 0.9 ** n * r)
    mk = lambda e, t: {"policy_output": {"actions": 0}, "next_state": {"x": np.array([e, t], np.float32)}, "reward": 1.0,   # noqa: E731
                       "done": False, "info": {}, "env_id": e}
    for t in range(4):
        assert hist.update([mk(e, t) for e in range(3)]) == {"discarded_steps": 0}
    b = hist.get_train_data(1)
    assert b["states"]["x"][0, 0, 0] == 0 and hist.last_env == 0            # env 0 first ...
    b = hist.get_train_data(1)
    assert b["states"]["x"][0, 0, 0] == 0 and hist.last_env == 0            # ... and again: `not 0` restarts at the first env
    b = hist.get_train_data(2)
    assert list(b["states"]["x"][0, :, 0]) == [1, 2] and hist.last_env == 2  # env 0 is empty now
    b = hist.get_train_data(1)
    assert b["states"]["x"][0, 0, 0] == 1                                    # after env 2 comes env 0 (empty) -> env 1
    out = {"discarded_steps": 0}
    for t in range(4, 12):
        out = hist.update([mk(2, t)])
    assert out == {"discarded_steps": 1} and len(hist.tracks[2]) == 6
    assert hist.needed_feed_count(4, 3) == 3 and hist.needed_feed_count(3, 3) is None


def test_cartpole_ppo_config_trains(tmp_path):
    """The shipped BASELINE configs[0] file through the product entry (rltime_amd.train), `--num-envs 4` as BASELINE.json
    words it: it runs on the CPU, writes the reference's log rows and weight checkpoint, and the episode reward rises."""
    from rltime_amd.general.config import load_config, validate_config
    from rltime_amd.train import train_from_config
    cfg = load_config("cartpole_ppo.json")
    validate_config(cfg)
    assert cfg["training"]["type"] == "ppo" and cfg["policy_args"] == {"cuda": False} and cfg["env"] == "CartPole-v0"
    assert cfg["training"]["args"]["nstep_train"] == 32 and cfg["model"]["args"]["layer_configs"][0]["args"]["fc_size"] == 64
    trainer = train_from_config("cartpole_ppo.json", num_envs=4, env="CartPole-v1", log_dir=str(tmp_path), log_name="run", seed=1,
                                conf_update={"training": {"args": {"total_steps": 24000, "log_freq": 4000}}})
    assert type(trainer.history_buffer).__name__ == "OnlineHistoryBuffer" and not trainer.policy.is_cuda()
    rows = [json.loads(line) for line in open(tmp_path / "run" / "train.json")]
    assert len(rows) == 6 and os.path.isfile(tmp_path / "run" / "checkpoint.p")
    first, last = rows[0]["last100"]["reward"], rows[-1]["last100"]["reward"]
    assert rows[-1]["this_interval"]["steps_trained"] > 0 and "policy_entropy" in rows[-1]["train"]
    assert last > 2.0 * first and last > 50.0, (first, last)


def test_online_history_equals_the_oracle_on_random_streams():
    """Property check beyond the fixture: random env counts, sequence lengths, n-step targets (with and without
    fixed_target), episode ends, ragged feeding (envs that skip steps) and small max_delayed_steps — the mirror's batches,
    feed decisions, discard counts and served-env order equal the oracle's restatement of online_history.py bit for bit."""
    from oracle import replay as orc
    from rltime_amd.history.online_history import OnlineHistoryBuffer
    rng = np.random.RandomState(7)
    for case in range(25):
        E, T = int(rng.randint(1, 6)), int(rng.randint(1, 7))
        n = int(rng.randint(1, T + 3))
        fixed = bool(rng.rand() < 0.6)
        cap = int(rng.choice([5000, 3 * T, T + 1]))
        gamma, lam = 0.97, float(rng.choice([1.0, 0.9]))
        kw = dict(max_delayed_steps=cap, fixed_target=fixed, nstep_target=n, nstep_train=T)
        mine = OnlineHistoryBuffer(discount_function=orc.make_gae_discount(gamma, lam), **kw)
        ref = orc.OracleOnline(discount_function=orc.make_gae_discount(gamma, lam), **kw)
        t = 0
        for _ in range(40):
            if rng.rand() < 0.7:
                envs = [e for e in range(E) if rng.rand() < 0.85] or [0]
                def samples():
                    r2 = np.random.RandomState(1000 * case + t)
                    return [{"policy_output": {"actions": int(r2.randint(3)), "values": np.float32(r2.randn()), "action_log_probs": np.float32(-r2.rand())},
                             "next_state": {"x": r2.randn(3).astype(np.float32), "layer0_state": {}}, "reward": float(r2.choice([0.0, 1.0, -0.5])),
                             "done": bool(r2.rand() < 0.15), "info": {}, "env_id": e} for e in envs]
                assert mine.update(samples()) == ref.update(samples())
                t += 1
            else:
                B = int(rng.randint(1, E + 2))
                assert mine.needed_feed_count(B, E) == ref.needed_feed_count(B, E)
                a, b = mine.get_train_data(B), ref.get_train_data(B)
                assert (a is None) == (b is None)
                if a is not None:
                    fa = {k: scenario.to_numpy(v) for k, v in scenario.flatten("", a, {}).items()}
                    fb = {k: scenario.to_numpy(v) for k, v in scenario.flatten("", b, {}).items()}
                    assert set(fa) == set(fb)
                    for k in fa:
                        assert fa[k].dtype == fb[k].dtype and np.array_equal(fa[k], fb[k]), (case, k)
                    assert mine.last_env == ref.last_env
