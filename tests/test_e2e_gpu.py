"""GPU: end-to-end pin of THE LOOP.  The unmodified reference trained
DQN + LSTM + prioritized sequence replay + burn-in + double-Q on CPU from a
scripted actor stream (tests/golden/generate.py: run_e2e_case); the same stream,
seeds and initial weights go through rltime_amd's loop on the GPU (device
replay, fused target/loss kernels, fused LSTM) and the per-learner-step loss and
gradient-norm series must follow the reference's.

Tolerance: 2e-3 relative over the first 40 steps — this is a *trajectory* (40
Adam updates, CPU fp32 vs GPU fp32 kernels), not a single evaluation; single
evaluations are held to 1e-4 in test_qmath_gpu.py."""
import copy
import io
import json
import os
import random

import numpy as np
import pytest
import torch

from tests import scenario
from tests.golden.streams import StreamSpec, vector_steps, as_reference_samples

pytestmark = pytest.mark.gpu


def test_training_series_follows_reference():
    from rltime_amd.acting.acting_interface import ActingInterface
    from rltime_amd.general.loggers import NullLogger
    from rltime_amd.spaces import Box, Discrete
    from rltime_amd.training.dqn import DQN
    d = np.load(os.path.join(scenario.GOLDEN, "e2e_dqn_lstm_per.npz"))
    cfg = json.loads(str(d["config"]))
    spec = StreamSpec(**cfg["spec"])

    class ScriptedActor(ActingInterface):
        def __init__(self):
            super().__init__(Box(0, 255, spec.frame_shape, np.uint8), Discrete(spec.n_actions))
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

    random.seed(cfg["seed"]); np.random.seed(cfg["seed"]); torch.manual_seed(cfg["seed"])
    pargs = dict(cfg["policy_args"])
    pargs["cuda"] = True
    tr = DQN(logger=NullLogger(), actors=ScriptedActor(), model_config=cfg["model"], policy_args=pargs)
    series = {"qloss": [], "grad_norm": []}
    orig = tr.value_log.log

    def tap(key, value, *a, **k):
        if key in series and k.get("group") == "train":
            series[key].append(float(value.item() if hasattr(value, "item") else value))
        return orig(key, value, *a, **k)
    tr.value_log.log = tap
    real_init = tr.init_policies

    def init_from_reference():
        real_init()
        tr.policy.load_state_dict(torch.load(io.BytesIO(d["init_online"].tobytes()), map_location="cuda"))
        tr.target_policy.load_state_dict(torch.load(io.BytesIO(d["init_target"].tobytes()), map_location="cuda"))
    tr.init_policies = init_from_reference
    tr.train(**copy.deepcopy(cfg["train"]))
    n = 40
    assert len(series["qloss"]) == len(d["qloss"])          # same number of learner steps
    np.testing.assert_allclose(series["qloss"][:n], d["qloss"][:n], rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(series["grad_norm"][:n], d["grad_norm"][:n], rtol=2e-3, atol=1e-5)
    print("max rel dev over all %d steps: qloss %.2e" % (
        len(d["qloss"]), np.max(np.abs(np.array(series["qloss"]) - d["qloss"]) / np.abs(d["qloss"]))))


def test_shared_online_cnn_gives_the_same_trajectory():
    """share_online_cnn (one conv pass over the union of states / target_states
    rows for the online net) vs two separate passes: same loss / grad-norm series."""
    from rltime_amd.acting.acting_interface import ActingInterface
    from rltime_amd.general.loggers import NullLogger
    from rltime_amd.spaces import Box, Discrete
    from rltime_amd.training.iqn import IQN
    d = np.load(os.path.join(scenario.GOLDEN, "e2e_dqn_lstm_per.npz"))
    cfg = json.loads(str(d["config"]))
    spec = StreamSpec(**cfg["spec"])

    class Scripted(ActingInterface):
        def __init__(self):
            super().__init__(Box(0, 255, spec.frame_shape, np.uint8), Discrete(spec.n_actions))
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

    runs = []
    for share in (True, False):
        random.seed(3); np.random.seed(3); torch.manual_seed(3)
        tr = IQN(logger=NullLogger(), actors=Scripted(), model_config=cfg["model"],
                 policy_args={"dueling": True, "cuda": True, "embedding_dim": 8, "num_sampling_quantiles": 4})
        series = []
        orig = tr.value_log.log

        def tap(key, value, *a, _s=series, _o=orig, **k):
            if key in ("qloss", "grad_norm") and k.get("group") == "train":
                _s.append(float(value.item() if hasattr(value, "item") else value))
            return _o(key, value, *a, **k)
        tr.value_log.log = tap
        args = copy.deepcopy(cfg["train"])
        args["share_online_cnn"] = share
        tr.train(**args)
        runs.append(series)
    assert len(runs[0]) == len(runs[1]) > 40
    np.testing.assert_allclose(runs[0][:60], runs[1][:60], rtol=2e-4, atol=1e-6)


def _scripted_actor(spec):
    from rltime_amd.acting.acting_interface import ActingInterface
    from rltime_amd.spaces import Box, Discrete

    class ScriptedActor(ActingInterface):
        def __init__(self):
            super().__init__(Box(0, 255, spec.frame_shape, np.uint8), Discrete(spec.n_actions))
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
    return ScriptedActor()


def _iqn_series_with_replayed_taus(fixture, channels_last=False, train_args=None):
    """Train rltime_amd's IQN on the fixture's scripted stream from the reference's initial weights, replaying the
    reference's quantile fractions in call order -> (series, fixture arrays).  channels_last: the CNN keeps NHWC
    activations (the shipped model configs' layout, rltime_amd/configs/models/modules/nature_cnn.json; a memory layout,
    not a change of arithmetic) — what the hand-written conv kernels take."""
    from rltime_amd.general.loggers import NullLogger
    from rltime_amd.training.iqn import IQN
    d = np.load(os.path.join(scenario.GOLDEN, fixture))
    cfg = json.loads(str(d["config"]))
    if channels_last:
        cfg["model"]["args"]["layer_configs"][0]["args"]["channels_last"] = True
    spec = StreamSpec(**cfg["spec"])
    sizes, flat = d["tau_sizes"], torch.from_numpy(d["taus"])
    cursor = {"call": 0, "at": 0}

    def replay(count):
        i = cursor["call"]
        assert i < len(sizes) and int(sizes[i]) == count, (i, count, int(sizes[i]) if i < len(sizes) else None)
        out = flat[cursor["at"]:cursor["at"] + count]
        cursor["call"], cursor["at"] = i + 1, cursor["at"] + count
        return out

    random.seed(cfg["seed"]); np.random.seed(cfg["seed"]); torch.manual_seed(cfg["seed"])   # noqa: E702
    pargs = dict(cfg["policy_args"])
    pargs["cuda"] = True
    tr = IQN(logger=NullLogger(), actors=_scripted_actor(spec), model_config=cfg["model"], policy_args=pargs)
    series = {"qloss": [], "grad_norm": []}
    orig = tr.value_log.log

    def tap(key, value, *a, **k):
        if key in series and k.get("group") == "train":
            series[key].append(float(value.item() if hasattr(value, "item") else value))
        return orig(key, value, *a, **k)
    tr.value_log.log = tap
    real_init = tr.init_policies

    def init_from_reference():
        real_init()
        if cfg.get("init_seeds"):
            # large model: the generator loaded seeded weights into the reference's policies (same names and shapes)
            from tests.golden.streams import seeded_weights
            for pol, seed in zip((tr.policy, tr.target_policy), cfg["init_seeds"]):
                host = {k: v.cpu() for k, v in pol.state_dict().items()}
                pol.load_state_dict(seeded_weights(host, seed))
        else:
            tr.policy.load_state_dict(torch.load(io.BytesIO(d["init_online"].tobytes()), map_location="cuda"))
            tr.target_policy.load_state_dict(torch.load(io.BytesIO(d["init_target"].tobytes()), map_location="cuda"))
        tr.policy.tau_source = tr.target_policy.tau_source = replay
    tr.init_policies = init_from_reference
    args = copy.deepcopy(cfg["train"])
    args["burn_in_full_forward"] = True
    args.update(train_args or {})
    tr.train(**args)
    assert cursor["call"] == len(sizes)                      # same number and order of tau draws
    assert len(series["qloss"]) == len(d["qloss"])
    return series, d


def test_iqn_lstm_training_series_follows_reference_with_replayed_taus():
    """The headline algorithm end to end: recurrent IQN (dueling, double-Q,
    rnn_bootstrap, burn-in) + prioritized sequence replay, trained by the unmodified
    reference on CPU (tests/golden/generate.py: run_e2e_iqn_case).  The reference
    draws its quantile fractions with torch.rand on the CPU; the fixture holds every
    tau tensor in call order and this run replays them through IQNPolicy.tau_source
    (burn_in_full_forward=True: the reference's burn-in runs the whole head and so
    consumes taus, multi_step_trainer.py:104-117).  Same tolerance as the DQN-LSTM
    series: 2e-3 over the first 40 Adam steps of a CPU-fp32 vs GPU-fp32 trajectory."""
    series, d = _iqn_series_with_replayed_taus("e2e_iqn_lstm_per.npz")
    n = 40
    np.testing.assert_allclose(series["qloss"][:n], d["qloss"][:n], rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(series["grad_norm"][:n], d["grad_norm"][:n], rtol=2e-3, atol=1e-5)
    print("IQN-LSTM e2e: max rel dev over all %d steps: qloss %.2e" % (
        len(d["qloss"]), np.max(np.abs(np.array(series["qloss"]) - d["qloss"]) / np.abs(d["qloss"]))))


# kernels of rounds 2-3 the wide trajectory must have gone through (names as mirl_profile_* records them)
ROUND3_KERNELS = ("k_gemm3_nt", "k_gemm3_nn", "k_gemm3_tn", "k_gemm3_nt_mul", "k_conv3_fwd", "k_conv1_u8_fwd",
                  "k_conv1_u8_wrw_b3", "k_conv2_bwd_data_b3", "k_conv3_bwd_data_b3", "k_conv_wrw_b3", "k_lstm_seq_fwd", "k_lstm_seq_bwd", "k_tail_bwd")


@pytest.mark.parametrize("forced,nhwc,full_sel,mid", [(True, True, False, 1), ("gemm3", True, False, 1), ("conv", True, False, 1), ("lstm", True, False, 1),
                                                      (False, True, False, 1), (False, False, False, 1), (True, True, True, 1), (True, True, False, 0)],
                         ids=["every-hip-kernel-forced-on", "only-gemm3-forced", "only-conv-forced", "only-persistent-lstm-forced",
                              "library-products-nhwc", "library-products-nchw", "every-hip-kernel-forced-on-full-head-selection",
                              "every-hip-kernel-forced-on-big-gemm-tile"])
def test_wide_iqn_lstm_series_follows_reference_through_the_bf16_pipe_and_persistent_kernels(forced, nhwc, full_sel, mid, monkeypatch):
    """The reference-pinned run that EXECUTES the round-3 arithmetic: the same algorithm as above on a model whose
    layer shapes the hand-written kernels take ((4,36,36) frames -> 32@8/4 -> 64@4/2 -> 64@3/1 -> LSTM 512 ->
    quantile 64 -> FC 128 | value-hidden 128; B = 16 sequences, T = 6, burn-in 4), trained by the unmodified
    reference on CPU (tests/golden/generate.py: E2E_IQN_WIDE).  forced: the work gates are lifted
    (gemm3._MIN_WORK = fused._CONV3_MIN_WORK = 0), so the input layer (bf16 pipe), conv layers 2-3 (split-bf16
    implicit GEMM), layer 2's data gradient, every nn.Linear product (split-bf16 NT / NN / TN, quantile product in the
    epilogue), the dueling tail's fused backward and the persistent LSTM sweeps (forward AND backward) all run inside
    the pinned trajectory — asserted from the per-kernel launch table.  Not forced: the same model with those
    products on the library (MIOpen / hipBLASLt f32) — both must follow the reference equally well.
    full_sel: the double-Q selection from the full dueling head (selection_advantage_only=False: V + A - mean_a A like the
    reference, instead of the advantage stream alone) — the same bars, so a deviation cannot hide behind that optimisation.
    mid: the plain NT products of this small model take the 256 x 128 tile (k_gemm3_mid, round 6: products with few big tiles);
    the last variant switches it off so that the 256 x 256 NT kernel runs inside a pinned trajectory too.
    Bar: 2e-3 over the first 40 Adam steps (rltime/training/torch/iqn.py:54-129, multi_step_trainer.py:278-353)."""
    from rltime_amd import _lib
    from rltime_amd.models.torch import fused, gemm3, lstm_seq
    on = lambda family: forced is True or forced == family           # noqa: E731
    monkeypatch.setattr(gemm3, "_MIN_WORK", 0 if on("gemm3") else 1 << 62)
    monkeypatch.setattr(fused, "_CONV3_MIN_WORK", 0 if on("conv") else 1 << 62)
    monkeypatch.setattr(fused, "_CONV_WRW_MIN_WORK", 0 if on("conv") else 1 << 62)
    monkeypatch.setattr(fused, "_CONV3_BWD", on("conv"))
    monkeypatch.setattr(lstm_seq, "_PERSISTENT", on("lstm"))
    _lib.check(_lib.lib.mirl_conv1_bf16_set(1 if on("conv") else 0))
    _lib.check(_lib.lib.mirl_gemm3_mid_set(mid))
    _lib.check(_lib.lib.mirl_profile_reset())
    _lib.check(_lib.lib.mirl_profile_set(2))
    try:
        series, d = _iqn_series_with_replayed_taus("e2e_iqn_lstm_per_wide.npz", channels_last=nhwc,
                                                   train_args={"selection_advantage_only": False} if full_sel else None)
        torch.cuda.synchronize()
        table = {r["name"]: r["calls"] for r in _lib.profile_table()}
    finally:
        _lib.check(_lib.lib.mirl_profile_set(0))
        _lib.check(_lib.lib.mirl_conv1_bf16_set(-1))
        _lib.check(_lib.lib.mirl_gemm3_mid_set(-1))
    n = 40
    dev = np.abs(np.array(series["qloss"]) - d["qloss"]) / np.abs(d["qloss"])
    gdev = np.abs(np.array(series["grad_norm"]) - d["grad_norm"]) / np.abs(d["grad_norm"])
    label = {True: "all forced" + (", full-head selection" if full_sel else ""), False: "library nhwc" if nhwc else "library nchw"}.get(forced, "only %s forced" % forced)
    line = "wide IQN-LSTM e2e (%s): max rel dev over the first 10 / 20 / 40 / all %d steps: qloss %.2e / %.2e / %.2e / %.2e, " \
           "grad norm %.2e / %.2e / %.2e / %.2e" % (label, len(d["qloss"]), dev[:10].max(), dev[:20].max(), dev[:n].max(), dev.max(),
                                                  gdev[:10].max(), gdev[:20].max(), gdev[:n].max(), gdev.max())
    print(line)
    art = os.environ.get("MIRL_TEST_ARTIFACTS")
    if art:
        os.makedirs(art, exist_ok=True)
        with open(os.path.join(art, "wide_e2e_deviation.txt"), "a") as f:
            f.write(line + "\n")
    np.testing.assert_allclose(series["qloss"][:n], d["qloss"][:n], rtol=2e-3, atol=1e-5)
    # gradient norms: 2e-3 like every other pinned trajectory — except at the steps where the REFERENCE's own run holds a
    # double-Q near-tie.  The fixture records, per learner step, the smallest gap between the best and second-best action of
    # the selection scores (mean over eight quantiles, training/torch/iqn.py:36-45) over the 96 batch rows
    # (tests/golden/generate.py boot_with_gap: sel_gap_min / sel_scale).  A float32 re-implementation whose scores differ
    # from the reference's in the fifth digit resolves such a row the other way, and that row then bootstraps from another
    # action's quantiles: a discrete change of one target, not rounding drift.  In the reference's run the first such step
    # is step 17 (gap 8.0e-5 of the scores' magnitude; every earlier step >= 2.8e-4) — exactly the one step every
    # arithmetic variant of round 4 showed at 2.445e-3 (profiles/r04_wide_e2e_deviation_by_kernel_family.txt).  Those
    # steps, and only those, get the 6e-3 bar; everything else is held to 2e-3.
    rel = np.abs(np.array(series["grad_norm"][:n]) - d["grad_norm"][:n]) / np.abs(d["grad_norm"][:n])
    near_tie = (d["sel_gap_min"][:n] / d["sel_scale"][:n]) <= 1e-4
    assert 1 <= int(near_tie.sum()) <= 6, near_tie.nonzero()
    assert rel[~near_tie].max() <= 2e-3, (rel, near_tie.nonzero())
    assert rel[near_tie].max() <= 6e-3, (rel, near_tie.nonzero())
    if forced is True:
        plain_nt = "k_gemm3_nt_mid" if mid else "k_gemm3_nt"
        missing = [k for k in ROUND3_KERNELS if not table.get(plain_nt if k == "k_gemm3_nt" else k)]
        assert not missing, ("kernels that never ran inside the pinned trajectory", missing, table)
    elif forced is False:
        ran = [k for k in ROUND3_KERNELS if table.get(k) and k.startswith(("k_gemm3", "k_conv3", "k_lstm_seq"))]
        assert not ran, ran
