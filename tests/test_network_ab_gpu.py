"""GPU: the whole learner evaluation at BASELINE configs[3] SHAPES — nature CNN on (4,84,84) uint8 frames,
LSTM 512, dueling IQN head with 32 quantiles, T = 80 train steps behind a 40-step burn-in, n = 2, prioritized
sequence replay — run twice on the same weights, the same gathered batch and the same quantile fractions:

  hip   every product on the hand-written kernels: input layer and conv layers 2-3 on the bf16 matrix pipe
        (exact operand splits, csrc/conv_in.hip, conv3.hip), nn.Linear products on csrc/gemm3.hip (NT / NN / TN,
        quantile product in the epilogue), persistent LSTM sweeps (csrc/lstm_seq.hip)
  lib   MIRL_GEMM3=0, conv layers on MIOpen / the f32-MFMA input kernel, LSTM as one rocBLAS GEMM + cell kernel
        per step: the arithmetic of rounds 1-2 (f32 MFMA pipe everywhere)

and, on a 16-sequence slice of the same batch, against oracle/network64.py — an independent float64
restatement of the reference's model and learner evaluation (rltime/models/torch/modules/{cnn,lstm,fc}.py,
policies/torch/{iqn,dqn}.py, training/multi_step_trainer.py:90-131,278-353, training/torch/iqn.py:15-129).

What is compared (reference semantics: training/torch/iqn.py:54-129, torch_trainer.py:101-147): the n-step
targets, the loss, the reported mean |td| per transition (the replay's priority signal) and EVERY parameter
gradient.  Bars (north_star: 1e-4 fp32): hip vs lib — targets <= 1e-5 of their scale (and the bootstrap part
gamma^n v alone <= 1e-4 of ITS scale: a random-init net's values are ~0.02 against returns of +-1, so one float32
rounding of the sum is already 3e-6 of it), loss <= 1e-5 relative, report <= 1e-4; against float64 — the hip path no
further away than max(2 x the library path's own distance, 1e-6) and inside 1e-4.  Gradients: per parameter no further
from float64 than the library path; the two float32 paths no further apart than either is from float64; global
gradient norm within 1e-4 (see the comment at the assertions for why an absolute 1e-4 per entry is not meaningful).
Rows whose double-Q action choice (argmax of a mean over 32 quantiles) is a numerical tie legitimately pick
another action in the two paths; they are counted, bounded, and left out of the target comparison."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T, P, NSTEP = 80, 40, 2


def _build(B, envs=16, steps_per_env=700):
    from rltime_amd.general.config import load_config
    from rltime_amd.general.loggers import NullLogger
    from rltime_amd.general.utils import deep_dictionary_update
    from rltime_amd.train import create_actors
    from rltime_amd.training.iqn import IQN
    config = load_config("synthetic_atari_iqn_lstm.json")
    deep_dictionary_update(config, {"acting": {"actor_envs": envs}, "training": {"args": {
        "mbatch_size": B, "warmup_steps": 0, "total_steps": 10 ** 12, "log_freq": 10 ** 12,
        "history_mode": {"args": {"size": envs * steps_per_env, "device_rng": True}}}}})
    ta = config["training"]["args"]
    assert (ta["nstep_train"], ta["burn_in_timesteps"], ta["nstep_target"]) == (T, P, NSTEP)
    actors = create_actors(config, "cuda", device_acting=True, use_graph=True)
    tr = IQN(logger=NullLogger(), actors=actors, model_config=config["model"], policy_args=config["policy_args"])
    tr.setup(**ta)
    # the target net is a different set of weights, as after some training (a fresh run copies the online net)
    with torch.no_grad():
        g = torch.Generator(device="cuda").manual_seed(11)
        for p in tr.target_policy.parameters():
            p.add_(torch.randn(p.shape, device="cuda", generator=g) * p.abs().mean() * 0.05)
    hist = tr.history_buffer
    fed = 0
    while fed < envs * (steps_per_env - 20):
        tr._sample_and_update_history(envs * 64)
        fed += envs * 64
    return tr, hist, config


class _Taus:
    """Deterministic quantile fractions for both policies, logged in draw order."""

    def __init__(self):
        self.log = []
        self.g = torch.Generator().manual_seed(2024)

    def reset(self):
        self.log = []
        self.g.manual_seed(2024)

    def __call__(self, count):
        t = torch.rand(count, generator=self.g)
        self.log.append(t)
        return t


def _set_mode(monkeypatch, mode):
    from rltime_amd import _lib
    from rltime_amd.models.torch import fused, gemm3, lstm_seq
    if mode == "hip":
        monkeypatch.delenv("MIRL_GEMM3", raising=False)
        monkeypatch.setattr(gemm3, "_MIN_WORK", 0)
        monkeypatch.setattr(fused, "_CONV3", True)
        monkeypatch.setattr(fused, "_CONV3_MIN_WORK", 0)
        monkeypatch.setattr(fused, "_CONV_WRW", True)
        monkeypatch.setattr(fused, "_CONV_WRW_MIN_WORK", 0)
        monkeypatch.setattr(fused, "_CONV3_BWD", True)
        monkeypatch.setattr(lstm_seq, "_PERSISTENT", True)
        _lib.check(_lib.lib.mirl_conv1_bf16_set(1))
    else:
        monkeypatch.setenv("MIRL_GEMM3", "0")
        monkeypatch.setattr(fused, "_CONV3", False)
        monkeypatch.setattr(fused, "_CONV_WRW", False)
        monkeypatch.setattr(fused, "_CONV3_BWD", False)
        monkeypatch.setattr(lstm_seq, "_PERSISTENT", False)
        _lib.check(_lib.lib.mirl_conv1_bf16_set(0))


def _evaluate(tr, hist, sample, b, taus):
    """Re-gather the first `b` sequences of `sample` (bit-identical every time; states and target_states stay two
    views of ONE block, so the shared-feature paths run as in the bench) and run the real learner_step up to the
    gradients.  -> dict(targets, loss, report, grads)."""
    got = {}
    taus.reset()
    data = hist._gather(b, sample["env"][:b].contiguous(), sample["start"][:b].contiguous(),
                        sample["weight"][:b].contiguous(), sample["loss_start"][:b].contiguous())

    def train_batch(states, targets, policy_outputs, extra_data, timesteps):
        got["targets"] = targets.detach().clone()
        tr.optimizer.zero_grad(set_to_none=True)
        tr._compute_grads(states, targets, policy_outputs, extra_data, timesteps)

    def report(losses, extra):
        got["report"] = losses.detach().clone()

    log = tr.value_log.log

    def tap(key, value, *a, **k):
        if key == "qloss":
            got["loss"] = value.detach().double().clone()
        return log(key, value, *a, **k)
    keep = (tr.train_batch, tr._report_losses_if_needed, tr.value_log.log)
    tr.train_batch, tr._report_losses_if_needed, tr.value_log.log = train_batch, report, tap
    try:
        tr.learner_step(data, T, NSTEP, P, T, True, 1, 1)
    finally:
        tr.train_batch, tr._report_losses_if_needed, tr.value_log.log = keep
    torch.cuda.synchronize()
    got["grads"] = {k: p.grad.detach().clone() for k, p in tr.policy.named_parameters()}
    got["returns"] = data["returns"][P:].reshape(-1).clone()
    got["taus"] = list(taus.log)
    return got


def _dev(a, b, scale=None):
    a, b = a.double(), b.double()
    s = float(b.abs().max()) if scale is None else float(scale)
    return float((a - b).abs().max()) / max(s, 1e-30)


def _grad_facts(got, ref):
    """Per-parameter distance of two gradient sets: max entry deviation over the parameter's largest entry, relative
    L2 distance, share of entries off by more than 1e-4 of the largest entry; and the relative distance of the global
    gradient norm (what clip_grad / the logged grad_norm see, torch_trainer.py:177-199)."""
    per = {}
    for k in ref:
        a, b = got[k].double(), ref[k].double()
        s = max(float(b.abs().max()), 1e-30)
        d = (a - b).abs()
        per[k] = {"max": float(d.max()) / s, "l2": float(d.norm()) / max(float(b.norm()), 1e-30),
                  "share_over_1e-4": float((d > 1e-4 * s).double().mean())}
    na = float(torch.sqrt(sum((got[k].double() ** 2).sum() for k in ref)))
    nb = float(torch.sqrt(sum((ref[k].double() ** 2).sum() for k in ref)))
    return per, abs(na - nb) / nb


def _kernels_that_ran():
    from rltime_amd import _lib
    return {r["name"]: r["calls"] for r in _lib.profile_table()}


def test_config_d_learner_evaluation_hip_vs_library_vs_float64(monkeypatch):
    from rltime_amd import _lib
    B, b = 64, 16
    tr, hist, config = _build(B)
    taus = _Taus()
    tr.policy.tau_source = tr.target_policy.tau_source = taus
    batch = hist.get_train_data(B, 0.5)
    assert batch is not None
    sample = dict(hist.last_sample)
    del batch
    out = {}
    try:
        for mode in ("hip", "lib"):
            _set_mode(monkeypatch, mode)
            _lib.check(_lib.lib.mirl_profile_reset())
            _lib.check(_lib.lib.mirl_profile_set(2))
            out[mode, B] = _evaluate(tr, hist, sample, B, taus)
            ran = _kernels_that_ran()
            _lib.check(_lib.lib.mirl_profile_set(0))
            new = [k for k in ("k_gemm3_nt", "k_gemm3_nn", "k_gemm3_tn", "k_gemm3_nt_mul", "k_conv3_fwd", "k_lstm_seq_fwd")
                   if ran.get(k)]
            assert (len(new) == 6) if mode == "hip" else (not new), (mode, ran)
            out[mode, b] = _evaluate(tr, hist, sample, b, taus)
        # the double-Q selection from the advantage stream alone (DQNPolicy.predict_selection, the default) against the
        # full dueling head: an algebraic identity of the arg-max (dqn.py:74-87), checked here on 5 120 real rows
        _set_mode(monkeypatch, "hip")
        tr.selection_advantage_only = False
        out["full-head-selection", B] = _evaluate(tr, hist, sample, B, taus)
        tr.selection_advantage_only = True
    finally:
        _lib.check(_lib.lib.mirl_profile_set(0))
        _lib.check(_lib.lib.mirl_conv1_bf16_set(-1))

    facts = {}
    sel_a, sel_f = out["hip", B]["targets"], out["full-head-selection", B]["targets"]
    rows_off = int(((sel_a - sel_f).abs().amax(1) > 1e-6 * float(sel_f.abs().max())).sum())
    facts["rows_where_advantage_only_selection_picks_another_action"] = rows_off
    # ---- hip vs lib at B = 64 ------------------------------------------------------------------------------
    h, l = out["hip", B], out["lib", B]
    boot_h, boot_l = h["targets"] - h["returns"].unsqueeze(-1), l["targets"] - l["returns"].unsqueeze(-1)
    boot_scale, tgt_scale = float(boot_l.abs().max()), float(l["targets"].abs().max())
    row_dev = (boot_h - boot_l).abs().amax(1)
    # a row whose double-Q choice flipped between the paths differs by the gap between two actions' target
    # quantiles, not by rounding: such rows are numerical ties of mean_N Z_sel (iqn.py:36-45)
    flipped = row_dev > 1e-3 * boot_scale
    facts["rows"] = int(row_dev.numel())
    facts["rows_with_another_double_q_action"] = int(flipped.sum())
    facts["targets_scale"], facts["bootstrap_scale"] = tgt_scale, boot_scale
    facts["targets_dev"] = float(row_dev[~flipped].max()) / tgt_scale               # relative to the targets' own scale
    facts["targets_bootstrap_dev"] = float(row_dev[~flipped].max()) / boot_scale    # ... and to gamma^n v alone (returns are exact)
    facts["loss_rel_dev"] = abs(float(h["loss"]) - float(l["loss"])) / abs(float(l["loss"]))
    facts["report_dev"] = _dev(h["report"][~flipped], l["report"][~flipped])
    per, gn = _grad_facts(h["grads"], l["grads"])
    facts["grad_dev"] = per
    facts["grad_dev_max"] = max(v["max"] for v in per.values())
    facts["grad_l2_max"] = max(v["l2"] for v in per.values())
    facts["grad_norm_rel_dev"] = gn

    # ---- both against float64 on the 16-sequence slice -------------------------------------------------------
    from oracle.network64 import Net64, learner_eval
    strides = [int(c.stride[0]) for c in tr.policy.model.layers[0].layers]
    N = tr.policy.num_sampling_quantiles
    online = Net64(tr.policy.state_dict(), strides, N, "cuda")
    target = Net64(tr.target_policy.state_dict(), strides, N, "cuda")
    data = hist._gather(b, sample["env"][:b].contiguous(), sample["start"][:b].contiguous(),
                        sample["weight"][:b].contiguous(), sample["loss_start"][:b].contiguous())
    s, t = data["states"], data["target_states"]
    raw = {"x": s["x"], "tx": t["x"], "hx": s["layer1_state"]["hx"], "cx": s["layer1_state"]["cx"],
           "thx": t["layer1_state"]["hx"], "tcx": t["layer1_state"]["cx"], "initials": s["layer1_state"]["initials"],
           "tinitials": t["layer1_state"]["initials"], "returns": data["returns"], "nsteps": data["nsteps"],
           "masks": data["target_masks"], "actions": data["policy_outputs"]["actions"],
           "weights": data["extra_data"]["importance_weights"]}
    ref = learner_eval(online, target, raw, out["hip", b]["taus"], tr.gamma, P, kappa=tr.huber_kappa, double_q=True)
    assert all(torch.equal(x, y) for x, y in zip(out["hip", b]["taus"], out["lib", b]["taus"]))
    boot64 = ref["targets"] - raw["returns"][P:].reshape(-1, 1).double()
    bscale, tscale = float(boot64.abs().max()), float(ref["targets"].abs().max())
    anchor = {}
    for mode in ("hip", "lib"):
        o = out[mode, b]
        rd = ((o["targets"] - o["returns"].unsqueeze(-1)).double() - boot64).abs().amax(1)
        ok = rd <= 1e-3 * bscale
        anchor[mode] = {
            "rows_with_another_double_q_action": int((~ok).sum()),
            "targets_dev": float(rd[ok].max()) / tscale,
            "targets_bootstrap_dev": float(rd[ok].max()) / bscale,
            "loss_rel_dev": abs(float(o["loss"]) - float(ref["loss"])) / abs(float(ref["loss"])),
            "report_dev": _dev(o["report"][ok], ref["report"][ok])}
        per, gn = _grad_facts(o["grads"], ref["grads"])
        anchor[mode].update({"grad_dev": per, "grad_dev_max": max(v["max"] for v in per.values()),
                             "grad_l2_max": max(v["l2"] for v in per.values()), "grad_norm_rel_dev": gn})
    facts["vs_float64_b16"] = anchor
    art = os.environ.get("MIRL_TEST_ARTIFACTS")
    if art:
        os.makedirs(art, exist_ok=True)
        with open(os.path.join(art, "network_ab.json"), "w") as f:
            json.dump(facts, f, indent=1)
    print(json.dumps({k: v for k, v in facts.items() if k not in ("grad_dev", "vs_float64_b16")}))
    print(json.dumps({m: {k: v for k, v in anchor[m].items() if k != "grad_dev"} for m in anchor}))

    # ---- the bars -----------------------------------------------------------------------------------------------
    assert rows_off <= 10, rows_off                   # of 5 120: only numerical ties of mean_N A may pick differently
    # forward quantities: north_star's 1e-4 with room to spare, between the two paths and against float64
    assert facts["rows_with_another_double_q_action"] <= max(2, facts["rows"] // 1000), facts
    assert facts["targets_dev"] <= 1e-5, facts
    assert facts["targets_bootstrap_dev"] <= 1e-4, facts
    assert facts["loss_rel_dev"] <= 1e-5, facts
    assert facts["report_dev"] <= 1e-4, facts
    ah, al = anchor["hip"], anchor["lib"]
    assert ah["rows_with_another_double_q_action"] <= 2 and al["rows_with_another_double_q_action"] <= 2, anchor
    for key in ("targets_dev", "targets_bootstrap_dev", "loss_rel_dev", "report_dev"):
        assert ah[key] <= 1e-4, (key, anchor)
        assert ah[key] <= max(2.0 * al[key], 1e-6), (key, anchor)
    # gradients.  Measured on MI355X (profiles/r04_network_ab_config_d.json): against float64 BOTH float32 paths sit at up to
    # 2.4e-3 of a parameter's largest gradient entry — the same figure, on the same parameters (last FC layer 2.2e-3 / 2.4e-3,
    # value-hidden 1.25e-3 / 1.25e-3: hip / library) — because a ReLU unit whose pre-activation is within float32 rounding of
    # zero takes the other branch than in float64 and its whole gradient row switches on or off; everywhere else the hip path
    # is CLOSER to float64 than the library path (LSTM 2.0e-4 vs 2.7e-4, conv stack 0.9-1.9e-4 vs 1.5-2.6e-4).  An absolute
    # 1e-4 per entry is therefore not a bar float32 itself meets at T = 80; the bars that mean something:
    #   * the worst parameter of the hip path is no further from float64 than 1.5 x the library path's worst (max entry and
    #     relative L2; measured 0.5 - 1.0 over most runs), and per parameter hip <= 2 x library — OR below 3e-3, the size of ONE
    #     flipped ReLU unit on the last FC layer, which a per-parameter ratio of two float32 paths cannot exclude: over seven
    #     runs (the sampled batch differs from run to run) that layer read 1.1e-3 / 1.8e-3 / 2.3e-3 / 2.6e-3 on BOTH paths (the
    #     same unit flipped in both), once 2.6e-3 on the hip path and 5.2e-4 on the library path, once 1.5e-3 against 2.8e-3;
    #   * the two float32 paths differ by no more than each differs from float64 (max entry <= 2.5e-3, observed 8.5e-4);
    #   * the global gradient norm — what clipping and the logged series see — agrees to 1e-4.
    # The 3e-3 escape is tied to the event it excuses: it applies only when the float64 evaluation itself holds hidden units
    # of the last FC / value-hidden layers whose pre-activation lies within float32 rounding of zero (oracle/network64.py
    # `relu_near_zero`: |pre| <= 4e-7 of the layer's largest, i.e. a few ulp); with none, every parameter must sit within
    # 2 x the library path's own distance from float64.
    escape = 3e-3 if ref["relu_near_zero"] > 0 else 0.0
    facts["relu_units_within_float32_rounding_of_zero"] = int(ref["relu_near_zero"])
    print(json.dumps({"relu_units_within_float32_rounding_of_zero": facts["relu_units_within_float32_rounding_of_zero"]}))
    for k, v in ah["grad_dev"].items():
        assert v["max"] <= max(2.0 * al["grad_dev"][k]["max"], escape), (k, v, al["grad_dev"][k], escape)
    assert ah["grad_dev_max"] <= max(1.5 * al["grad_dev_max"], escape) and ah["grad_l2_max"] <= max(1.5 * al["grad_l2_max"], 4e-4), \
        (ah["grad_dev_max"], al["grad_dev_max"], ah["grad_l2_max"], al["grad_l2_max"])
    assert facts["grad_dev_max"] <= 2.5e-3, facts["grad_dev_max"]
    assert facts["grad_norm_rel_dev"] <= 1e-4 and ah["grad_norm_rel_dev"] <= max(2.0 * al["grad_norm_rel_dev"], 1e-5), (facts["grad_norm_rel_dev"], ah["grad_norm_rel_dev"], al["grad_norm_rel_dev"])
