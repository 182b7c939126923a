"""Replays the golden replay scenarios (tests/golden/replay_*.npz, produced by
the unmodified reference) against a history-buffer implementation."""
import json
import os
import random

import numpy as np

from tests.golden.streams import StreamSpec, vector_steps, as_reference_samples

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENARIOS = ["uniform_t1", "uniform_seq", "uniform_seq_noxing", "per_t1",
             "per_seq", "per_seq_global", "per_seq_tuple_obs", "per_seq_noxing",
             "per_seq_full_overlap"]

# importance weights are floating point: the reference evaluates them in f32 or
# f64 depending on scalar kinds, the device in f64 -> f32.
WEIGHT_RTOL = 2e-6


def load(name):
    d = np.load(os.path.join(GOLDEN, "replay_%s.npz" % name))
    return d, json.loads(str(d["config"]))


def flatten(prefix, tree, out):
    if isinstance(tree, dict):
        for k, v in tree.items():
            flatten(prefix + "." + k if prefix else k, v, out)
    elif isinstance(tree, (tuple, list)):
        for i, v in enumerate(tree):
            flatten("%s.%d" % (prefix, i), v, out)
    else:
        out[prefix] = tree
    return out


def to_numpy(x):
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def make_tensor_dtype(a):
    """models/torch/utils.py:95-123: u8 and f32 stay, everything else -> f32."""
    if a.dtype in (np.uint8, np.float32):
        return a
    return a.astype(np.float32)


def check_batch(gold, tag, batch, exact_dtypes):
    """Compare one get_train_data result with the golden batch."""
    got = {k: to_numpy(v) for k, v in flatten("", batch, {}).items()}
    want = {k[len(tag + ".batch."):]: gold[k] for k in gold.files
            if k.startswith(tag + ".batch.")}
    assert set(got) == set(want), (sorted(got), sorted(want))
    for key, w in want.items():
        g = got[key]
        assert g.shape == w.shape, (key, g.shape, w.shape)
        if exact_dtypes:
            assert g.dtype == w.dtype, (key, g.dtype, w.dtype)
            assert np.array_equal(g, w), key
            continue
        if key.endswith("importance_weights"):
            np.testing.assert_allclose(g, w.astype(np.float32), rtol=WEIGHT_RTOL, err_msg=key)
        elif key.endswith("actions") or key.endswith("loss_indices"):
            assert g.dtype == np.int64 and np.array_equal(g, w), key
        else:
            w32 = make_tensor_dtype(w)
            assert g.dtype == w32.dtype, (key, g.dtype, w32.dtype)
            assert np.array_equal(g, w32), key


def feed_dicts(buf, spec, steps):
    """The reference's hand-over: one list of per-env sample dicts per vector step (actor.py:132-145)."""
    for step in steps:
        buf.update(as_reference_samples(spec, step))


def _device_fields(buf, spec, step):
    """One vector step as the device actor emits it (acting/actor.py _device_steps): (E, ...) device tensors."""
    import torch
    dev = buf.device
    f = {"frames": torch.from_numpy(step["frames"]).to(dev),
         "actions": torch.from_numpy(step["actions"].astype(np.int32)).to(dev),
         "rewards": torch.from_numpy(step["rewards"].astype(np.float32)).to(dev),
         "dones": torch.from_numpy(step["dones"].astype(np.uint8)).to(dev),
         "policy": torch.from_numpy(step["qvalues"]).to(dev)}
    if spec.extra_features:
        f["extra"] = torch.from_numpy(step["extra"]).to(dev)
    if spec.lstm_units:
        f["state"] = torch.from_numpy(np.concatenate([step["hx"], step["cx"]], axis=1)).to(dev)
        f["initials"] = torch.from_numpy(step["initials"]).to(dev)
    return f


def feed_device_samples(buf, spec, steps):
    """acting_interface.DeviceSamples through History.update (-> update_batch without host env ids): what
    Actor.get_samples returns in device mode."""
    from rltime_amd.acting.acting_interface import DeviceSamples
    if not steps:
        return
    example = as_reference_samples(spec, steps[0])[0]["next_state"]
    batch = DeviceSamples(example, spec.num_envs, spec.env_base)
    for step in steps:
        batch.append(**_device_fields(buf, spec, step))
    buf.update(batch)


def feed_planned(buf, spec, steps):
    """The fused rollout's ingest (acting/fast_step.py): the host bookkeeping of a whole run of vector steps first
    (mirl_replay_ingest_plan), then one fused device launch per step (mirl_replay_ingest_planned)."""
    if not steps:
        return
    if buf._h is None:
        example = as_reference_samples(spec, steps[0])[0]["next_state"]
        buf.configure(example, spec.num_envs, spec.env_base, policy_f32=spec.n_actions if buf._keep_policy else 0)
    assert buf.supports_planned_ingest()
    for at in range(0, len(steps), 48):              # the plan buffer is sized for >= 64 steps at its first call
        chunk = steps[at:at + 48]
        buf.plan_ingest(len(chunk), spec.num_envs)
        for k, step in enumerate(chunk):
            f = _device_fields(buf, spec, step)
            if not buf._policy_f32:
                f.pop("policy")
            buf.ingest_planned(k, **f)


FEEDS = {"dicts": feed_dicts, "device_samples": feed_device_samples, "planned": feed_planned}


def run(name, make_buffer, exact_dtypes, per_state=None, on_round=None, feeder=feed_dicts):
    """Drive ``make_buffer(cfg, gamma)`` through the scenario script.  `feeder(buf, spec, steps)` hands a run of vector
    steps to the buffer (default: the reference's per-env dict lists).

    per_state(buf) -> dict with leaf_val/leaf_kind/free_slots/slot_env/slot_base/
    env_first (global env ids) or None to skip the priority-state checks."""
    gold, cfg = load(name)
    spec = StreamSpec(**cfg["spec"])
    buf = make_buffer(cfg, cfg["gamma"])
    step_no, rnd = 0, 0
    last_batch = None
    for op in cfg["script"]:
        if op[0] == "feed":
            feeder(buf, spec, list(vector_steps(spec, op[1], start_step=step_no)))
            step_no += op[1]
            continue
        tag = "r%d" % rnd
        rnd += 1
        if op[0] == "draw":
            _, B, seed, progress = op
            random.seed(seed)
            np.random.seed(seed)
            feed = buf.needed_feed_count(B, spec.num_envs)
            assert (-1 if feed is None else feed) == int(gold[tag + ".feed_count"]), tag
            batch = buf.get_train_data(B, train_progress=progress)
            assert int(gold[tag + ".quota_after"]) == buf.train_quota, tag
            assert (batch is None) == bool(gold[tag + ".is_none"]), tag
            if batch is not None:
                check_batch(gold, tag, batch, exact_dtypes)
                last_batch = batch
        else:
            idx = gold[tag + ".indices"]
            losses = gold[tag + ".losses"]
            buf.update_losses(idx, losses)
        if per_state is not None and (tag + ".leaf_val") in gold.files:
            st = per_state(buf)
            if st is not None:
                assert np.array_equal(st["free_slots"], gold[tag + ".free_slots"]), tag
                assert np.array_equal(st["slot_env"], gold[tag + ".slot_env"]), tag
                assert np.array_equal(st["slot_base"], gold[tag + ".slot_base"]), tag
                assert np.array_equal(st["env_first"], gold[tag + ".env_first"]), tag
                if "leaf_kind" in st:
                    assert np.array_equal(st["leaf_kind"], gold[tag + ".leaf_kind"]), tag
                    check_leaf_values(st["leaf_val"], gold[tag + ".leaf_val"],
                                      gold[tag + ".leaf_kind"], tag, st.get("leaf_exact", False))
        if on_round is not None:
            on_round(tag, buf, gold, last_batch)
    return buf


def check_leaf_values(got, want, kind, tag, exact):
    """Leaf priorities are floating point (x ** alpha through the platform's
    pow): bit-exact for the oracle, <= 1 ulp of their own kind on the device."""
    if exact:
        assert np.array_equal(got, want), tag
        return
    f32 = kind == 1
    np.testing.assert_allclose(got[f32], want[f32], rtol=1.3e-7, atol=0, err_msg=tag)
    np.testing.assert_allclose(got[~f32], want[~f32], rtol=4.5e-16, atol=0, err_msg=tag)
