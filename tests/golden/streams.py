"""Deterministic synthetic actor streams shared by the golden generator, the
oracle tests and the HIP parity tests.

A *stream* is what a synchronous vectorised actor (reference
rltime/acting/actor.py:97-149) hands to ``History.update`` per vector step: one
transition per env, iteration-major / env-minor.  Frames carry their identity
(env, per-env offset) in their bytes so a gather that picks the wrong row,
the wrong ring slot or the wrong time-major position is caught.
"""
import numpy as np


class StreamSpec:
    def __init__(self, seed, num_envs, frame_shape=(2, 3, 3), lstm_units=0,
                 n_actions=4, done_prob=0.1, fractional_rewards=True,
                 env_base=0, extra_features=0, stacked=False):
        self.seed = seed
        self.num_envs = num_envs
        self.frame_shape = tuple(frame_shape)
        self.lstm_units = lstm_units
        self.n_actions = n_actions
        self.done_prob = done_prob
        self.fractional_rewards = fractional_rewards
        self.env_base = env_base
        # >0: tuple observation (frame, extra f32 vector), as the reference's
        # ExtraFeaturesEnvWrapper produces (env_wrappers/common.py)
        self.extra_features = extra_features
        # True: frames follow the frame-stack wrapper's shift contract (reference
        # env_wrappers/common.py:141-178 under an auto-resetting vec env): the leading
        # axis is a window of planes, newest last; every step rolls it by one plane, a
        # `done` step returns the reset observation (zeros + the new plane)
        self.stacked = stacked


def frame_for(env, offset, shape):
    """u8 frame whose bytes encode (env, offset)."""
    n = int(np.prod(shape))
    k = np.arange(n, dtype=np.int64)
    v = (env * 131 + offset * 7 + k * 29 + (offset >> 8) * 3) % 251
    return v.astype(np.uint8).reshape(shape)


def _advance_stacks(spec, stacks, s, dones):
    plane = spec.frame_shape[1:]
    for e in range(spec.num_envs):
        new = frame_for(spec.env_base + e, s, plane)
        if dones[e]:
            stacks[e] = 0                      # WindowedEnv.reset: zeros, then the first frame
        else:
            stacks[e, :-1] = stacks[e, 1:]     # np.roll by one plane
        stacks[e, -1] = new


def vector_steps(spec, count, start_step=0):
    """Yield ``count`` vector steps as dicts of (E, ...) arrays.  Step ``s`` of
    env ``e`` is transition offset ``s`` of that env."""
    E = spec.num_envs
    palette = np.array([-1.0, 0.0, 0.0, 0.0, 1.0, 0.5, 0.25, 2.0]) \
        if spec.fractional_rewards else np.array([-1.0, 0.0, 0.0, 1.0])
    stacks = None
    if spec.stacked:
        plane = spec.frame_shape[1:]
        # reset observation: zeros + the reset frame (identity offset -1); then replay
        # the steps before start_step (dones are a pure function of (seed, step))
        stacks = np.zeros((E,) + spec.frame_shape, dtype=np.uint8)
        for e in range(E):
            stacks[e, -1] = frame_for(spec.env_base + e, -1, plane)
        for s in range(start_step):
            dones = np.random.RandomState((spec.seed * 1000003 + s) % (2 ** 31))
            dones.randint(0, len(palette), size=E)
            dones = dones.rand(E) < spec.done_prob
            _advance_stacks(spec, stacks, s, dones)
    for s in range(start_step, start_step + count):
        rng = np.random.RandomState((spec.seed * 1000003 + s) % (2 ** 31))
        if spec.stacked:
            peek = np.random.RandomState((spec.seed * 1000003 + s) % (2 ** 31))
            peek.randint(0, len(palette), size=E)
            _advance_stacks(spec, stacks, s, peek.rand(E) < spec.done_prob)
        out = {
            "frames": stacks.copy() if spec.stacked else np.stack([
                frame_for(spec.env_base + e, s, spec.frame_shape)
                for e in range(E)]),
            "rewards": palette[rng.randint(0, len(palette), size=E)],
            "dones": rng.rand(E) < spec.done_prob,
            "actions": rng.randint(0, spec.n_actions, size=E),
            "qvalues": rng.randn(E, spec.n_actions).astype(np.float32),
        }
        if spec.extra_features:
            out["extra"] = rng.randn(E, spec.extra_features).astype(np.float32)
        if spec.lstm_units:
            out["hx"] = rng.randn(E, spec.lstm_units).astype(np.float32)
            out["cx"] = rng.randn(E, spec.lstm_units).astype(np.float32)
            # initials of state t is the done of the transition producing it
            # (reference actor.py:128)
            out["initials"] = out["dones"].astype(np.float32)
        yield out


def as_reference_samples(spec, step, empty_layers=()):
    """One vector step -> the list of per-env sample dicts the reference actor
    would emit (acting_interface.py:83-90, actor.py:132-145).  `empty_layers`
    adds the `layer{i}_state: {}` entries a real model's make_input_state emits
    for its non-recurrent layers (sequential.py:143-145)."""
    samples = []
    for e in range(spec.num_envs):
        state = {"x": step["frames"][e]}
        if spec.extra_features:
            state["x"] = (step["frames"][e], step["extra"][e])
        for i in empty_layers:
            state["layer%d_state" % i] = {}
        if spec.lstm_units:
            state["layer1_state"] = {
                "hx": step["hx"][e], "cx": step["cx"][e],
                "initials": step["initials"][e]}
        samples.append({
            "policy_output": {"actions": step["actions"][e],
                              "qvalues": step["qvalues"][e]},
            "next_state": state,
            "reward": step["rewards"][e],      # np.float64 scalar, as np.stack(rews)[i]
            "done": step["dones"][e],          # np.bool_
            "info": {},
            "env_id": spec.env_base + e,
        })
    return samples


def scalar_kind(v):
    """0 = weak Python scalar (float/int), 1 = np.float32, 2 = np.float64."""
    if isinstance(v, np.float32):
        return 1
    if isinstance(v, np.floating):
        return 2
    return 0


def seeded_weights(state_dict, seed):
    """Deterministic initial weights for a policy, as a pure function of (parameter names, shapes, seed): uniform in
    +-sqrt(1 / fan_in) for matrices / filters (the reference's init_weight range, models/torch/utils.py:6-25) and
    +-0.05 for biases (non-zero, so that every bias path matters).  The golden generator loads them into the
    REFERENCE's policies and the GPU test into the mirror's — large models need no weight blobs in their fixture."""
    import torch
    out = {}
    for i, key in enumerate(sorted(state_dict)):
        t = state_dict[key]
        if not t.dtype.is_floating_point or key.endswith("embedding_range"):
            out[key] = t.clone()
            continue
        rng = np.random.RandomState((seed * 7919 + i * 104729) % (2 ** 31))
        shape = tuple(t.shape)
        bound = 0.05 if len(shape) == 1 else float(1.0 / np.sqrt(np.prod(shape[1:])))
        out[key] = torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))
    return out


# --------------------------------------------------------------------------
# On-policy stream for the online-history / PPO plumbing fixture (BASELINE configs[0])
# --------------------------------------------------------------------------
ONLINE_CASE = {
    "seed": 23, "num_envs": 3, "obs_dim": 4, "n_actions": 2, "done_prob": 0.12,
    "nstep_train": 5, "gamma": 0.99, "advlam": 0.95, "vf_coef": 0.5, "entropy_factor": 1e-2, "clip_value": 0.2,
    "weights_seed": 5,
    "model": {"type": "sequential", "args": {"layer_configs": [{"type": "fc", "args": {"fc_size": 16}},
                                                               {"type": "fc", "args": {"fc_size": 16}}]}},
    # feed N vector steps / draw a batch of B sequences (None expected where the buffer cannot serve it)
    "script": [["feed", 4], ["draw", 2], ["feed", 3], ["draw", 3], ["draw", 1], ["feed", 9], ["draw", 4], ["draw", 2],
               ["feed", 11], ["draw", 6]],
}


def online_vector_steps(case, count, start_step=0):
    """`count` vector steps of the on-policy case as the reference's per-env sample dicts (acting_interface.py:83-90 with
    an actor-critic policy_output, actor_critic.py:63-71): float32 vector observations, unit rewards (CartPole's), the
    acting-time action / log-probability / value estimate drawn from the seed instead of a policy."""
    E = case["num_envs"]
    for s in range(start_step, start_step + count):
        rng = np.random.RandomState((case["seed"] * 1000003 + s) % (2 ** 31))
        obs = rng.randn(E, case["obs_dim"]).astype(np.float32)
        dones = rng.rand(E) < case["done_prob"]
        actions = rng.randint(0, case["n_actions"], size=E)
        logp = np.log(rng.uniform(0.2, 0.8, size=E)).astype(np.float32)
        values = (rng.randn(E) * 3 + 10).astype(np.float32)
        rewards = np.where(rng.rand(E) < 0.9, 1.0, 0.5)
        yield [{"policy_output": {"actions": actions[e], "action_log_probs": logp[e], "values": values[e]},
                "next_state": {"x": obs[e], "layer0_state": {}, "layer1_state": {}},
                "reward": rewards[e], "done": dones[e], "info": {}, "env_id": e} for e in range(E)]
