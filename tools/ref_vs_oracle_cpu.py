#!/usr/bin/env python3
"""DEV CONTAINER ONLY: the unmodified reference and the oracle restatement
(bench.py's CPU baseline path) timed on IDENTICAL inputs, to state how the
"port" baseline bench.py reports relates to the real reference (SURVEY.md
section 8d).  The reference is imported from /root/reference through the import
shims of oracle/ref_shims; nothing of it travels to the GPU box.

    python tools/ref_vs_oracle_cpu.py [B] [steps]     # default B=4, 3 learner steps each

Both sides: config D shapes (T=80, burn-in 40, n=2, (4,84,84) u8 frames, LSTM512,
IQN 32 quantiles, dueling, double-Q, rnn_bootstrap), prioritized sequence replay of
the same 8 x 400 synthetic transitions, torch-CPU fp32, 1 thread (the reference
calls torch.set_num_threads(1), models/torch/torch_model.py:25).  Timed: one learner
step = get_train_data + burn-in + calc_target_values + train_batch (+ update_losses).
Prints one JSON line; BASELINE.md records the ratio."""
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle", "ref_shims"), "/root/reference"]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def oracle_side():
    import bench
    cpu = bench.CpuPath()
    cpu.learner_step(2)
    times = []
    for _ in range(STEPS):
        t = time.time()
        cpu.learner_step(B)
        times.append(time.time() - t)
    return times


def reference_side():
    import gym
    from rltime.acting.acting_interface import ActingInterface
    from rltime.general.config import load_config
    from rltime.training.torch.iqn import IQN as RefIQN
    E, H, A = 8, 512, 6
    rng = np.random.RandomState(0)
    frame_pool = [rng.randint(0, 256, (4, 84, 84)).astype(np.uint8) for _ in range(64)]

    class Scripted(ActingInterface):
        def __init__(self):
            super().__init__(gym.spaces.Box(0, 255, (4, 84, 84), dtype=np.uint8), gym.spaces.Discrete(A))
            self.s = 0

        def get_env_count(self):
            return E

        def set_actor_policy(self, p):
            pass

        def update_state(self, progress, policy_state=None):
            pass

        def close(self):
            pass

        def get_samples(self, min_samples):
            out = []
            for _ in range((max(1, min_samples) + E - 1) // E):
                for e in range(E):
                    out.append(self._create_sample(
                        {"actions": int(rng.randint(A))},
                        {"x": frame_pool[(self.s * E + e) % 64].copy(), "layer0_state": {},
                         "layer1_state": {"hx": rng.randn(H).astype(np.float32), "cx": rng.randn(H).astype(np.float32),
                                          "initials": np.float32(rng.rand() < 0.002)},
                         "layer2_state": {}},
                        float(rng.choice([-1.0, 0.0, 1.0], p=[.1, .8, .1])), bool(rng.rand() < 0.002), {}, e))
                self.s += 1
            return out

    class Quiet:
        def log_result(self, *a, **k):
            pass

        def save_checkpoint(self, *a, **k):
            pass

    model = load_config(os.path.join("/root/reference/rltime/configs/models", "nature_cnn_lstm512_fc512.json"))
    tr = RefIQN(logger=Quiet(), actors=Scripted(), model_config=model,
                policy_args={"dueling": True, "embedding_dim": 64, "num_sampling_quantiles": 32, "cuda": False})
    spans, stack = [], {}
    for name in ("_burn_in", "calc_target_values", "train_batch"):
        inner = getattr(tr, name)

        def timed(*a, _inner=inner, _name=name, **k):
            t = time.time()
            try:
                return _inner(*a, **k)
            finally:
                stack[_name] = stack.get(_name, 0.0) + time.time() - t
        setattr(tr, name, timed)
    real_init = tr._init_history_buffer

    def init_and_wrap(*a, **k):
        real_init(*a, **k)
        g = tr.history_buffer.get_train_data

        def timed_get(*aa, **kk):
            t = time.time()
            out = g(*aa, **kk)
            if out is not None:
                if "train_batch" in stack:             # the previous batch went through a learner step
                    spans.append(dict(stack))
                stack.clear()
                stack["get_train_data"] = time.time() - t
            return out
        tr.history_buffer.get_train_data = timed_get
    tr._init_history_buffer = init_and_wrap
    # fill 400 vector steps (warm-up), then STEPS + 1 learner steps (the first one is the untimed warm-up)
    acted_per_step = B * 80 // 4
    total = E * 400 + (STEPS + 1) * acted_per_step + E
    tr.train(total_steps=total, log_freq=10 ** 9, target_update_freq=10 ** 9, clip_rewards=True,
             double_q=True, clip_grad=40.0, adam_epsilon=1e-5, gamma=0.99, nstep_train=80, burn_in_timesteps=40,
             nstep_target=2, mbatch_size=B, lr=3e-4, rnn_bootstrap=True, warmup_steps=E * 400,
             history_mode={"type": "prioritized_replay", "args": {
                 "size": E * 500, "train_frequency": 4, "alpha": 0.9, "beta": 0.6, "max_weight_factor": 0.9}})
    if "train_batch" in stack:
        spans.append(dict(stack))
    return spans[1:]                          # drop the first (allocator / MKL warm-up) step


if __name__ == "__main__":
    random.seed(0); np.random.seed(0); torch.manual_seed(0)          # noqa: E702
    torch.set_num_threads(1)
    ref = reference_side()
    torch.set_num_threads(1)
    ora = oracle_side()
    ref_tot = [sum(x.values()) for x in ref]
    best = ref[int(np.argmin(ref_tot))]
    out = {"B": B, "T": 80, "burn_in": 40, "nstep_target": 2, "threads": 1, "host_cores": os.cpu_count(),
           "reference_s_per_learner_step": [round(x, 3) for x in ref_tot],
           "reference_phases_of_fastest_step": {k: round(v, 3) for k, v in best.items()},
           "oracle_s_per_learner_step": [round(x, 3) for x in ora],
           "reference_min": min(ref_tot), "oracle_min": min(ora), "oracle_over_reference": min(ora) / min(ref_tot)}
    print(json.dumps(out))
