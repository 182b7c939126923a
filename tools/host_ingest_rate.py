#!/usr/bin/env python3
"""PCIe-inclusive ingest rate of the reference-compatible path: History.update(list of
per-env sample dicts with host numpy payloads) -> stack -> pageable H2D -> mirl_replay_ingest."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rltime_amd.history import PrioritizedReplayHistoryBuffer  # noqa: E402

E, H = 256, 512
rng = np.random.RandomState(0)
pool = [rng.randint(0, 256, (4, 84, 84)).astype(np.uint8) for _ in range(64)]
buf = PrioritizedReplayHistoryBuffer(size=200000, train_frequency=4, nstep_target=2, nstep_train=80, prefix_steps=40,
                                     gamma=0.99, keep_policy_outputs=False)


def step(s):
    return [{"policy_output": {"actions": 1}, "reward": 0.0, "done": False, "info": {}, "env_id": e,
             "next_state": {"x": pool[(s + e) % 64], "layer0_state": {},
                            "layer1_state": {"hx": np.zeros(H, np.float32), "cx": np.zeros(H, np.float32),
                                             "initials": np.float32(0)}, "layer2_state": {}}} for e in range(E)]


for s in range(3):
    buf.update(step(s))
torch.cuda.synchronize()
t0 = time.time()
n = 20
for s in range(n):
    buf.update(step(s))
torch.cuda.synchronize()
dt = time.time() - t0
print(json.dumps({"transitions_per_s": n * E / dt, "ms_per_256_env_vector_step": dt / n * 1e3,
                  "MB_per_s_incl_PCIe": n * E * (28224 + 4096) / dt / 1e6}))
