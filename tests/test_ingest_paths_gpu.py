"""GPU: the two ingest paths of History.update write the same replay — the
reference-style list of per-env sample dicts (regrouped and uploaded) and the
device-resident vector-step tensors (acting/actor.py DeviceSamples / update_batch)."""
import random

import numpy as np
import pytest
import torch

from tests import scenario
from tests.golden.streams import StreamSpec, vector_steps, as_reference_samples

pytestmark = pytest.mark.gpu


def test_dict_and_batched_ingest_are_identical():
    from rltime_amd.history import PrioritizedReplayHistoryBuffer
    spec = StreamSpec(seed=41, num_envs=6, frame_shape=(4, 21, 21), lstm_units=16, n_actions=5, done_prob=0.05,
                      fractional_rewards=False, extra_features=8)     # 32 B extra rows: the 16 B/lane path
    hist = dict(size=300, train_frequency=4, nstep_target=2, nstep_train=6, prefix_steps=3, alpha=0.8, beta=0.5)
    a = PrioritizedReplayHistoryBuffer(**hist, gamma=0.99)
    b = PrioritizedReplayHistoryBuffer(**hist, gamma=0.99)
    first = True
    for st in vector_steps(spec, 80):
        samples = as_reference_samples(spec, st)
        a.update(samples)
        if first:
            b.configure(samples[0]["next_state"], spec.num_envs, 0, policy_f32=spec.n_actions)
            first = False
        dev = b.device
        b.update_batch(
            torch.from_numpy(st["frames"]).to(dev), torch.from_numpy(st["actions"].astype(np.int32)).to(dev),
            torch.from_numpy(st["rewards"].astype(np.float32)).to(dev),
            torch.from_numpy(st["dones"].astype(np.uint8)).to(dev),
            extra=torch.from_numpy(st["extra"]).to(dev),
            state=torch.from_numpy(np.concatenate([st["hx"], st["cx"]], axis=1)).to(dev),
            initials=torch.from_numpy(st["initials"]).to(dev),
            policy=torch.from_numpy(st["qvalues"]).to(dev))
    random.seed(3)
    x = a.get_train_data(7, 0.4)
    random.seed(3)
    y = b.get_train_data(7, 0.4)
    fx = {k: scenario.to_numpy(v) for k, v in scenario.flatten("", x, {}).items()}
    fy = {k: scenario.to_numpy(v) for k, v in scenario.flatten("", y, {}).items()}
    assert set(fx) == set(fy)
    for k in fx:
        assert np.array_equal(fx[k], fy[k]), k
    va, ka, _ = a.tree_nodes()
    vb, kb, _ = b.tree_nodes()
    assert np.array_equal(va, vb) and np.array_equal(ka, kb)
    a.close(); b.close()
