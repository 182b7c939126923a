"""CPU: epsilon-greedy remapping (same np.random consumption order as the
reference, epsilon_greedy.py:64-99) and the linear anneal (utils.py:85-103)
against golden vectors from the unmodified reference."""
import json
import os

import numpy as np

from rltime_amd.exploration.epsilon_greedy import EpsilonGreedyExplorationManager
from rltime_amd.general.utils import anneal_value
from rltime_amd.spaces import Discrete
from tests import scenario


def test_epsilon_greedy_matches_reference():
    d = np.load(os.path.join(scenario.GOLDEN, "schedule_cases.npz"))
    cases = json.loads(str(d["cases"]))
    E = 16
    for name, kw in cases.items():
        m = EpsilonGreedyExplorationManager(**kw, total_actors=E)
        for pi, progress in enumerate((0.0, 0.03, 0.25, 0.5, 0.9, 1.0)):
            np.random.seed(100 + pi)
            acts, info = m.remap_actions(np.arange(E) % 6, list(range(E)), Discrete(6), progress)
            assert np.array_equal(np.array(acts), d["%s.p%d.actions" % (name, pi)]), (name, pi)
            assert np.array_equal(info["eps"], d["%s.p%d.eps" % (name, pi)]), (name, pi)


def test_anneal_matches_reference():
    d = np.load(os.path.join(scenario.GOLDEN, "schedule_cases.npz"))
    grid = d["anneal.progress"]
    assert [anneal_value(0.4, p, True, 1.0) for p in grid] == list(d["anneal.true"])
    assert [anneal_value(3e-4, p, 1e-5) for p in grid] == list(d["anneal.to"])
    assert [anneal_value(0.6, p, False) for p in grid] == list(d["anneal.off"])


def test_device_epsilon_matches_host_formula():
    """remap_actions_device computes the same per-actor epsilons as the
    reference's per-actor loop (epsilon_greedy.py:80-88)."""
    import torch
    d = np.load(os.path.join(scenario.GOLDEN, "schedule_cases.npz"))
    cases = json.loads(str(d["cases"]))
    E = 16
    for name, kw in cases.items():
        m = EpsilonGreedyExplorationManager(**kw, total_actors=E)
        for pi, progress in enumerate((0.0, 0.03, 0.25, 0.5, 0.9, 1.0)):
            np.random.seed(100 + pi)          # _get_eps consumes the same np.random draw
            _, info = m.remap_actions_device(torch.arange(E) % 6, list(range(E)), Discrete(6), progress)
            np.testing.assert_allclose(info["eps"].numpy(), d["%s.p%d.eps" % (name, pi)], rtol=1e-6)
