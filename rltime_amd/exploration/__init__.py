"""Exploration plugins (registry group "exploration")."""
from rltime_amd.general.lazy_types import LazyTypes

_TABLE = LazyTypes({
    "epsilon_greedy": "rltime_amd.exploration.epsilon_greedy:EpsilonGreedyExplorationManager",
})


def get_types():
    return _TABLE
