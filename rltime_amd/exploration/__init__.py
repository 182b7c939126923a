"""Exploration plugins (reference rltime/exploration/__init__.py)."""


def get_types():
    from .epsilon_greedy import EpsilonGreedyExplorationManager
    return {"epsilon_greedy": EpsilonGreedyExplorationManager}
