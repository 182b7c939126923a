"""Trainer plugins, registry keys as in the reference
(rltime/training/torch/__init__.py:8-15).  Only the Q-learning trainers are on
the MI355X hot path; a2c / ppo / dist_dqn are out of scope."""


def get_types():
    from .dqn import DQN
    from .iqn import IQN
    return {"dqn": DQN, "iqn": IQN}
