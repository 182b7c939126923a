"""Trainer plugins (registry group "trainers").  Only the Q-learning trainers
are on the MI355X hot path; a2c / ppo / dist_dqn are out of scope."""
from rltime_amd.general.lazy_types import LazyTypes

_TABLE = LazyTypes({
    "dqn": "rltime_amd.training.dqn:DQN",
    "iqn": "rltime_amd.training.iqn:IQN",
})


def get_types():
    return _TABLE
