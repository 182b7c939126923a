"""Trainer plugins (registry group "trainers").  The Q-learning trainers are the MI355X hot path; a2c / ppo serve the
CPU plumbing config (BASELINE configs[0], cartpole_ppo.json) in plain PyTorch; dist_dqn is out of scope."""
from rltime_amd.general.lazy_types import LazyTypes

_TABLE = LazyTypes({
    "dqn": "rltime_amd.training.dqn:DQN",
    "iqn": "rltime_amd.training.iqn:IQN",
    "a2c": "rltime_amd.training.a2c:A2C",
    "ppo": "rltime_amd.training.ppo:PPO",
})


def get_types():
    return _TABLE
