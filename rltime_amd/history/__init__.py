"""History (replay) plugins, same registry keys as the reference
(rltime/history/__init__.py:6-11)."""
from .replay_history import ReplayHistoryBuffer, PrioritizedReplayHistoryBuffer


def get_types():
    from .online_history import OnlineHistoryBuffer
    return {
        "online": OnlineHistoryBuffer,
        "replay": ReplayHistoryBuffer,
        "prioritized_replay": PrioritizedReplayHistoryBuffer,
    }
