"""History plugins.  Registry group "history" with the reference's key names
(online / replay / prioritized_replay); resolved lazily."""
from rltime_amd.general.lazy_types import LazyTypes

_TABLE = LazyTypes({
    "replay": "rltime_amd.history.replay_history:ReplayHistoryBuffer",
    "prioritized_replay": "rltime_amd.history.replay_history:PrioritizedReplayHistoryBuffer",
    "online": "rltime_amd.history.online_history:OnlineHistoryBuffer",
})


def get_types():
    return _TABLE


def __getattr__(name):          # `from rltime_amd.history import ReplayHistoryBuffer`
    if name in ("ReplayHistoryBuffer", "PrioritizedReplayHistoryBuffer"):
        from . import replay_history
        return getattr(replay_history, name)
    raise AttributeError(name)
