from .time_limit import TimeLimit  # noqa: F401
