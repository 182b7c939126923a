"""rltime_amd — MI355X-native backend for rltime's Q-learning hot path
(replay ingest -> uniform / sum-tree sampling -> n-step sequence gather ->
double-Q / IQN targets -> loss -> priority update), behind the reference's
history / trainer plugin API.  See DESIGN.md."""
__version__ = "0.1.0"
