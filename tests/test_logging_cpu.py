"""CPU: the aggregating value log and the directory logger (how
steps_trained_per_second etc. reach train.json; reference
rltime/general/{value_log,loggers}.py) — grouping, aggregation scopes, row and
checkpoint files."""
import json
import os
import pickle

from rltime_amd.general.loggers import DirectoryLogger
from rltime_amd.general.value_log import ValueLog


def test_value_log_groups_scopes_and_aggregations():
    v = ValueLog()
    for x in (1.0, 2.0, 6.0):
        v.log("loss", x, group="train")                      # interval mean
        v.log("loss_max", x, agg="max", group="train")
        v.log("steps", x, agg="sum", group="total", scope=None)   # never reset
        v.log("reward", x, scope=2, group="last2", precision=1)    # sliding window of 2
    v.log("eps", 0.25, group="acting->exploration")              # nested group
    row = v.get()
    assert row["train"] == {"loss": 3.0, "loss_max": 6.0}
    assert row["total"]["steps"] == 9.0
    assert row["last2"]["reward"] == 4.0
    assert row["acting"]["exploration"]["eps"] == 0.25
    v.log("loss", 10.0, group="train")
    v.log("steps", 1.0, agg="sum", group="total", scope=None)
    row = v.get()
    assert row["train"]["loss"] == 10.0                          # interval entries were reset
    assert row["total"]["steps"] == 10.0                         # scope=None keeps accumulating
    assert row["last2"]["reward"] == 4.0                         # window persists
    assert "loss_max" not in row["train"]                        # nothing logged this interval


def test_directory_logger_files(tmp_path):
    lg = DirectoryLogger.create_new(str(tmp_path), "run")
    lg2 = DirectoryLogger.create_new(str(tmp_path), "run")       # uniquified, never overwrites
    assert lg.path != lg2.path
    lg.echo = False
    lg.log_config({"training": {"type": "iqn"}})
    lg.log_result("train", {"this_interval": {"steps_trained_per_second": 123}}, 1000)
    lg.log_result("train", {"this_interval": {"steps_trained_per_second": 456}}, 2000)
    lg.save_checkpoint({"policy_state": b"abc", "train_state": {}}, 2000)
    assert json.load(open(os.path.join(lg.path, "config.json")))["training"]["type"] == "iqn"
    rows = [json.loads(line) for line in open(os.path.join(lg.path, "train.json"))]
    assert [r["step"] for r in rows] == [1000, 2000]
    assert rows[1]["this_interval"]["steps_trained_per_second"] == 456
    ck = pickle.load(open(os.path.join(lg.path, "checkpoint.p"), "rb"))
    assert ck["step"] == 2000 and ck["data"]["policy_state"] == b"abc"       # reference layout (loggers.py:176-192)
