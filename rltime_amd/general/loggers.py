"""Result loggers (reference rltime/general/loggers.py:99-192): a directory
logger writing config.json, one JSON line per log interval to <name>.json and
checkpoint.p (pickle of {step, data}); a console/null logger for benches."""
import json
import os
import pickle
import time


class NullLogger:
    def __init__(self, echo=False, path=None):
        self.echo = echo
        self.rows = []
        if path is not None:        # a rank that does not log still knows the run directory (per-rank checkpoints)
            self.path = path

    def log_config(self, config):
        pass

    def log_result(self, name, data, step):
        self.rows.append((name, step, data))
        if self.echo:
            print("[%s] step %s: %s" % (name, step, json.dumps(data, default=str)))

    def save_checkpoint(self, data, step):
        pass


class DirectoryLogger(NullLogger):
    def __init__(self, path, echo=True):
        super().__init__(echo)
        self.path = path
        os.makedirs(path, exist_ok=True)

    @classmethod
    def create_new(cls, base_dir, name=None):
        """loggers.py:118-131: <base>/<name or timestamp>, uniquified."""
        name = name or time.strftime("%Y%m%d_%H%M%S")
        path, k = os.path.join(base_dir, name), 1
        while os.path.exists(path):
            k += 1
            path = os.path.join(base_dir, "%s_%d" % (name, k))
        return cls(path)

    def log_config(self, config):
        with open(os.path.join(self.path, "config.json"), "w") as f:
            json.dump(config, f, indent=2, default=str)

    def log_result(self, name, data, step):
        super().log_result(name, data, step)
        with open(os.path.join(self.path, name + ".json"), "a") as f:
            f.write(json.dumps({"step": step, **data}, default=str) + "\n")

    def save_checkpoint(self, data, step):
        with open(os.path.join(self.path, "checkpoint.p"), "wb") as f:
            pickle.dump({"step": step, "data": data}, f)
