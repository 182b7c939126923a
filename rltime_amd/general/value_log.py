"""Aggregating key/value log (reference rltime/general/value_log.py:4-105):
values are grouped ("a->b" nests), aggregated by mean/sum/max/min over a scope
(None = forever, "interval" = until the next get(), int = sliding window)."""
from collections import deque


class _Entry:
    def __init__(self, agg, scope, precision):
        self.agg, self.scope, self.precision = agg, scope, precision
        self.values = deque(maxlen=scope) if isinstance(scope, int) else []
        self.total, self.count = 0.0, 0
        self.best = None

    def add(self, v):
        if hasattr(v, "detach") and getattr(v, "is_cuda", False):
            # device scalar: accumulate on the device, resolve at get() time
            v = v.detach().float()
            if self.agg in ("mean", "sum") and not isinstance(self.scope, int):
                self.dev_total = v if getattr(self, "dev_total", None) is None else self.dev_total + v
                self.count += 1
                return
            v = v.item()
        v = float(v)
        if isinstance(self.scope, int):
            self.values.append(v)
            return
        self.total += v
        self.count += 1
        if self.best is None:
            self.best = v
        elif self.agg == "max":
            self.best = max(self.best, v)
        elif self.agg == "min":
            self.best = min(self.best, v)

    def value(self):
        if isinstance(self.scope, int):
            if not self.values:
                return None
            vals = list(self.values)
            r = {"mean": sum(vals) / len(vals), "sum": sum(vals), "max": max(vals), "min": min(vals)}[self.agg]
        else:
            if not self.count:
                return None
            if getattr(self, "dev_total", None) is not None:
                self.total += float(self.dev_total.item())
                self.dev_total = None
            r = {"mean": self.total / self.count, "sum": self.total, "max": self.best, "min": self.best}[self.agg]
        return round(r, self.precision) if self.precision is not None else r

    def reset_interval(self):
        if self.scope == "interval":
            self.total, self.count, self.best = 0.0, 0, None
            self.dev_total = None


class ValueLog:
    def __init__(self):
        self._entries = {}

    def log(self, key, value, agg="mean", group=None, scope="interval", precision=None):
        k = (group, key)
        e = self._entries.get(k)
        if e is None:
            e = self._entries[k] = _Entry(agg, scope, precision)
        e.add(value)

    def log_dict(self, values, group=None, **kw):
        for k, v in values.items():
            try:
                import numpy as np
                v = float(np.mean(v))
            except Exception:
                continue
            self.log(k, v, group=group, **kw)

    def get(self):
        out = {}
        for (group, key), e in self._entries.items():
            v = e.value()
            e.reset_interval()
            if v is None:
                continue
            node = out
            if group:
                for part in group.split("->"):
                    node = node.setdefault(part, {})
            node[key] = v
        return out
