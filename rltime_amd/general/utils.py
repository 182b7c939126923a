"""Pytree and schedule helpers (reference rltime/general/utils.py:8-103)."""
import importlib

import numpy as np

# top-level modules "@python('a.b.c')" config references may import
# (reference general/allowed_modules.py:12 + this package)
ALLOWED_MODULES = ["rltime_amd", "rltime", "gym", "retro", "gym_ple"]


def import_by_full_name(full_name):
    """general/utils.py:8-22."""
    parts = full_name.split(".")
    assert parts[0] in ALLOWED_MODULES, \
        "Can't import %s by string: '%s' is not in ALLOWED_MODULES %s" % (
            full_name, parts[0], ALLOWED_MODULES)
    return getattr(importlib.import_module(".".join(parts[:-1])), parts[-1])


def deep_apply(x, f):
    """general/utils.py:56-68."""
    if isinstance(x, (list, tuple)):
        return type(x)(deep_apply(v, f) for v in x)
    if isinstance(x, dict):
        return {k: deep_apply(v, f) for k, v in x.items()}
    if x is None:
        return None
    return f(x)


def deep_stack(x, op=np.stack, args=None, base_type=np.ndarray):
    """general/utils.py:25-53."""
    args = args or {}
    head = x[0]
    if isinstance(head, base_type):
        return op(x, **args)
    if isinstance(head, (list, tuple)):
        return type(head)(
            deep_stack([it[i] for it in x], op, args, base_type) for i in range(len(head)))
    if isinstance(head, dict):
        return {k: deep_stack([it[k] for it in x], op, args, base_type) for k in head}
    if head is None:
        return None
    return op(list(x), **args)


def deep_dictionary_update(dest, source):
    """general/utils.py:71-82 (in place)."""
    assert isinstance(dest, dict) and isinstance(source, dict)
    for key, val in source.items():
        if isinstance(val, dict):
            deep_dictionary_update(dest.setdefault(key, {}), val)
        else:
            dest[key] = val


def anneal_value(base_value, progress, anneal_mode, default_target=0.0):
    """general/utils.py:85-103."""
    assert progress >= 0
    progress = min(progress, 1.0)
    if anneal_mode is False or anneal_mode is None:
        return base_value
    target = default_target if anneal_mode is True else float(anneal_mode)
    return base_value + (target - base_value) * progress


class quiet_gc:
    """Context manager around a HIP graph capture: collect garbage BEFORE it and keep the cyclic collector off
    while the stream is capturing.  A collection that fires inside a capture can finalise objects of earlier work
    (HIP graphs, events, cached allocator blocks of a dead trainer) whose destructors issue HIP calls that are illegal
    on a capturing stream — the process aborts.  torch.cuda.graph itself stopped collecting on entry (torch >= 2.4:
    only with torch.compiler.config.force_cudagraph_gc)."""

    def __enter__(self):
        import gc
        gc.collect()
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False
