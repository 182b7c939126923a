"""String -> class registry with the reference's groups and keys
(rltime/general/type_registry.py:6-39)."""

_registry = None


def _build():
    from rltime_amd import training, models, history, exploration
    from rltime_amd.models.torch import modules
    return {
        "trainers": training.get_types(),
        "models": models.get_types(),
        "modules": modules.get_types(),
        "history": history.get_types(),
        "exploration": exploration.get_types(),
    }


def get_registered_type(group, ref):
    global _registry
    if not isinstance(ref, str):
        return ref                      # a python class passes straight through
    if _registry is None:
        _registry = _build()
    if group not in _registry:
        raise TypeError("No types registered for group '%s'" % group)
    if ref not in _registry[group]:
        raise TypeError(
            "No type '%s' registered in group '%s', available types in this "
            "group are: %s" % (ref, group, list(_registry[group].keys())))
    return _registry[group][ref]
