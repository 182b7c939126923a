"""Plugin registry: (group, name) -> class.  Groups and names are the ones the
reference's json configs use (trainers, models, modules, history, exploration);
anything that is not a string is returned unchanged, so configs may also carry
classes directly."""
import importlib

_GROUP_MODULES = {
    "trainers": "rltime_amd.training",
    "models": "rltime_amd.models",
    "modules": "rltime_amd.models.torch.modules",
    "history": "rltime_amd.history",
    "exploration": "rltime_amd.exploration",
}
_cache = {}


def _group(group):
    if group not in _GROUP_MODULES:
        raise TypeError("No types registered for group '%s'" % group)
    if group not in _cache:
        _cache[group] = importlib.import_module(_GROUP_MODULES[group]).get_types()
    return _cache[group]


def get_registered_type(group, ref):
    if not isinstance(ref, str):
        return ref
    table = _group(group)
    if ref not in table:
        raise TypeError("No type '%s' registered in group '%s', available types in this group are: %s"
                        % (ref, group, list(table.keys())))
    return table[ref]
