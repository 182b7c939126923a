"""JSON config loader with the reference's extensions
(rltime/general/config.py:16-117, config_template.py:5-42):

  "@json('file','a->b')"   nested json (optionally a nested key)
  "@python('pkg.mod.attr')" python object by name (allow-listed top modules)
  keys starting with "_"    comments
  "**": {...}               shallow merge into the enclosing object
  "***": {...}              deep merge into the enclosing object
Relative files fall back to this package's configs/ directory."""
import json
import os
import re

from .utils import import_by_full_name, deep_dictionary_update


class ConfigException(Exception):
    pass


_PKG_CONFIGS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")


def resolve_file_path(path, base_dir=None):
    for cand in ([os.path.join(base_dir, path)] if base_dir else []) + [path, os.path.join(_PKG_CONFIGS, path)]:
        if os.path.isfile(cand):
            return cand
    return path


def parse_ref(ref):
    m = re.match(r"^@(\w+)\s*\((.*)\)\s*$", ref, flags=re.S)
    if not m:
        raise ConfigException("Invalid config ref: %s" % ref)
    return m.group(1), re.findall(r"'([^']*)'", m.group(2))


def _load_json_ref(base_dir, json_file, nested_key=None):
    res = load_config(resolve_file_path(json_file, base_dir))
    if nested_key is not None:
        try:
            for key in nested_key.split("->"):
                res = res[key]
        except KeyError:
            raise ConfigException("Could not find nested key '%s' in '%s'" % (nested_key, json_file))
    return res


def _resolve(val, base_dir):
    if isinstance(val, str) and val[:1] == "@":
        kind, args = parse_ref(val)
        if kind == "json":
            try:
                return _load_json_ref(base_dir, *args)
            except FileNotFoundError:
                raise ConfigException("Could not load referenced file '%s' from '%s'" % (val, base_dir))
        if kind == "python":
            return import_by_full_name(*args)
        raise ConfigException("Unknown reference type: %s" % kind)
    if isinstance(val, list):
        return [_resolve(v, base_dir) for v in val]
    return val


def _pairs(obj, base_dir):
    res = {}
    for key, val in obj:
        if key[:1] == "_":
            continue
        val = _resolve(val, base_dir)
        if key == "**":
            res = {**res, **val}
        elif key == "***":
            deep_dictionary_update(res, val)
        else:
            res[key] = val
    return res


def load_config(file_path):
    file_path = resolve_file_path(file_path)
    base_dir = os.path.dirname(os.path.abspath(file_path))
    with open(file_path, "r") as f:
        return json.load(f, object_pairs_hook=lambda o: _pairs(o, base_dir))


CONFIG_TEMPLATE = {
    "acting": {"actor_envs": None, "actor_cls": None, "exploration": None, "extra_args": None,
               "pool": {"type": None, "args": None}},
    "env": None, "env_args": None,
    "model": {"type": None, "args": None},
    "policy_args": None,
    "training": {"type": None, "args": None},
}


def validate_config(config, template=None, path=""):
    """config_template.py:30-42: only known keys, recursively where the
    template nests."""
    template = CONFIG_TEMPLATE if template is None else template
    for key, val in config.items():
        if key not in template:
            raise ConfigException("Invalid config key '%s%s' (allowed: %s)" % (path, key, list(template)))
        if isinstance(template[key], dict) and isinstance(val, dict):
            validate_config(val, template[key], path + key + "->")
