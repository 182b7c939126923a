"""CPU: the json config loader keeps the reference's semantics
(rltime/general/config.py:32-117 — '@json' with nested key paths, '@python',
'_' comment keys, '**' shallow and '***' deep merges, refs inside lists), and
the template check / registry behave like config_template.py / type_registry.py.
Expected values were produced by the unmodified reference's load_config on the
files in tests/golden/configs (tests/golden/generate.py: run_config_cases)."""
import json
import os
import shutil

import pytest

from rltime_amd.general.config import load_config, validate_config, ConfigException
from rltime_amd.general.type_registry import get_registered_type
from tests import scenario

CFG_DIR = os.path.join(scenario.GOLDEN, "configs")


def _enc(o):
    if callable(o):
        return "<python:%s.%s>" % (o.__module__.split(".", 1)[-1], o.__name__)
    raise TypeError(o)


def test_loader_matches_reference(tmp_path):
    want = json.load(open(os.path.join(scenario.GOLDEN, "config_cases.json")))
    # same files, with the '@python' package prefix pointed at this package
    work = tmp_path / "configs"
    shutil.copytree(CFG_DIR, work)
    for root, _, files in os.walk(work):
        for f in files:
            p = os.path.join(root, f)
            s = open(p).read().replace("@python('rltime.", "@python('rltime_amd.")
            open(p, "w").write(s)
    for name, expected in want.items():
        got = json.loads(json.dumps(load_config(str(work / name)), default=_enc))
        assert got == expected, name


def test_shipped_configs_load_and_validate():
    for name in ("synthetic_atari_dqn.json", "synthetic_atari_rainbow_iqn.json", "synthetic_atari_iqn_lstm.json"):
        cfg = load_config(name)                     # resolved relative to rltime_amd/configs
        validate_config(cfg)
        assert get_registered_type("trainers", cfg["training"]["type"])
        assert cfg["model"]["type"] == "sequential"


def test_template_and_registry_errors():
    with pytest.raises(ConfigException):
        validate_config({"acting": {"actor_envs": 1, "bogus": 2}})
    with pytest.raises(ConfigException):
        validate_config({"nonsense": 1})
    with pytest.raises(TypeError):
        get_registered_type("history", "no_such_history")
    with pytest.raises(TypeError):
        get_registered_type("no_such_group", "x")
    assert get_registered_type("history", dict) is dict          # classes pass straight through
    with pytest.raises(AssertionError):
        from rltime_amd.general.utils import import_by_full_name
        import_by_full_name("os.system")                          # not on the allow-list
