"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/mirl.h declares (no compute calls here)."""
import os
import re

from rltime_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mirl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mirl_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    names = header_symbols()
    assert len(names) >= 35
    for name in names:
        assert hasattr(_lib.lib, name), name


def test_binding_covers_header():
    declared = set(header_symbols()) - {"mirl_last_error"}
    assert declared == set(_lib._SIGNATURES), declared ^ set(_lib._SIGNATURES)


def test_create_without_gpu_fails_loudly():
    if _lib.device_count() > 0:
        return
    import ctypes as C
    cfg = _lib.ReplayConfig(size=16, num_envs=2, frame_bytes=16, nstep_train=1,
                            nstep_target=1, gamma=0.99)
    h = C.c_void_p()
    rc = _lib.lib.mirl_replay_create(C.byref(cfg), C.byref(h))
    assert rc < 0 and "no HIP device" in _lib.last_error()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under rltime_amd/ may import it,
    and the package has no CPU fallback for the hot path."""
    import re
    pkg = os.path.join(ROOT, "rltime_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(d, f)
    for f in ("bench.py",):
        src = open(os.path.join(ROOT, f)).read()
        # bench.py may use the oracle only inside cpu_baseline()
        head, _, tail = src.partition("def cpu_baseline")
        body, _, rest = tail.partition("\ndef main")
        assert "from oracle" not in head and "from oracle" not in rest


def test_history_requires_gpu():
    if _lib.device_count() > 0:
        return
    import pytest
    from rltime_amd.history import ReplayHistoryBuffer
    with pytest.raises(_lib.MirlError, match="no CPU fallback"):
        ReplayHistoryBuffer(size=8, train_frequency=1, nstep_target=1, nstep_train=1)
