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
