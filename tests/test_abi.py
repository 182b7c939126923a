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
        # bench.py may use the oracle only inside its cpu_baseline leg (class CpuPath + cpu_baseline())
        head, _, tail = src.partition("class CpuPath")
        body, _, rest = tail.partition("\ndef main")
        assert "from oracle" not in head and "from oracle" not in rest


def test_history_requires_gpu():
    if _lib.device_count() > 0:
        return
    import pytest
    from rltime_amd.history import ReplayHistoryBuffer
    with pytest.raises(_lib.MirlError, match="no CPU fallback"):
        ReplayHistoryBuffer(size=8, train_frequency=1, nstep_target=1, nstep_train=1)


# --------------------------------------------------------------------------
# plugin surface: the mirror classes against the reference's signatures
# (tests/golden/signatures.json, written by tests/golden/generate.py from the
# imported reference with inspect.signature)
# --------------------------------------------------------------------------
_MIRROR = {
    "history.History": ("rltime_amd.history.replay_history", "History"),
    "history.ReplayHistoryBuffer": ("rltime_amd.history.replay_history", "ReplayHistoryBuffer"),
    "history.PrioritizedReplayHistoryBuffer": ("rltime_amd.history.replay_history", "PrioritizedReplayHistoryBuffer"),
    "policies.Policy": ("rltime_amd.policies.policy", "Policy"),
    "policies.TorchPolicy": ("rltime_amd.policies.torch_policy", "TorchPolicy"),
    "policies.DQNPolicy": ("rltime_amd.policies.dqn", "DQNPolicy"),
    "policies.IQNPolicy": ("rltime_amd.policies.iqn", "IQNPolicy"),
    "training.PolicyTrainer": ("rltime_amd.training.policy_trainer", "PolicyTrainer"),
    "training.MultiStepTrainer": ("rltime_amd.training.multi_step_trainer", "MultiStepTrainer"),
    "training.TorchTrainer": ("rltime_amd.training.torch_trainer", "TorchTrainer"),
    "training.DQN": ("rltime_amd.training.dqn", "DQN"),
    "acting.ActingInterface": ("rltime_amd.acting.acting_interface", "ActingInterface"),
    "acting.Actor": ("rltime_amd.acting.actor", "Actor"),
    "models.SequentialModel": ("rltime_amd.models.torch.sequential", "SequentialModel"),
    "history.OnlineHistoryBuffer": ("rltime_amd.history.online_history", "OnlineHistoryBuffer"),
    "policies.ActorCriticPolicy": ("rltime_amd.policies.actor_critic", "ActorCriticPolicy"),
    "training.A2C": ("rltime_amd.training.a2c", "A2C"),
    "training.PPO": ("rltime_amd.training.ppo", "PPO"),
}
# where the mirror deliberately differs from the reference's parameter list
_KNOWN = {
    # the device replay evaluates the n-step return on the GPU and takes the discount as a number too
    ("history.History", "__init__"): "appends gamma",
    # History methods live on the concrete device buffers (History itself only normalises arguments)
    ("history.History", "update"): "on ReplayHistoryBuffer",
    ("history.History", "needed_feed_count"): "on ReplayHistoryBuffer",
    ("history.History", "get_train_data"): "on ReplayHistoryBuffer",
    ("history.History", "update_losses"): "on ReplayHistoryBuffer",
}


def _params(fn):
    import inspect
    out = []
    for name, p in inspect.signature(fn).parameters.items():
        kind = {p.VAR_POSITIONAL: "*", p.VAR_KEYWORD: "**"}.get(p.kind, "")
        out.append([kind + name, p.default is not p.empty])
    return out


def test_mirror_classes_keep_the_reference_signatures():
    import importlib
    import json
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "signatures.json")))
    problems = []
    for key, methods in want.items():
        module, cls_name = _MIRROR[key]
        cls = getattr(importlib.import_module(module), cls_name)
        for m, ref in methods.items():
            owner = cls
            if (key, m) in _KNOWN and _KNOWN[(key, m)].startswith("on "):
                owner = getattr(importlib.import_module(module), _KNOWN[(key, m)][3:])
            if not hasattr(owner, m):
                problems.append("%s.%s is missing" % (key, m))
                continue
            got = _params(getattr(owner, m))
            ref_named = [p for p in ref if not p[0].startswith("*")]
            got_named = [p for p in got if not p[0].startswith("*")]
            # same names, same order, same optionality; the mirror may append OPTIONAL parameters
            if got_named[:len(ref_named)] != ref_named:
                problems.append("%s.%s: reference %s, mirror %s" % (key, m, ref_named, got_named))
            elif any(not opt for _, opt in got_named[len(ref_named):]):
                problems.append("%s.%s: mirror adds required parameters %s" % (key, m, got_named[len(ref_named):]))
            if any(p[0].startswith("**") for p in ref) and not any(p[0].startswith("**") for p in got):
                problems.append("%s.%s: reference forwards **kwargs, mirror does not" % (key, m))
    assert not problems, "\n".join(problems)


def test_conv_kernel_shape_gates_are_host_logic():
    """The shape gates of the hand-written conv kernels (include/mirl.h:
    mirl_conv1_u8_supported, mirl_conv2_bwd_data_supported) and the weight-gradient
    scratch size are pure host code: callable without a GPU, and they refuse every
    shape the kernels' register / LDS plans do not cover."""
    import ctypes as C
    from rltime_amd._lib import lib
    assert lib.mirl_conv1_u8_supported(4, 84, 84, 32, 8, 4) == 1          # the Atari input layer
    assert lib.mirl_conv1_u8_supported(4, 8, 8, 32, 8, 4) == 1            # a single output position
    assert lib.mirl_conv1_u8_supported(4, 100, 100, 32, 8, 4) == 1        # 4 padded planes = 40 KB of LDS
    for bad in [(1, 84, 84, 32, 8, 4), (4, 84, 84, 64, 8, 4), (4, 84, 84, 32, 4, 4), (4, 84, 84, 32, 8, 2),
                (4, 84, 82, 32, 8, 4), (4, 7, 84, 32, 8, 4), (4, 42, 42, 32, 8, 4), (4, 128, 132, 32, 8, 4)]:
        assert lib.mirl_conv1_u8_supported(*bad) == 0, bad
    assert lib.mirl_conv2_bwd_data_supported(32, 64, 4, 2, 20, 20, 9, 9) == 1
    assert lib.mirl_conv2_bwd_data_supported(32, 64, 4, 2, 4, 4, 1, 1) == 1
    for bad in [(32, 64, 4, 2, 21, 20, 9, 9), (32, 64, 4, 2, 20, 20, 9, 10), (64, 64, 4, 2, 20, 20, 9, 9),
                (32, 64, 3, 2, 20, 20, 9, 9), (32, 64, 4, 1, 20, 20, 9, 9), (32, 64, 4, 2, 2, 2, 0, 0),
                (32, 64, 4, 2, 130, 130, 64, 64)]:
        assert lib.mirl_conv2_bwd_data_supported(*bad) == 0, bad
    need = C.c_int64()
    assert lib.mirl_conv1_u8_wrw_scratch_floats(C.byref(need)) == 0 and need.value == 512 * (32 * 4 * 8 * 8 + 32)     # 512 slabs of 8192 weight-gradient + 32 bias-gradient sums
    assert lib.mirl_conv1_u8_wrw_scratch_floats(None) < 0                 # null out pointer: error code, no crash


def test_acting_network_shape_gates_are_host_logic():
    """The shape gates and workspace sizes of the acting network's kernels (include/mirl.h: mirl_act_*_supported,
    mirl_act_head_parts, mirl_act_lstm_workspace_bytes) are pure host code: callable without a GPU; launches with bad
    arguments are refused before anything touches a device."""
    import ctypes as C
    from rltime_amd._lib import lib
    assert lib.mirl_act_conv_supported(2, 32, 64, 4, 2, 20, 20) == 1 and lib.mirl_act_conv_supported(3, 64, 64, 3, 1, 9, 9) == 1
    for bad in [(2, 64, 64, 4, 2, 20, 20), (2, 32, 32, 4, 2, 20, 20), (3, 64, 64, 3, 2, 9, 9), (1, 4, 32, 8, 4, 84, 84), (3, 64, 64, 3, 1, 2, 9)]:
        assert lib.mirl_act_conv_supported(*bad) == 0, bad
    assert lib.mirl_act_lstm_supported(32, 512, 3648) == 1 and lib.mirl_act_lstm_supported(64, 64, 3200) == 1
    for bad in [(65, 512, 3648), (32, 508, 3648), (32, 512, 3640), (0, 512, 3648)]:
        assert lib.mirl_act_lstm_supported(*bad) == 0, bad
    need = C.c_int64()
    assert lib.mirl_act_lstm_workspace_bytes(32, 512, 3648, C.byref(need)) == 0
    assert need.value >= 4 * 4 * 64 * 32 * 32 + 4 * 64                       # 4 K slices x 64 column blocks x E x 32 floats + the arrival counters
    assert lib.mirl_act_lstm_workspace_bytes(65, 512, 3648, C.byref(need)) < 0
    assert lib.mirl_act_head_supported(32, 32, 512, 64, 1024, 7) == 1 and lib.mirl_act_head_supported(256, 1, 64, 0, 128, 19) == 1
    for bad in [(32, 32, 500, 64, 1024, 7), (32, 32, 512, 80, 1024, 7), (32, 32, 512, 64, 1000, 7), (32, 32, 512, 64, 1024, 33),
                (0, 32, 512, 64, 1024, 7), (32, 32, 192, 64, 1024, 7)]:
        assert lib.mirl_act_head_supported(*bad) == 0, bad
    parts, pitch = C.c_int32(), C.c_int32()
    assert lib.mirl_act_head_parts(1024, 7, C.byref(parts), C.byref(pitch)) == 0 and (parts.value, pitch.value) == (16, 8)
    assert lib.mirl_act_head_parts(272, 19, C.byref(parts), C.byref(pitch)) == 0 and (parts.value, pitch.value) == (5, 24)
    assert lib.mirl_act_head_parts(0, 7, C.byref(parts), C.byref(pitch)) < 0
    # refused before any launch: null operands / unsupported shapes
    assert lib.mirl_act_conv_fwd(4, 1, 20, 20, None, None, None, None, 0, None) < 0
    assert lib.mirl_act_lstm_fwd(32, 512, 3648, None, 3648, None, None, None, None, None, None, None) < 0
    assert lib.mirl_act_head_hidden(1024, 500, 1024, 7, None, None, None, None, None, None) < 0
    assert lib.mirl_act_head_select(32, 32, 6, 16, 8, None, None, 1, None, None, 0.0, 0, None, None, None, None) < 0


def test_rollout_plan_refuses_before_bookkeeping():
    """mirl_replay_ingest_plan (csrc/replay.hip): argument errors are MIRL_ERR_ARG and carry the library's code to Python
    (MirlError.code), which is what FastActingStep.rollout's fallback keys on."""
    from rltime_amd import _lib
    assert _lib.MIRL_ERR_ARG == -1 and _lib.MIRL_ERR_STATE == -3
    rc = _lib.lib.mirl_replay_ingest_plan(None, 4, 8, None, None)
    assert rc == _lib.MIRL_ERR_ARG
    try:
        _lib.check(rc, "mirl_replay_ingest_plan")
        raise AssertionError("check() must raise")
    except _lib.MirlError as e:
        assert e.code == _lib.MIRL_ERR_ARG
