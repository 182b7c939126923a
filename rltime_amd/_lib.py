"""ctypes binding of librltime_hip.so (include/mirl.h).

The library is the product: if it is missing or cannot be loaded this module
raises — there is no Python/torch fallback for any op it exports.
"""
import ctypes as C
import os

# torch must load ITS bundled HIP runtime (libamdhip64.so.7) before our library
# is dlopen'ed: both carry the same SONAME, the first one loaded wins for the
# whole process, and torch cannot see the GPU through the system copy.
import torch  # noqa: F401  (order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librltime_hip.so")

MIRL_OK = 0
MIRL_NEED_MORE = 1
MODE_UNIFORM = 0
MODE_PER = 1
INT32_MIN = -2 ** 31


MIRL_OK, MIRL_ERR_ARG, MIRL_ERR_HIP, MIRL_ERR_STATE, MIRL_ERR_NOGPU = 0, -1, -2, -3, -4     # include/mirl.h:33-38


class MirlError(RuntimeError):
    """`code` is the library's return value (MIRL_ERR_*), None when raised on the Python side."""
    code = None


class ReplayConfig(C.Structure):
    _fields_ = [
        ("size", C.c_int64),
        ("num_envs", C.c_int32),
        ("env_base", C.c_int32),
        ("frame_bytes", C.c_int32),
        ("extra_f32", C.c_int32),
        ("state_f32", C.c_int32),
        ("has_initials", C.c_int32),
        ("policy_f32", C.c_int32),
        ("nstep_train", C.c_int32),
        ("prefix_steps", C.c_int32),
        ("nstep_target", C.c_int32),
        ("gamma", C.c_double),
        ("mode", C.c_int32),
        ("train_frequency", C.c_int32),
        ("avoid_episode_crossing", C.c_int32),
        ("overlap", C.c_int32),
        ("alpha", C.c_double),
        ("beta", C.c_double),
        ("eps", C.c_double),
        ("max_weight_factor", C.c_double),
        ("beta_anneal_mode", C.c_int32),
        ("beta_anneal_to", C.c_double),
        ("global_importance_scaling", C.c_int32),
        ("env_ring_slack", C.c_int32),
        ("device", C.c_int32),
        ("acting_priority_init", C.c_int32),
        ("acting_vf_eps", C.c_double),
        ("stack_planes", C.c_int32),
    ]


class Ingest(C.Structure):
    _fields_ = [
        ("count", C.c_int32),
        ("env_ids_host", C.c_void_p),
        ("frames", C.c_void_p),
        ("extra", C.c_void_p),
        ("state", C.c_void_p),
        ("initials", C.c_void_p),
        ("actions", C.c_void_p),
        ("policy", C.c_void_p),
        ("rewards", C.c_void_p),
        ("dones", C.c_void_p),
        ("newest_plane_only", C.c_int32),
        ("frames_stride", C.c_int64),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("frames", C.c_void_p),
        ("extra", C.c_void_p),
        ("state", C.c_void_p),
        ("initials", C.c_void_p),
        ("returns", C.c_void_p),
        ("nsteps", C.c_void_p),
        ("masks", C.c_void_p),
        ("actions", C.c_void_p),
        ("policy", C.c_void_p),
        ("weights", C.c_void_p),
        ("loss_indices", C.c_void_p),
    ]


def _load():
    if not os.path.isfile(LIB_PATH):
        raise MirlError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (rltime_amd/csrc/build.sh). rltime_amd has no CPU "
            "fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.mirl_last_error.restype = C.c_char_p
    return lib


lib = _load()

_vp, _i32, _i64, _u64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
_P = C.POINTER

_SIGNATURES = {
    "mirl_device_count": [],
    "mirl_replay_create": [_P(ReplayConfig), _P(_vp)],
    "mirl_replay_destroy": [_vp],
    "mirl_replay_ingest": [_vp, _P(Ingest), _vp],
    "mirl_replay_ingest_plan": [_vp, _i32, _i32, _vp, _vp],
    "mirl_replay_ingest_planned": [_vp, _i32, _P(Ingest), _vp],
    "mirl_ingest_fused_set": [_i32],
    "mirl_replay_prime_stack": [_vp, _vp, _i64, _vp],
    "mirl_replay_needed_feed_count": [_vp, _i32, _i32, _P(_i64)],
    "mirl_replay_sample": [_vp, _i32, _f64, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mirl_replay_sample_ready": [_vp, _i32, _P(_i32)],
    "mirl_replay_sample_skip": [_vp, _i32],
    "mirl_replay_tree_root": [_vp, _vp, _vp],
    "mirl_replay_sample_global": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _f64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mirl_replay_profile": [_vp, _i32, _P(_i64), _P(_f64)],
    "mirl_profile_set": [_i32],
    "mirl_profile_collect": [_P(_i32)],
    "mirl_profile_get": [_i32, C.c_char_p, _i32, _P(_i64), _P(_f64), _P(_f64)],
    "mirl_profile_get_flop": [_i32, _vp],
    "mirl_profile_reset": [],
    "mirl_replay_save": [_vp, C.c_char_p],
    "mirl_replay_load": [_vp, C.c_char_p],
    "mirl_replay_uniform_total": [_vp, _P(_i64)],
    "mirl_replay_set_train_quota": [_vp, _i64],
    "mirl_replay_state_rows": [_vp, _P(_i32), _P(_i32)],
    "mirl_replay_gather": [_vp, _i32, _vp, _vp, _vp, _vp, _P(Batch), _vp],
    "mirl_replay_update_losses": [_vp, _i64, _vp, _vp, _vp],
    "mirl_replay_stats": [_vp, _P(_i64), _P(_i64), _P(_i64), _P(_i64), _P(_i64)],
    "mirl_replay_env_meta": [_vp, _vp, _vp],
    "mirl_replay_free_slots": [_vp, _vp, _P(_i64)],
    "mirl_replay_slot_table": [_vp, _vp, _vp],
    "mirl_replay_tree_nodes": [_vp, _vp, _vp, _vp],
    "mirl_replay_tree_set_leaves": [_vp, _i64, _vp, _vp, _vp],
    "mirl_replay_tree_find": [_vp, _i32, _vp, _vp, _vp],
    "mirl_replay_losses_peek": [_vp, _i32, _i64, _i32, _vp],
    "mirl_q_target_dqn": [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _vp, _vp],
    "mirl_q_target_iqn": [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _vp, _vp],
    "mirl_loss_dqn": [_i64, _i32, _vp, _vp, _vp, _vp, _f64, _i32, _f64, _vp, _vp, _vp, _vp],
    "mirl_loss_iqn": [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _vp, _vp, _vp, _vp],
    "mirl_lstm_cell_fwd": [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mirl_lstm_seq_supported": [_i32, _i32, _i32],
    "mirl_lstm_seq_workspace_bytes": [_i32, _i32, _P(_i64)],
    "mirl_lstm_seq_fwd": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp],
    "mirl_lstm_seq_status": [_P(_i32)],
    "mirl_gemm3_mid_set": [_i32],
    "mirl_lstm_seq_fwd_grid": [_i32, _i32, _P(_i32), _P(_i64), _P(_i32), _P(_i32)],
    "mirl_lstm_seq_bwd_supported": [_i32, _i32, _i32],
    "mirl_lstm_seq_bwd_workspace_bytes": [_i32, _i32, _P(_i64)],
    "mirl_lstm_seq_bwd": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mirl_lstm_cell_bwd": [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp],
    "mirl_frames_to_f32_nhwc": [_i64, _i32, _i32, _vp, C.c_float, _vp, _vp],
    "mirl_frames_to_f32_nhwc_ex": [_i64, _i32, _i32, _vp, C.c_float, _vp, _i32, _i32, _vp],
    "mirl_conv1_u8_supported": [_i32, _i32, _i32, _i32, _i32, _i32],
    "mirl_conv1_bf16_set": [_i32],
    "mirl_conv1_wrw_bf16_set": [_i32],
    "mirl_conv1_u8_wpk_floats": [_P(_i64)],
    "mirl_conv1_u8_fwd": [_i64, _i32, _i32, _vp, _vp, _i64, _i64, _i64, _i64, _vp, C.c_float, _vp, _vp, _vp],
    "mirl_conv1_u8_fwd_ex": [_i64, _i32, _i32, _vp, _vp, _i64, _i64, _i64, _i64, _vp, C.c_float, _vp, _vp, _i32, _vp],
    "mirl_conv1_u8_wrw_scratch_floats": [_P(_i64)],
    "mirl_conv1_u8_wrw_ex": [_i64, _i32, _i32, _vp, _vp, C.c_float, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp],
    "mirl_conv1_u8_wrw_masked": [_i64, _i32, _i32, _vp, _vp, _vp, C.c_float, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp],
    "mirl_conv1_u8_wrw": [_i64, _i32, _i32, _vp, _vp, C.c_float, _vp, _vp, _i64, _i64, _i64, _i64, _vp],
    "mirl_conv2_bwd_data_supported": [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "mirl_conv2_bwd_data": [_i64, _i32, _i32, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp],
    "mirl_conv2_bwd_data_wpk_floats": [_P(_i64)],
    "mirl_conv3_bwd_data_supported": [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "mirl_conv3_bwd_data_wpk_floats": [_P(_i64)],
    "mirl_conv3_bwd_data": [_i64, _i32, _i32, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _vp],
    "mirl_conv2_bwd_data_ex": [_i64, _i32, _i32, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _i32, _vp],
    "mirl_conv3_fwd_supported": [_i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "mirl_conv3_fwd": [_i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp],
    "mirl_conv_wrw_b3_supported": [_i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "mirl_conv_wrw_b3_scratch_bytes": [_i32, _i32, _i32, _i32, _P(_i64)],
    "mirl_conv_wrw_b3": [_i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp],
    "mirl_gemm3_supported": [_i32, _i64, _i64, _i64],
    "mirl_gemm3_workspace_bytes": [_i32, _i64, _i64, _i64, _P(_i64)],
    "mirl_gemm3": [_i32, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _i64, _vp],
    "mirl_gemm3_nt_mul": [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _i64, _i32, _vp, _i64, _vp],
    "mirl_gemm3_nn_qp_partial_rows": [_i64, _P(_i64)],
    "mirl_gemm3_nn_qp": [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp],
    "mirl_gemm3_head_workspace_bytes": [_i64, _i64, _P(_i64)],
    "mirl_gemm3_nt_head": [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _i32, _vp, _vp, _i64, _vp, _i64, _vp],
    "mirl_act_conv_supported": [_i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "mirl_act_conv_fwd": [_i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp],
    "mirl_act_lstm_supported": [_i32, _i32, _i32],
    "mirl_act_lstm_workspace_bytes": [_i32, _i32, _i32, _P(_i64)],
    "mirl_act_lstm_fwd": [_i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mirl_act_embed": [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp],
    "mirl_act_head_supported": [_i32, _i32, _i32, _i32, _i32, _i32],
    "mirl_act_head_parts": [_i32, _i32, _P(_i32), _P(_i32)],
    "mirl_act_head_hidden": [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "mirl_act_head_select": [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _f64, _u64, _vp, _vp, _vp, _vp],
    "mirl_bias_relu_rows": [_i64, _i32, _vp, _vp, _vp],
    "mirl_colsum_blocks": [_i64, _i32, _P(_i32)],
    "mirl_relu_bwd_bias_rows": [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp],
    "mirl_cos_embed": [_i64, _i32, _vp, _vp, _vp, _vp],
    "mirl_cos_embed_rng": [_i64, _i32, _u64, _vp, _vp, _vp, _vp, _vp],
    "mirl_adam_clip_workspace_bytes": [_i32, _vp, _vp],
    "mirl_adam_clip_step": [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _f64, _f64, _f64, _f64, _vp, _i64, _vp, _vp],
    "mirl_iqn_mul_fwd": [_i64, _i32, _i32, _vp, _vp, _vp, _vp],
    "mirl_iqn_mul_bwd": [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp],
    "mirl_dueling_tail_bwd": [_i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp],
    "mirl_dueling_tail_bwd_w": [_i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp],
    "mirl_actor_head": [_i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _vp, _vp],
    "mirl_actor_head_rng": [_i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _f64, _u64, _vp, _vp, _vp, _vp, _vp],
    "mirl_stack_shift": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "mirl_synth_env_step": [_i32, _i64, _vp, _i32, _vp, _i32, _u64, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp, _vp],
    "mirl_synth_env_step_pre": [_i32, _i64, _vp, _i32, _vp, _i32, _u64, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp,
                                _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _vp],
    "mirl_actor_pre": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp,
                       _vp, _vp, _u64, _vp],
    "mirl_episode_track": [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mirl_copy_bytes": [_vp, _vp, _i64, _vp],
    "mirl_copy_bytes_ex": [_vp, _vp, _i64, _i32, _vp],
    "mirl_book_create": [_P(ReplayConfig), _P(_vp)],
    "mirl_book_destroy": [_vp],
    "mirl_book_ingest": [_vp, _i32, _vp],
    "mirl_book_stats": [_vp, _P(_i64), _P(_i64), _P(_i64), _P(_i64), _P(_i64)],
    "mirl_book_env_meta": [_vp, _vp, _vp],
    "mirl_book_free_slots": [_vp, _vp, _P(_i64)],
    "mirl_book_slot_table": [_vp, _vp, _vp],
    "mirl_book_needed_feed_count": [_vp, _i32, _i32, _P(_i64)],
    "mirl_book_charge_quota": [_vp, _i32],
    "mirl_book_uniform_total": [_vp, _P(_i64)],
    "mirl_book_uniform_map": [_vp, _i32, _vp, _vp, _vp],
    "mirl_emul_build_tree": [_i64, _vp, _vp, _vp, _vp],
    "mirl_emul_find": [_i64, _vp, _vp, _i32, _vp, _vp],
    "mirl_emul_seq_priority": [_i32, _f64, _f64, _vp, _P(_f64), _P(C.c_uint8)],
}

for _name, _args in _SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.argtypes = _args
    _fn.restype = C.c_int


def last_error():
    return (lib.mirl_last_error() or b"").decode()


def check(rc, what=""):
    """Raise on error codes; pass MIRL_OK / MIRL_NEED_MORE through."""
    if rc < 0:
        err = MirlError("%s failed (%d): %s" % (what or "mirl call", rc, last_error()))
        err.code = int(rc)
        raise err
    return rc


def device_count():
    return lib.mirl_device_count()


def require_gpu():
    if device_count() <= 0:
        raise MirlError(
            "librltime_hip: no HIP device visible. The rltime_amd hot path runs "
            "only on an AMD GPU (MI355X / gfx950); there is no CPU fallback.")


def profile_table():
    """Collect the per-kernel event timings gathered since the last reset:
    [{name, calls, total_ms, algorithmic_bytes}] (mirl_profile_*)."""
    n = C.c_int32()
    check(lib.mirl_profile_collect(C.byref(n)), "mirl_profile_collect")
    out = []
    for i in range(n.value):
        name = C.create_string_buffer(96)
        calls, ms, by = C.c_int64(), C.c_double(), C.c_double()
        check(lib.mirl_profile_get(i, name, 96, C.byref(calls), C.byref(ms), C.byref(by)), "mirl_profile_get")
        fl = C.c_double()
        check(lib.mirl_profile_get_flop(i, C.byref(fl)), "mirl_profile_get_flop")
        out.append({"name": name.value.decode(), "calls": calls.value, "total_ms": ms.value,
                    "algorithmic_bytes": by.value, "flop": fl.value})
    return out


def np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)
