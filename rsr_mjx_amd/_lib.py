"""ctypes binding of librsrmjx.so (include/rsr_mjx.h).  Fails loudly when the library is missing:
there is no CPU or PyTorch fallback for the stepper."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

FIELDS = [
    "qpos", "qvel", "ctrl", "qacc_warmstart", "time", "xpos", "site_xpos",
    "obs", "reward", "done", "metrics",
    "info_target_pos", "info_new_cube_pos", "info_site_pos", "info_cube_pos", "info_last_action",
    "info_target_base_pos", "info_target_vertical_pos", "info_target_w", "info_new_T_pos", "info_T_pos", "info_xita", "info_go2",
    "info_steps", "info_truncation", "info_episode_done", "info_episode_metrics",
    "first_qpos", "first_qvel", "first_ctrl", "first_warmstart", "first_time", "first_xpos", "first_site_xpos",
    "first_obs", "privileged_obs", "first_privileged_obs", "stats",
]
FIELD_ID = {n: i for i, n in enumerate(FIELDS)}
DEBUG_FLOATS = 8192


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nq", "nv", "nu", "nbody", "njnt", "ngeom", "nsite", "neq", "npair",
        "obs_dim", "nmetrics", "n_frames", "episode_length", "env_kind",
        "rec_floats", "ncon_max", "nefc_max", "lds_bytes")]


_lib = None

# every symbol include/rsr_mjx.h declares
# enum rsr_dr_field (include/rsr_mjx.h), in order
DR_FIELDS = [
    "geom_friction", "body_mass", "dof_damping", "dof_frictionloss",
    "body_ipos", "qpos0", "dof_armature", "actuator_gainprm", "actuator_biasprm",
]

SYMBOLS = [
    "rsr_model_create", "rsr_model_dims", "rsr_model_destroy", "rsr_batch_create", "rsr_batch_destroy",
    "rsr_batch_set_dr", "rsr_batch_set_dr_field", "rsr_batch_set_schedule", "rsr_batch_set_whole_envs", "rsr_batch_set_priority", "rsr_batch_set_action_repeat", "rsr_batch_check", "rsr_batch_set_fault_injection", "rsr_rollout_metrics", "rsr_reset", "rsr_step", "rsr_view", "rsr_batch_set_debug",
    "rsr_timing_begin", "rsr_timing_end", "rsr_last_error",
]


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("RSR_MJX_LIB", _build.LIB)   # override: diagnostic builds (tools/gpu_stage_profile.py)
    # torch ships its own HIP runtime; import it first so librsrmjx.so binds to the same libamdhip64
    # (two runtimes in one process do not see each other's device context).
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m rsr_mjx_amd.build` (hipcc, gfx950). "
            "The stepper has no CPU fallback.")
    L = C.CDLL(path)
    vp, i32, i64p = C.c_void_p, C.c_int, C.POINTER(C.c_int64)
    L.rsr_model_create.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.rsr_model_dims.argtypes = [vp, C.POINTER(Dims)]
    L.rsr_model_destroy.argtypes = [vp]
    L.rsr_model_destroy.restype = None
    L.rsr_batch_create.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
    L.rsr_batch_set_dr_field.argtypes = [vp, i32, vp]
    L.rsr_batch_destroy.argtypes = [vp]
    L.rsr_batch_destroy.restype = None
    L.rsr_batch_set_dr.argtypes = [vp, vp, vp, vp, vp]
    L.rsr_reset.argtypes = [vp, vp, vp]
    L.rsr_step.argtypes = [vp, vp, vp]
    L.rsr_view.argtypes = [vp, i32, C.POINTER(vp), i64p, i64p]
    L.rsr_batch_set_debug.argtypes = [vp, vp]
    L.rsr_batch_set_schedule.argtypes = [vp, i32]
    L.rsr_batch_set_whole_envs.argtypes = [vp, i32]
    L.rsr_batch_set_priority.argtypes = [vp, i32]
    L.rsr_batch_set_action_repeat.argtypes = [vp, i32]
    L.rsr_rollout_metrics.argtypes = [vp, vp, vp]
    L.rsr_batch_check.argtypes = [vp, vp, C.POINTER(C.c_int)]
    L.rsr_batch_set_fault_injection.argtypes = [vp, i32, i32]
    L.rsr_timing_begin.argtypes = [vp, vp]
    L.rsr_timing_end.argtypes = [vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.rsr_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"librsrmjx error {rc}: {lib().rsr_last_error().decode()}")
