"""ctypes wrapper around the CPU oracle (oracle/rsr_oracle.c).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product.
State is held as plain numpy float32 arrays [N, k] (struct of arrays).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

_BATCH_FIELDS = [
    "qpos", "qvel", "ctrl", "qacc_warmstart", "time", "xpos", "site_xpos",
    "obs", "reward", "done", "metrics",
    "info_target_pos", "info_new_cube_pos", "info_site_pos", "info_cube_pos", "info_last_action",
    "info_steps", "info_truncation", "info_episode_done", "info_episode_metrics",
    "info_target_base_pos", "info_target_vertical_pos", "info_target_w", "info_new_T_pos", "info_T_pos", "info_xita",
    "info_go2",
    "first_qpos", "first_qvel", "first_ctrl", "first_warmstart", "first_time", "first_xpos", "first_site_xpos",
    "first_obs", "priv_obs", "first_priv_obs",
    "dr_geom_friction", "dr_body_mass", "dr_dof_damping", "dr_dof_frictionloss",
    "dr_body_ipos", "dr_qpos0", "dr_dof_armature", "dr_gainprm", "dr_biasprm",
]
_DR_KEYS = [f for f in _BATCH_FIELDS if f.startswith("dr_")]


class _OBatch(C.Structure):
    _fields_ = [("n", C.c_int)] + [(f, C.POINTER(C.c_float)) for f in _BATCH_FIELDS] + [("stats", C.POINTER(C.c_int))]


def build(force: bool = False) -> None:
    """Compiles liboracle_f32.so / liboracle_f64.so next to the source (gcc, seconds)."""
    need = force or any(
        not os.path.exists(os.path.join(_HERE, f)) or
        os.path.getmtime(os.path.join(_HERE, f)) < os.path.getmtime(os.path.join(_HERE, "rsr_oracle.c"))
        for f in ("liboracle_f32.so", "liboracle_f64.so"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)


def reset_switches() -> None:
    """The oracle's switches (contact capacity, culling, line-search rule and cycle cut) are variables of the shared library, not of
    an Oracle object: back to the defaults in both builds (tests/conftest.py calls this before every test, so that no test inherits
    what another one set)."""
    for precision in ("f32", "f64"):
        path = os.path.join(_HERE, f"liboracle_{precision}.so")
        if os.path.exists(path):
            lib = C.CDLL(path)
            lib.oracle_set_ncon_cap(1 << 20)
            lib.oracle_set_cull(1)
            lib.oracle_set_ls_rule.argtypes = [C.c_int, C.c_double]
            lib.oracle_set_ls_rule(0, 1.0)
            lib.oracle_set_ls_cycle(0)


def _load(precision: str) -> C.CDLL:
    path = os.path.join(_HERE, f"liboracle_{precision}.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.oracle_model_create.restype = C.c_void_p
    lib.oracle_model_create.argtypes = [C.c_void_p, C.c_int]
    lib.oracle_model_destroy.argtypes = [C.c_void_p]
    lib.oracle_reset.argtypes = [C.c_void_p, C.POINTER(_OBatch), C.c_void_p, C.c_int]
    lib.oracle_step.argtypes = [C.c_void_p, C.POINTER(_OBatch), C.c_void_p, C.c_int]
    lib.oracle_debug_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.oracle_debug_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
    lib.oracle_threefry2x32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.oracle_split.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.oracle_uniform.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.oracle_box_box.argtypes = [C.c_void_p] * 8
    lib.oracle_plane_box.argtypes = [C.c_void_p] * 7
    for prec_real in ((C.c_float if precision == "f32" else C.c_double),):
        lib.oracle_plane_capsule.argtypes = [C.c_void_p] * 4 + [prec_real, prec_real] + [C.c_void_p] * 3
        lib.oracle_plane_cylinder.argtypes = [C.c_void_p] * 4 + [prec_real, prec_real] + [C.c_void_p] * 2
    lib.oracle_hfield_sphere.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


class Oracle:
    """One compiled model + env config (an RSRM blob) on the CPU restatement."""

    def __init__(self, blob: bytes, precision: str = "f32"):
        self.lib = _load(precision)
        self.real = np.float32 if precision == "f32" else np.float64
        self._blob = blob
        buf = C.create_string_buffer(blob, len(blob))
        self.h = self.lib.oracle_model_create(buf, len(blob))
        if not self.h:
            raise RuntimeError("oracle_model_create failed (model exceeds oracle maxima?)")
        from rsr_mjx_amd.model import unpack_blob  # data-format helper only
        f = unpack_blob(blob)
        d = f["dims"]
        self.nq, self.nv, self.nu, self.nbody, self.njnt, self.ngeom, self.nsite, self.neq, self.npair = map(int, d)
        ei = f["env_int"]
        self.obs_dim, self.nmetrics = int(ei[4]), int(ei[5])

    def __del__(self):
        try:
            self.lib.oracle_model_destroy(self.h)
        except Exception:
            pass

    # ---- batch state ----
    def new_state(self, n: int, dr: Optional[Dict[str, np.ndarray]] = None) -> Dict[str, np.ndarray]:
        z = lambda *s: np.zeros((n,) + s, dtype=np.float32)
        st = dict(
            qpos=z(self.nq), qvel=z(self.nv), ctrl=z(self.nu), qacc_warmstart=z(self.nv), time=z(),
            xpos=z(self.nbody, 3), site_xpos=z(self.nsite, 3),
            obs=z(self.obs_dim), reward=z(), done=z(), metrics=z(self.nmetrics),
            info_target_pos=z(3), info_new_cube_pos=z(2), info_site_pos=z(3), info_cube_pos=z(3), info_last_action=z(),
            info_steps=z(), info_truncation=z(), info_episode_done=z(), info_episode_metrics=z(2 + self.nmetrics),
            info_target_base_pos=z(3), info_target_vertical_pos=z(3), info_target_w=z(), info_new_T_pos=z(2), info_T_pos=z(3), info_xita=z(), info_go2=z(144),
            first_qpos=z(self.nq), first_qvel=z(self.nv), first_ctrl=z(self.nu), first_warmstart=z(self.nv),
            first_time=z(), first_xpos=z(self.nbody, 3), first_site_xpos=z(self.nsite, 3), first_obs=z(self.obs_dim),
            priv_obs=z(123), first_priv_obs=z(123),
            stats=np.zeros((n, 4), dtype=np.int32),
        )
        for k in _DR_KEYS:
            st[k] = None if dr is None or dr.get(k[3:]) is None else np.ascontiguousarray(dr[k[3:]], dtype=np.float32).reshape(n, -1)
        return st

    def _batch(self, st) -> _OBatch:
        b = _OBatch()
        b.n = st["qpos"].shape[0]
        for f in _BATCH_FIELDS:
            a = st[f]
            setattr(b, f, None if a is None else a.ctypes.data_as(C.POINTER(C.c_float)))
        b.stats = st["stats"].ctypes.data_as(C.POINTER(C.c_int))
        return b

    def reset(self, st, keys: np.ndarray, threads: int = 0) -> None:
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        assert keys.shape == (st["qpos"].shape[0], 2)
        b = self._batch(st)
        rc = self.lib.oracle_reset(self.h, C.byref(b), keys.ctypes.data, threads)
        if rc:
            raise RuntimeError(f"oracle_reset rc={rc}")

    def step(self, st, action: np.ndarray, threads: int = 0) -> None:
        action = np.ascontiguousarray(action, dtype=np.float32)
        assert action.shape == (st["qpos"].shape[0], self.nu)
        b = self._batch(st)
        rc = self.lib.oracle_step(self.h, C.byref(b), action.ctypes.data, threads)
        if rc:
            raise RuntimeError(f"oracle_step rc={rc}")

    # ---- debugging / invariants ----
    def forward(self, qpos, qvel, ctrl, warm=None, step: bool = False) -> int:
        r = self.real
        qpos, qvel, ctrl = (np.ascontiguousarray(x, dtype=r) for x in (qpos, qvel, ctrl))
        warm = None if warm is None else np.ascontiguousarray(warm, dtype=r)
        return self.lib.oracle_debug_forward(self.h, qpos.ctypes.data, qvel.ctypes.data, ctrl.ctypes.data,
                                             None if warm is None else warm.ctypes.data, int(step))

    def get(self, name: str, cap: int = 1 << 20) -> np.ndarray:
        out = np.zeros(cap, dtype=np.float64)
        n = self.lib.oracle_debug_get(self.h, name.encode(), out.ctypes.data, cap)
        if n < 0:
            raise KeyError(name)
        return out[:n].copy()

    def cost(self, qacc) -> float:
        """The solver's objective at `qacc` on the constraint rows of the last forward() (evaluated in double)."""
        q = np.ascontiguousarray(qacc, dtype=self.real)
        self.lib.oracle_debug_cost.restype = C.c_double
        self.lib.oracle_debug_cost.argtypes = [C.c_void_p, C.c_void_p]
        return float(self.lib.oracle_debug_cost(self.h, q.ctypes.data))

    def set_cull(self, on: bool) -> None:
        self.lib.oracle_set_cull(int(on))

    def set_ncon_cap(self, cap: int) -> None:
        self.lib.oracle_set_ncon_cap(int(cap))

    def set_ls_rule(self, rule: int, noise_eps: float = 1.0) -> None:
        """Line-search stop rule: 0 = MJX's (default), 1 = + the HIP kernel's fp32 noise-floor stop, 2 = 1 without the sign test."""
        self.lib.oracle_set_ls_rule.argtypes = [C.c_int, C.c_double]
        self.lib.oracle_set_ls_rule(int(rule), float(noise_eps))

    def set_ls_cycle(self, on: bool) -> None:
        """Exact shortcut of the line search's limit cycles (bit-identical results, fewer iterations)."""
        self.lib.oracle_set_ls_cycle(int(on))

    def ls_counters(self, reset: bool = False):
        """(calls, iterations, histogram[52] of iterations per call) of the line search since the last reset."""
        out = (C.c_longlong * 54)()
        self.lib.oracle_ls_counters(out, int(reset))
        return int(out[0]), int(out[1]), np.array(out[2:54], dtype=np.int64)

    def max_threads(self) -> int:
        return int(self.lib.oracle_max_threads())


# ---- PRNG helpers (threefry restatement) ----
def threefry2x32(key, ctr, precision="f32"):
    lib = _load(precision)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    out = np.zeros(2, dtype=np.uint32)
    lib.oracle_threefry2x32(key.ctypes.data, ctr.ctypes.data, out.ctypes.data)
    return out


def split(key, n, precision="f32"):
    lib = _load(precision)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros((n, 2), dtype=np.uint32)
    lib.oracle_split(key.ctypes.data, n, out.ctypes.data)
    return out


def uniform(key, n, lo, hi, precision="f32"):
    lib = _load(precision)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    lo = np.ascontiguousarray(np.broadcast_to(np.asarray(lo, dtype=np.float32), (n,)))
    hi = np.ascontiguousarray(np.broadcast_to(np.asarray(hi, dtype=np.float32), (n,)))
    out = np.zeros(n, dtype=np.float32)
    lib.oracle_uniform(key.ctypes.data, n, lo.ctypes.data, hi.ctypes.data, 1, out.ctypes.data)
    return out
