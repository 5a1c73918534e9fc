"""Airbot cube-push env on the HIP stepper, mirroring the reference interface
(ppo_train/airbot_training/cube_env.py: `AirbotPlayBase.reset(rng) -> State`,
`.step(state, action) -> State`, properties observation_size / action_size / dt / sys / unwrapped)
and the training wrappers the reference applies at RSR/train.py:224-235
(`wrap(env, episode_length, action_repeat, randomization_fn)`).

Differences forced by the platform (SURVEY.md finding 0.7): the env is batched natively (one
wavefront per env; there is no vmap), arrays are torch tensors on the GPU, and `State` objects are
views into the batch's persistent record, updated in place by `step` (the reference returns new
immutable pytrees).  The stepper itself is rsr_step in librsrmjx.so; nothing here computes physics.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, Optional

import numpy as np

from .. import _lib, prng
from ..mjcf import CompiledModel, compile_mjcf
from ..model import model_fields, pack_blob
from . import config as cfg

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


@dataclass
class PipelineState:
    """Subset of brax.mjx.base.State the env and its callers read (q, qd = qpos, qvel)."""
    qpos: Any
    qvel: Any
    ctrl: Any
    qacc_warmstart: Any
    time: Any
    xpos: Any        # [N, nbody, 3], from the last forward pass (one substep stale, as in the reference)
    site_xpos: Any   # [N, nsite, 3]

    @property
    def q(self):
        return self.qpos

    @property
    def qd(self):
        return self.qvel


@dataclass
class State:
    """brax.envs.base.State: pipeline_state, obs, reward, done, metrics, info."""
    pipeline_state: PipelineState
    obs: Any
    reward: Any
    done: Any
    metrics: Dict[str, Any] = field(default_factory=dict)
    info: Dict[str, Any] = field(default_factory=dict)

    def replace(self, **kw) -> "State":
        d = dict(pipeline_state=self.pipeline_state, obs=self.obs, reward=self.reward, done=self.done,
                 metrics=self.metrics, info=self.info)
        d.update(kw)
        return State(**d)


class AirbotPlayBase:
    """Env definition (model + task constants).  `reset`/`step` run a batch sized by the keys."""

    _default_asset = "airbot_cube.npz"
    _fields_fn = staticmethod(cfg.cube_env_fields)
    _defaults = cfg.CUBE_DEFAULTS
    _obs_dim = cfg.CUBE_OBS_DIM
    _metrics = cfg.CUBE_METRICS

    def __init__(self, model_path: Optional[str] = None, device: str = "cuda:0", **kwargs):
        if model_path is None:
            self.sys: CompiledModel = CompiledModel.load(os.path.join(_ASSETS, self._default_asset))
        elif model_path.endswith(".npz"):
            self.sys = CompiledModel.load(model_path)
        else:
            self.sys = compile_mjcf(model_path)     # reference: mujoco.MjModel.from_xml_path("cube.xml")
        self._kwargs = dict(kwargs)
        self._device = device
        self._n_frames = kwargs.get("n_frames", kwargs.get("decimation", self._defaults["decimation"]))
        self._batched: Optional[BatchedEnv] = None
        self._fields_fn(self.sys, **self._kwargs)       # validates kwargs early, like the reference constructor

    # --- reference properties ---
    @property
    def observation_size(self) -> int:
        return self._obs_dim

    @property
    def action_size(self) -> int:
        return self.sys.nu

    @property
    def dt(self) -> float:
        return float(self.sys.arrays["opt_timestep"][0]) * self._n_frames

    @property
    def unwrapped(self) -> "AirbotPlayBase":
        return self

    @property
    def backend(self) -> str:
        return "hip-gfx950"

    def batched(self, num_envs: int, episode_length: int = 0, auto_reset: bool = False,
                randomization: Optional[Dict[str, Any]] = None) -> "BatchedEnv":
        return BatchedEnv(self, num_envs, episode_length, auto_reset, randomization)

    # unwrapped env semantics (no Episode/AutoReset), batch size taken from the keys
    def reset(self, rng) -> State:
        n = 1 if np.ndim(rng) == 1 else int(np.shape(rng)[0])
        if self._batched is None or self._batched.num_envs != n:
            self._batched = self.batched(n)
        return self._batched.reset(rng)

    def step(self, state: State, action) -> State:
        return self._batched.step(state, action)


class AirbotPlaySF(AirbotPlayBase):
    """The variant every RSR script uses (reference test/airbot.py, model test/sf.xml): wrist target held within 3 cm
    of the goal, task-complete bonus, done = cube at target."""

    _default_asset = "airbot_sf.npz"
    _fields_fn = staticmethod(cfg.sf_env_fields)
    _defaults = cfg.SF_DEFAULTS


class AirbotTShape(AirbotPlayBase):
    """T-block pushing task (reference ppo_train/airbot_training/T_shape_env.py, model T_shape.xml):
    16-dim obs, angle term `xita`, two geom-distance rewards."""

    _default_asset = "airbot_tshape.npz"
    _fields_fn = staticmethod(cfg.tshape_env_fields)
    _defaults = cfg.TSHAPE_DEFAULTS
    _obs_dim = cfg.TSHAPE_OBS_DIM
    _metrics = cfg.TSHAPE_METRICS


class BatchedEnv:
    """N envs on one GPU behind the C ABI; optionally with the Episode/AutoReset wrapper semantics fused."""

    def __init__(self, env: AirbotPlayBase, num_envs: int, episode_length: int, auto_reset: bool,
                 randomization: Optional[Dict[str, Any]]):
        import torch
        self.env, self.num_envs = env, int(num_envs)
        self.episode_length, self.auto_reset = int(episode_length), bool(auto_reset)
        self.sys = env.sys
        self.device = torch.device(env._device)
        if self.device.type != "cuda":
            raise RuntimeError("the stepper runs on a HIP device only (torch device 'cuda:N')")
        L = _lib.lib()
        f = model_fields(self.sys)
        f.update(env._fields_fn(self.sys, episode_length=self.episode_length, auto_reset=self.auto_reset, **env._kwargs))
        self.blob = pack_blob(f)
        buf = C.create_string_buffer(self.blob, len(self.blob))
        self._model = C.c_void_p()
        _lib.check(L.rsr_model_create(buf, len(self.blob), C.byref(self._model)))
        self.dims = _lib.Dims()
        _lib.check(L.rsr_model_dims(self._model, C.byref(self.dims)))
        self.record = torch.zeros((self.num_envs, self.dims.rec_floats), dtype=torch.float32, device=self.device)
        self._batch = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(L.rsr_batch_create(self._model, self.num_envs, dev_index, C.c_void_p(self.record.data_ptr()),
                                      C.byref(self._batch)))
        self._views: Dict[str, Any] = {}
        for name, fid in _lib.FIELD_ID.items():
            ptr, shape, stride = C.c_void_p(), (C.c_int64 * 2)(), (C.c_int64 * 2)()
            _lib.check(L.rsr_view(self._batch, fid, C.byref(ptr), shape, stride))
            off = (ptr.value - self.record.data_ptr()) // 4
            v = self.record[:, off:off + shape[1]]
            self._views[name] = v.view(torch.int32) if name == "stats" else v
        self._dr = {}
        self._debug = None
        if randomization is not None:
            self.set_randomization(randomization)
        self._state: Optional[State] = None

    # --- properties the reference callers read ---
    @property
    def observation_size(self) -> int:
        return self.env.observation_size

    @property
    def action_size(self) -> int:
        return self.env.action_size

    @property
    def dt(self) -> float:
        return self.env.dt

    @property
    def unwrapped(self):
        return self.env

    def __del__(self):
        try:
            L = _lib.lib()
            if getattr(self, "_batch", None):
                L.rsr_batch_destroy(self._batch)
            if getattr(self, "_model", None):
                L.rsr_model_destroy(self._model)
        except Exception:
            pass

    def view(self, name: str):
        return self._views[name]

    def set_randomization(self, dr: Dict[str, Any]) -> None:
        """Per-env model leaves (numpy or torch, leading dim N): geom_friction [N,ngeom,3], body_mass [N,nbody],
        dof_damping [N,nv], dof_frictionloss [N,nv]; the Go2 kernels also take body_ipos [N,nbody,3], qpos0 [N,nq],
        dof_armature [N,nv], actuator_gainprm / actuator_biasprm [N,nu,3] (go2/randomize.py:6-109).  Missing or None
        keys restore the model's value."""
        import torch
        unknown = set(dr) - set(_lib.DR_FIELDS)
        if unknown:
            raise KeyError(f"not a randomisable model field: {sorted(unknown)}")
        keep = {}
        for fid, k in enumerate(_lib.DR_FIELDS):
            v = dr.get(k)
            if v is None:
                if self._dr.get(k) is not None:
                    _lib.check(_lib.lib().rsr_batch_set_dr_field(self._batch, fid, None))
                continue
            t = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v, dtype=torch.float32).to(self.device)
            keep[k] = t.reshape(self.num_envs, -1).contiguous()
            _lib.check(_lib.lib().rsr_batch_set_dr_field(self._batch, fid, C.c_void_p(keep[k].data_ptr())))
        self._dr = keep

    def set_schedule(self, units: int) -> None:
        """Work units per env-step of the persistent step launch (results are bit-identical for every value)."""
        _lib.check(_lib.lib().rsr_batch_set_schedule(self._batch, int(units)))

    def set_whole_envs(self, whole_envs: int = -1) -> None:
        """Envs stepped as one work unit each, the rest in `units` phases (rsr_batch_set_whole_envs; -1 = the default split)."""
        _lib.check(_lib.lib().rsr_batch_set_whole_envs(self._batch, int(whole_envs)))

    def rollout_metrics(self, out=None):
        """One launch: float tensor [4] = (num_envs, sum of reward, sum of done, mean of the running episodes' summed reward) of
        this batch (rsr_rollout_metrics); needs the Episode wrapper's bookkeeping for the last entry."""
        import torch
        if out is None:
            out = torch.empty(4, dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().rsr_rollout_metrics(self._batch, C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def set_action_repeat(self, repeat: int = 1) -> None:
        """action_repeat of the training wrappers (brax EpisodeWrapper): one step = `repeat` env steps with the same action, rewards
        summed, steps advanced by `repeat`, done / truncation / AutoReset once after the last (rsr_batch_set_action_repeat)."""
        _lib.check(_lib.lib().rsr_batch_set_action_repeat(self._batch, int(repeat)))
        self.action_repeat = int(repeat)

    def set_priority(self, policy: int = -1) -> None:
        """Wave priority schedule of the plain-launch step kernels (rsr_batch_set_priority: 0 off, 1 rotate, 2 catch up, -1 by batch
        size); timing only, results are bit-identical."""
        _lib.check(_lib.lib().rsr_batch_set_priority(self._batch, int(policy)))

    def handoff_timeouts(self) -> int:
        """Synchronises the launch stream and returns the number of work-unit hand-off waits that timed out since the batch was
        created (rsr_batch_check; 0 on a healthy batch).  The envs concerned read stats[:, 3] == -1 after that step."""
        n = C.c_int()
        rc = _lib.lib().rsr_batch_check(self._batch, self._stream(), C.byref(n))
        if rc not in (0, -5):
            _lib.check(rc)
        return int(n.value)

    def check(self) -> None:
        """Raises if any hand-off wait has timed out (rsr_batch_check)."""
        _lib.check(_lib.lib().rsr_batch_check(self._batch, self._stream(), None))

    def set_fault_injection(self, spin_cap: int = 0, withhold_env: int = -1) -> None:
        """Test hook (rsr_batch_set_fault_injection)."""
        _lib.check(_lib.lib().rsr_batch_set_fault_injection(self._batch, int(spin_cap), int(withhold_env)))

    def enable_debug(self, on: bool = True):
        import torch
        if on:
            self._debug = torch.zeros((self.num_envs, _lib.DEBUG_FLOATS), dtype=torch.float32, device=self.device)
            _lib.check(_lib.lib().rsr_batch_set_debug(self._batch, C.c_void_p(self._debug.data_ptr())))
        else:
            self._debug = None
            _lib.check(_lib.lib().rsr_batch_set_debug(self._batch, None))
        return self._debug

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _make_state(self) -> State:
        v = self._views
        n = self.num_envs
        ps = PipelineState(qpos=v["qpos"], qvel=v["qvel"], ctrl=v["ctrl"], qacc_warmstart=v["qacc_warmstart"],
                           time=v["time"][:, 0], xpos=v["xpos"].unflatten(1, (self.dims.nbody, 3)),
                           site_xpos=v["site_xpos"].unflatten(1, (self.dims.nsite, 3)))
        mnames = self.env._metrics
        metrics = {name: v["metrics"][:, i] for i, name in enumerate(mnames)}
        if isinstance(self.env, AirbotTShape):     # T_shape_env.py:127-134
            info = {"target_base_pos": v["info_target_base_pos"], "target_vertical_pos": v["info_target_vertical_pos"],
                    "target_w": v["info_target_w"][:, 0], "new_T_pos": v["info_new_T_pos"], "site_pos": v["info_site_pos"],
                    "T_pos": v["info_T_pos"], "xita": v["info_xita"][:, 0]}
        else:                                      # cube_env.py:135-140
            info = {
                "target_pos": v["info_target_pos"], "new_cube_pos": v["info_new_cube_pos"],
                "site_pos": v["info_site_pos"], "cube_pos": v["info_cube_pos"],
                "reached_box": self.record.new_zeros((n,)),
            }
        if isinstance(self.env, AirbotPlaySF):
            info["last_action"] = v["info_last_action"][:, 0]
        if self.episode_length > 0:
            em = v["info_episode_metrics"]
            info.update(steps=v["info_steps"][:, 0], truncation=v["info_truncation"][:, 0],
                        episode_done=v["info_episode_done"][:, 0],
                        episode_metrics={"sum_reward": em[:, 0], "length": em[:, 1],
                                         **{name: em[:, 2 + i] for i, name in enumerate(mnames)}})
        if self.auto_reset:
            info.update(first_obs=v["first_obs"],
                        first_pipeline_state=PipelineState(
                            qpos=v["first_qpos"], qvel=v["first_qvel"], ctrl=v["first_ctrl"],
                            qacc_warmstart=v["first_warmstart"], time=v["first_time"][:, 0],
                            xpos=v["first_xpos"].unflatten(1, (self.dims.nbody, 3)),
                            site_xpos=v["first_site_xpos"].unflatten(1, (self.dims.nsite, 3))))
        return State(pipeline_state=ps, obs=v["obs"], reward=v["reward"][:, 0], done=v["done"][:, 0],
                     metrics=metrics, info=info)

    def reset(self, rng) -> State:
        """rng: uint32 key data [N, 2] (numpy or torch), as produced by jax.random.split / prng.split."""
        import torch
        if torch.is_tensor(rng):
            keys = rng.to(device=self.device, dtype=torch.int64).to(torch.int32) if rng.dtype != torch.int32 else rng.to(self.device)
        else:
            k = np.ascontiguousarray(np.asarray(rng, dtype=np.uint32)).reshape(-1, 2)
            keys = torch.from_numpy(k.view(np.int32)).to(self.device)
        if keys.shape != (self.num_envs, 2):
            raise ValueError(f"reset expects keys of shape ({self.num_envs}, 2), got {tuple(keys.shape)}")
        keys = keys.contiguous()
        _lib.check(_lib.lib().rsr_reset(self._batch, C.c_void_p(keys.data_ptr()), self._stream()))
        self._keys = keys
        self._state = self._make_state()
        return self._state

    def step(self, state: Optional[State], action) -> State:
        import torch
        if self._state is None:
            raise RuntimeError("step before reset")
        a = torch.as_tensor(action, dtype=torch.float32, device=self.device)
        if a.shape != (self.num_envs, self.dims.nu):
            raise ValueError(f"step expects actions of shape ({self.num_envs}, {self.dims.nu}), got {tuple(a.shape)}")
        a = a.contiguous()
        _lib.check(_lib.lib().rsr_step(self._batch, C.c_void_p(a.data_ptr()), self._stream()))
        self._last_action = a
        return self._state

    # --- timing hooks used by bench.py (HIP events on the launch stream) ---
    def timing_begin(self):
        _lib.check(_lib.lib().rsr_timing_begin(self._batch, self._stream()))

    def timing_end(self):
        ms, n = C.c_float(), C.c_int()
        _lib.check(_lib.lib().rsr_timing_end(self._batch, self._stream(), C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)


def wrap(env: AirbotPlayBase, num_envs: int, episode_length: int = 1000, action_repeat: int = 1,
         randomization_fn: Optional[Callable[[CompiledModel], Dict[str, Any]]] = None) -> BatchedEnv:
    """Counterpart of brax.envs.training.wrap as called at reference RSR/train.py:224-229:
    Vmap | DomainRandomizationVmap -> Episode -> AutoReset, fused into the step kernel."""
    dr = randomization_fn(env.sys) if randomization_fn is not None else None
    benv = env.batched(num_envs, episode_length=episode_length, auto_reset=True, randomization=dr)
    if action_repeat != 1:          # (the reference trains with action_repeat = 1, train.py:47: the wrappers fused in the step kernel)
        benv.set_action_repeat(action_repeat)
    return benv


def wrap_sub_batches(env, num_envs: int, parts: int, episode_length: int = 1000, action_repeat: int = 1,
                     randomization_fn: Optional[Callable[[CompiledModel], Dict[str, Any]]] = None):
    """`wrap`, but as `parts` independent BatchedEnvs over contiguous blocks of the env index, for consumers that work per
    sub-batch on separate HIP streams (rollout.generate_unroll_pipelined, bench.py `sub_batched`): one sub-batch's launch
    tail is then hidden behind the other's next launch.  Env i is the same env as in `wrap`: keys and
    randomised leaves are sliced by index.  Works for every env definition with a `batched` method (Airbot, Go2)."""
    if num_envs % parts:
        raise ValueError("num_envs must be a multiple of parts")
    m = num_envs // parts
    dr = randomization_fn(env.sys) if randomization_fn is not None else None
    out = []
    for k in range(parts):
        sub = None if dr is None else {f: v[k * m:(k + 1) * m] for f, v in dr.items()}
        out.append(env.batched(m, episode_length=episode_length, auto_reset=True, randomization=sub))
        if action_repeat != 1:
            out[-1].set_action_repeat(action_repeat)
    return out


def domain_randomize(sys: CompiledModel, rng: np.ndarray) -> Dict[str, np.ndarray]:
    """reference ppo_train/airbot_training/domain_randomize.py:26-91: six uniforms per env scale table / cube /
    finger friction, cube mass, arm dof damping and frictionloss.  rng: uint32 [N, 2]."""
    rng = np.asarray(rng, dtype=np.uint32).reshape(-1, 2)
    n = rng.shape[0]
    A = sys.arrays
    table, cube_g, cube_b = sys.id("geom", "table-b"), sys.id("geom", "geom_for_push"), sys.id("body", "cube_for_push")
    finger_bodies = {sys.id("body", "left"), sys.id("body", "right")}
    fingers = [g for g in range(sys.ngeom) if int(A["geom_bodyid"][g]) in finger_bodies]
    scales = []
    for lo, hi in ((0.68, 1.32), (0.68, 1.32), (0.84, 1.16), (0.76, 1.24), (0.92, 1.08), (0.92, 1.08)):
        ks = prng.split(rng, 2)                 # rng, key = split(rng)
        rng, key = ks[:, 0], ks[:, 1]
        scales.append(prng.uniform(key, (), lo, hi).astype(np.float32))
    table_s, cube_s, mass_s, finger_s, damp_s, floss_s = scales
    fr = np.tile(A["geom_friction"].astype(np.float32)[None], (n, 1, 1))
    fr[:, table] *= table_s[:, None]
    fr[:, cube_g] *= cube_s[:, None]
    fr[:, fingers] *= finger_s[:, None, None]
    mass = np.tile(A["body_mass"].astype(np.float32)[None], (n, 1))
    mass[:, cube_b] *= mass_s
    damp = np.tile(A["dof_damping"].astype(np.float32)[None], (n, 1))
    damp[:, 0:8] *= damp_s[:, None]
    floss = np.tile(A["dof_frictionloss"].astype(np.float32)[None], (n, 1))
    floss[:, 0:8] *= floss_s[:, None]
    return dict(geom_friction=fr, body_mass=mass, dof_damping=damp, dof_frictionloss=floss)
