"""Go2 joystick env on the HIP stepper, mirroring the reference Playground interface
(mujoco_playground/_src/locomotion/go2/joystick.py `Joystick.reset/step`, `_src/mjx_env.py` `State` with `.data`,
`_src/registry.py` `load(name, config)`, `_src/wrapper.py` `wrap_for_brax_training`).

Built: tasks "flat_terrain" and "rough_terrain" (scene_mjx_feetonly_{flat,rough}_terrain.xml; the rough scene's floor is
a 256x256 height field), the 48-dim `state` observation (what `SelectObservationWrapper(obs_key="state")`,
wrapper.py:77-104, hands to the learner), all 21 reward terms, command resampling, action / IMU delay FIFOs, the
domain randomisation of go2/randomize.py, the 123-dim `privileged_state` (accelerometer via the post-constraint body
accelerations), perturbation kicks (`pert_config.enable`), Episode + AutoReset wrappers fused.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

import numpy as np

from .. import prng

from ..mjcf import CompiledModel, compile_mjcf
from . import config as cfg
from .airbot import BatchedEnv, State, _ASSETS


class Go2Data:
    """Subset of mjx.Data the env callers read (State.data in the Playground protocol, _src/mjx_env.py:66-74)."""

    def __init__(self, v, dims):
        self.qpos, self.qvel, self.ctrl = v["qpos"], v["qvel"], v["ctrl"]
        self.qacc_warmstart, self.time = v["qacc_warmstart"], v["time"][:, 0]
        self.xpos = v["xpos"].unflatten(1, (dims.nbody, 3))
        self.site_xpos = v["site_xpos"].unflatten(1, (dims.nsite, 3))


# slices of the 144-float info block (rsr_mjx.hip enum G2_*; joystick.py:175-196)
_INFO = dict(command=(0, 3), steps_until_next_cmd=(3, 4), last_act=(4, 16), last_last_act=(16, 28), feet_air_time=(28, 32),
             feet_contact_time=(32, 36), last_contact=(36, 40), swing_peak=(40, 44), action_buffer=(44, 92),
             gyro_buffer=(92, 104), linvel_buffer=(104, 116), gravity_buffer=(116, 128), steps_until_next_pert=(128, 129),
             pert_duration_seconds=(129, 130), pert_duration=(130, 131), steps_since_last_pert=(131, 132),
             pert_steps=(132, 133), pert_dir=(133, 136), pert_mag=(136, 137), rng=(137, 139),
             xfrc_applied_torso=(139, 142))      # data.xfrc_applied[torso, :3] (zeroed again by auto-reset)


_TASK_ASSET = {"flat_terrain": "go2_flat.npz", "rough_terrain": "go2_rough.npz"}


class Joystick:
    """Env definition; `batched()` / `wrap_for_brax_training()` give the N-env GPU batch."""

    _obs_dim = cfg.GO2_OBS_DIM
    _metrics = cfg.GO2_METRICS

    def __init__(self, task: str = "flat_terrain", config: Optional[dict] = None,
                 config_overrides: Optional[Dict[str, Any]] = None, model_path: Optional[str] = None, device: str = "cuda:0"):
        if task not in _TASK_ASSET:                                      # go2_constants.py:15-19 task_to_xml
            raise KeyError(f"Go2 task {task!r}: built tasks are {sorted(_TASK_ASSET)}")
        self._config = cfg._merge(config or cfg.GO2_DEFAULT_CONFIG, config_overrides or {})
        if model_path is None:
            base = CompiledModel.load(os.path.join(_ASSETS, _TASK_ASSET[task]))
        elif model_path.endswith(".npz"):
            base = CompiledModel.load(model_path)
        else:
            base = compile_mjcf(model_path)
        self.sys = cfg.go2_apply_overrides(base, self._config)          # go2/base.py:25-31
        self._device = device
        self._kwargs: Dict[str, Any] = {}
        self._n_frames = int(round(self._config["ctrl_dt"] / self._config["sim_dt"]))
        cfg.go2_env_fields(self.sys, self._config)                      # validates the config early

    def _fields_fn(self, sys, episode_length=0, auto_reset=False, **_):
        return cfg.go2_env_fields(sys, self._config, episode_length, auto_reset)

    @property
    def observation_size(self) -> int:
        return self._obs_dim

    @property
    def observation_sizes(self) -> Dict[str, tuple]:
        """the reference's dict-valued observation_size (_src/mjx_env.py:143-149)"""
        return {"state": (self._obs_dim,), "privileged_state": (cfg.GO2_PRIV_OBS_DIM,)}

    @property
    def action_size(self) -> int:
        return self.sys.nu

    @property
    def dt(self) -> float:
        return float(self._config["ctrl_dt"])

    @property
    def sim_dt(self) -> float:
        return float(self._config["sim_dt"])

    @property
    def n_substeps(self) -> int:
        return self._n_frames

    @property
    def unwrapped(self):
        return self

    def batched(self, num_envs: int, episode_length: int = 0, auto_reset: bool = False, randomization=None) -> "Go2Batched":
        return Go2Batched(self, num_envs, episode_length, auto_reset, randomization)


class Go2Batched(BatchedEnv):
    def _make_state(self) -> State:
        v = self._views
        data = Go2Data(v, self.dims)
        g = v["info_go2"]
        info: Dict[str, Any] = {}
        for k, (a, b) in _INFO.items():
            t = g[:, a:b]
            info[k] = t[:, 0] if b - a == 1 else t
        info["action_buffer"] = g[:, 44:92].unflatten(1, (4, 12))
        for k, (a, b) in (("gyro_buffer", (92, 104)), ("linvel_buffer", (104, 116)), ("gravity_buffer", (116, 128))):
            info[k] = g[:, a:b].unflatten(1, (4, 3))
        metrics = {name: v["metrics"][:, i] for i, name in enumerate(cfg.GO2_METRICS)}
        if self.episode_length > 0:
            em = v["info_episode_metrics"]
            info.update(steps=v["info_steps"][:, 0], truncation=v["info_truncation"][:, 0], episode_done=v["info_episode_done"][:, 0],
                        episode_metrics={"sum_reward": em[:, 0], "length": em[:, 1],
                                         **{name: em[:, 2 + i] for i, name in enumerate(cfg.GO2_METRICS)}})
        if self.auto_reset:
            info["first_obs"] = {"state": v["first_obs"], "privileged_state": v["first_privileged_obs"]}
        st = State(pipeline_state=data, obs=v["obs"], reward=v["reward"][:, 0], done=v["done"][:, 0], metrics=metrics, info=info)
        st.data = data                      # Playground name of the physics state
        # joystick.py:363-366 returns both observations; `obs` above is the one SelectObservationWrapper(obs_key="state")
        # hands on, `obs_dict` the full reference dict (asymmetric actor-critic reads "privileged_state")
        st.obs_dict = {"state": v["obs"], "privileged_state": v["privileged_obs"]}
        return st


def wrap_for_brax_training(env: Joystick, num_envs: int, episode_length: int = 1000, action_repeat: int = 1,
                           randomization_fn=None) -> Go2Batched:
    """Counterpart of reference _src/wrapper.py:41-74 (Vmap -> Episode -> AutoReset), fused into the step kernel."""
    benv = env.batched(num_envs, episode_length=episode_length, auto_reset=True)
    if randomization_fn is not None:
        benv.set_randomization(randomization_fn(env.sys))     # the per-env leaves of MjxDomainRandomizationVmapWrapper (wrapper.py:107-138)
    if action_repeat != 1:          # wrapper.py:69: brax EpisodeWrapper(env, episode_length, action_repeat)
        benv.set_action_repeat(action_repeat)
    return benv


FLOOR_GEOM_ID = 0
TORSO_BODY_ID = 1


def domain_randomize(sys, rng: np.ndarray) -> Dict[str, np.ndarray]:
    """Counterpart of reference _src/locomotion/go2/randomize.py:6-109: per env, floor friction U(0.4, 1), leg
    frictionloss x U(0.9, 1.1), leg armature x U(1, 1.05), kp x U(0.95, 1.05) (gainprm[:, 0] and biasprm[:, 1]), kd (leg
    dof_damping) x U(0.95, 1.05), torso com + U(-0.2, 0.2)^3, link masses x U(0.9, 1.1), torso mass + U(-3, 3), and
    qpos0[7:] + U(-0.05, 0.05).  rng: uint32 [N, 2] (one key per env).  Returns the nine per-env model leaves keyed by
    field name (actuator_gainprm / actuator_biasprm in this build's 3-column form)."""
    rng = np.asarray(rng, dtype=np.uint32).reshape(-1, 2)
    n = rng.shape[0]
    A = sys.arrays
    f32 = lambda k: np.tile(A[k].astype(np.float32)[None], (n,) + (1,) * A[k].ndim)

    def draw(shape, lo, hi):
        nonlocal rng
        ks = prng.split(rng, 2)                 # rng, key = jax.random.split(rng)
        rng, key = ks[:, 0], ks[:, 1]
        return prng.uniform(key, shape, lo, hi).astype(np.float32)

    fr = f32("geom_friction")
    fr[:, FLOOR_GEOM_ID, 0] = draw((), 0.4, 1.0)
    floss = f32("dof_frictionloss")
    floss[:, 6:] = floss[:, 6:] * draw((12,), 0.9, 1.1)
    arma = f32("dof_armature")
    arma[:, 6:] = arma[:, 6:] * draw((12,), 1.0, 1.05)
    kp = draw((12,), 0.95, 1.05)
    gain, bias = f32("actuator_gainprm"), f32("actuator_biasprm")
    gain[:, :, 0] = gain[:, :, 0] * kp
    bias[:, :, 1] = bias[:, :, 1] * kp
    damp = f32("dof_damping")
    damp[:, 6:] = damp[:, 6:] * draw((12,), 0.95, 1.05)
    dpos_x = draw((), -0.2, 0.2)
    dpos_yz = draw((2,), -0.2, 0.2)
    ipos = f32("body_ipos")
    ipos[:, TORSO_BODY_ID] = ipos[:, TORSO_BODY_ID] + np.concatenate([dpos_x[:, None], dpos_yz], axis=1)
    mass = f32("body_mass") * draw((sys.nbody,), 0.9, 1.1)
    mass[:, TORSO_BODY_ID] = mass[:, TORSO_BODY_ID] + draw((), -3.0, 3.0)
    qpos0 = f32("qpos0")
    qpos0[:, 7:] = qpos0[:, 7:] + draw((12,), -0.05, 0.05)
    return dict(geom_friction=fr, body_ipos=ipos, body_mass=mass, qpos0=qpos0, dof_frictionloss=floss,
                dof_armature=arma, actuator_gainprm=gain, actuator_biasprm=bias, dof_damping=damp)


class Handstand(Joystick):
    """Go2 Handstand task (reference go2/handstand.py:53-291) on the full-collision-vs-floor model (scene_mjx_flat_terrain.xml ->
    go2_mjx.xml: 4 foot spheres, 20 capsules, 6 cylinders against the floor plane): 45-dim `state`, 94-dim `privileged_state`,
    eleven reward terms, termination on a fall, an unwanted contact or the energy threshold."""

    _obs_dim = cfg.HANDSTAND_OBS_DIM
    _priv_dim = cfg.HANDSTAND_PRIV_OBS_DIM
    _metrics = cfg.HANDSTAND_METRICS
    _variant = "handstand"

    def __init__(self, config: Optional[dict] = None, config_overrides: Optional[Dict[str, Any]] = None,
                 model_path: Optional[str] = None, device: str = "cuda:0"):
        self._config = cfg._merge(config or cfg.HANDSTAND_DEFAULT_CONFIG, config_overrides or {})
        if model_path is None:
            base = CompiledModel.load(os.path.join(_ASSETS, "go2_full.npz"))
        elif model_path.endswith(".npz"):
            base = CompiledModel.load(model_path)
        else:
            base = compile_mjcf(model_path)
        self.sys = cfg.go2_apply_overrides(base, self._config)          # go2/base.py:25-31
        self._device = device
        self._kwargs: Dict[str, Any] = {}
        self._n_frames = int(round(self._config["ctrl_dt"] / self._config["sim_dt"]))
        cfg.handstand_env_fields(self.sys, self._config, variant=self._variant)

    def _fields_fn(self, sys, episode_length=0, auto_reset=False, **_):
        return cfg.handstand_env_fields(sys, self._config, episode_length, auto_reset, variant=self._variant)

    @property
    def observation_sizes(self) -> Dict[str, tuple]:
        return {"state": (self._obs_dim,), "privileged_state": (self._priv_dim,)}

    def batched(self, num_envs: int, episode_length: int = 0, auto_reset: bool = False, randomization=None) -> "HandstandBatched":
        return HandstandBatched(self, num_envs, episode_length, auto_reset, randomization)


class HandstandBatched(Go2Batched):
    """State of the Handstand env: info = {step, rng, last_act} (handstand.py:139-143), obs dict with the 94-dim privileged state."""

    def _make_state(self) -> State:
        v = self._views
        data = Go2Data(v, self.dims)
        g = v["info_go2"]
        info: Dict[str, Any] = {"step": g[:, 0], "last_act": g[:, 4:16], "rng": g[:, 137:139]}
        metrics = {name: v["metrics"][:, i] for i, name in enumerate(cfg.HANDSTAND_METRICS)}
        if self.episode_length > 0:
            em = v["info_episode_metrics"]
            info.update(steps=v["info_steps"][:, 0], truncation=v["info_truncation"][:, 0], episode_done=v["info_episode_done"][:, 0],
                        episode_metrics={"sum_reward": em[:, 0], "length": em[:, 1],
                                         **{name: em[:, 2 + i] for i, name in enumerate(cfg.HANDSTAND_METRICS)}})
        pd = cfg.HANDSTAND_PRIV_OBS_DIM
        if self.auto_reset:
            info["first_obs"] = {"state": v["first_obs"], "privileged_state": v["first_privileged_obs"][:, :pd]}
        st = State(pipeline_state=data, obs=v["obs"], reward=v["reward"][:, 0], done=v["done"][:, 0], metrics=metrics, info=info)
        st.data = data
        st.obs_dict = {"state": v["obs"], "privileged_state": v["privileged_obs"][:, :pd]}
        return st


class Footstand(Handstand):
    """reference go2/handstand.py:293-342: the mirrored task (stand on the hind feet)."""

    _variant = "footstand"


_ENVS = {"Go2JoystickFlatTerrain": (Joystick, dict(task="flat_terrain")),       # _src/locomotion/__init__.py:16-26
         "Go2JoystickRoughTerrain": (Joystick, dict(task="rough_terrain")),
         "Go2Handstand": (Handstand, {}), "Go2Footstand": (Footstand, {})}


def load(env_name: str, config: Optional[dict] = None, config_overrides: Optional[Dict[str, Any]] = None, **kw):
    """Counterpart of reference _src/registry.py:24-31."""
    if env_name not in _ENVS:
        raise ValueError(f"Env '{env_name}' not found. Available envs: {sorted(_ENVS)}")
    cls, args = _ENVS[env_name]
    return cls(config=config, config_overrides=config_overrides, **args, **kw)
