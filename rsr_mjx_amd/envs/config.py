"""Env configuration -> `env_*` blob fields (the constants the reference bakes into its XLA program).

Airbot cube task: constructor defaults and index look-ups of
reference ppo_train/airbot_training/cube_env.py:9-94; reset constants :99-118, :127.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from ..mjcf import CompiledModel

ENV_CUBE, ENV_TSHAPE, ENV_AIRBOT_SF, ENV_GO2, ENV_GO2_HANDSTAND = 0, 1, 2, 3, 4
WRAP_EPISODE, WRAP_AUTORESET = 1, 2

CUBE_DEFAULTS = dict(
    push_reward_weight=6.0, endpoint_to_target_reward_weight=0.01, ctrl_cost_weight=0.003,
    box_still_cost_weight=0.01, joint_vel_cost_weight=0.1, siet_to_box_reward_weight=3.0,
    healthy_reward=1.0, max_vel=0.4, endpoint_min_z_pos=0.778, noise_scale=1e-2,
    cube_degree_cost_weight=0.01, coll_cost_weight=0.05, cube_vel_cost_weight=0.001,
    joint_num=7, decimation=4,
    cube_min_x=0.29, cube_max_x=0.34, cube_min_y=-0.04, cube_max_y=0.01,
    target_min_x=0.4364427, target_max_x=0.4864427, target_min_y=0.07352592, target_max_y=0.12352592,
)
CUBE_OBS_DIM = 23
CUBE_METRICS = ("push_reward", "ctrl_cost", "siet_to_box_reward")


# test/airbot.py:10-42: the variant every RSR script uses (model test/sf.xml)
SF_DEFAULTS = dict(CUBE_DEFAULTS, endpoint_min_z_pos=0.8, reset_action_scale=0.03,
                   cube_min_x=0.28, cube_max_x=0.29, cube_min_y=-0.005, cube_max_y=0.005,
                   target_min_x=0.5, target_max_x=0.51, target_min_y=-0.005, target_max_y=0.005)


def sf_env_fields(m: CompiledModel, episode_length: int = 0, auto_reset: bool = False, **kwargs) -> Dict[str, np.ndarray]:
    return cube_env_fields(m, episode_length, auto_reset, _kind=ENV_AIRBOT_SF, **kwargs)


def cube_env_fields(m: CompiledModel, episode_length: int = 0, auto_reset: bool = False, _kind: int = ENV_CUBE,
                    **kwargs) -> Dict[str, np.ndarray]:
    kind = _kind
    cfg = dict(SF_DEFAULTS if kind == ENV_AIRBOT_SF else CUBE_DEFAULTS)
    n_frames = kwargs.pop("n_frames", None)
    unknown = set(kwargs) - set(cfg)
    if unknown:
        raise TypeError(f"unknown AirbotPlayBase arguments: {sorted(unknown)}")
    cfg.update(kwargs)
    if n_frames is None:
        n_frames = cfg["decimation"]          # cube_env.py:39-40
    A = m.arrays
    jq = [int(A["jnt_qposadr"][m.id("joint", f"joint{i}")]) for i in range(1, 7)]
    cube_b, target_b = m.id("body", "cube_for_push"), m.id("body", "target_pos")
    box_q = int(A["jnt_qposadr"][A["body_jntadr"][cube_b]])
    tgt_q = int(A["jnt_qposadr"][A["body_jntadr"][target_b]])
    finger_q = int(A["jnt_qposadr"][m.id("joint", "endleft")])
    ids = np.array([cube_b, target_b, m.id("site", "endpoint"), box_q, tgt_q, finger_q] + jq, dtype=np.int32)
    flags = (WRAP_EPISODE if episode_length > 0 else 0) | (WRAP_AUTORESET if auto_reset else 0)
    reset = np.array(
        [cfg["noise_scale"],
         0, -0.5422302, 0.45173569, 1.5718, -1.4794435, 1.1731174,      # cube_env.py:102
         -0.033,                                                          # :103
         0, -0.73151061, 0.455936904, -1.4794435, 1.1731174,             # :107
         cfg["target_min_x"], cfg["target_min_y"], 0.82, cfg["target_max_x"], cfg["target_max_y"], 0.82,
         cfg["cube_min_x"], cfg["cube_min_y"], 0.82, cfg["cube_max_x"], cfg["cube_max_y"], 0.82,
         0.37342, -0.07989],                                              # :127
        dtype=np.float32)
    # [push weight, siet weight, healthy reward, endpoint_min_z, "at target" threshold, task-complete bonus]
    reward = np.array([cfg["push_reward_weight"], cfg["siet_to_box_reward_weight"], cfg["healthy_reward"],
                       cfg["endpoint_min_z_pos"], 0.003 if kind == ENV_AIRBOT_SF else 0.005,
                       5.0 if kind == ENV_AIRBOT_SF else 0.0], dtype=np.float32)
    return dict(
        env_int=np.array([kind, n_frames, episode_length, flags, CUBE_OBS_DIM, len(CUBE_METRICS)], dtype=np.int32),
        env_ids=ids,
        env_action_scale=np.array([0.02, 0.02, 0.02, 0.0, 0.0], dtype=np.float32),   # cube_env.py:60
        env_ctrl_lo=A["actuator_ctrlrange"][:, 0].astype(np.float32),                 # :63-64
        env_ctrl_hi=A["actuator_ctrlrange"][:, 1].astype(np.float32),
        env_reset=reset,
        env_reward=reward,
    )


# ---------------------------------------------------------------------------------------------------------
# T-shape task: reference ppo_train/airbot_training/T_shape_env.py:11-97 (constructor), :98-137 (reset constants)
TSHAPE_DEFAULTS = dict(CUBE_DEFAULTS, push_reward_weight=10.0, endpoint_min_z_pos=0.78)
TSHAPE_OBS_DIM = 16
TSHAPE_METRICS = ("push_reward", "siet2cube_reward", "health_reward", "task_complete_reward", "site_z_reward")


def tshape_env_fields(m: CompiledModel, episode_length: int = 0, auto_reset: bool = False, **kwargs) -> Dict[str, np.ndarray]:
    cfg = dict(TSHAPE_DEFAULTS)
    n_frames = kwargs.pop("n_frames", None)
    unknown = set(kwargs) - set(cfg)
    if unknown:
        raise TypeError(f"unknown AirbotPlayBase (T-shape) arguments: {sorted(unknown)}")
    cfg.update(kwargs)
    if n_frames is None:
        n_frames = cfg["decimation"]
    A = m.arrays
    jq = [int(A["jnt_qposadr"][m.id("joint", f"joint{i}")]) for i in range(1, 7)]
    ids = np.array([m.id("body", "T_block"), m.id("body", "T_target"), m.id("site", "endpoint"), m.id("site", "T_tail"),
                    m.id("site", "T_target_tail"), m.id("geom", "base_block"), m.id("geom", "vertical_block"),
                    m.id("geom", "base_target"), m.id("geom", "vertical_target")] + jq, dtype=np.int32)
    flags = (WRAP_EPISODE if episode_length > 0 else 0) | (WRAP_AUTORESET if auto_reset else 0)
    reset = np.array([cfg["noise_scale"],
                      0, -0.57303354, 0.381795, 1.5718, -1.3787, 1.1731174,     # T_shape_env.py:105
                      0, -0.57303354, 0.381795, -1.3787, 1.1731174,             # :109
                      0.24739072, -0.00496255,                                   # :116
                      0.2876], dtype=np.float32)                                 # :132
    reward = np.array([cfg["push_reward_weight"], cfg["siet_to_box_reward_weight"], cfg["healthy_reward"],
                       cfg["endpoint_min_z_pos"]], dtype=np.float32)
    return dict(
        env_int=np.array([ENV_TSHAPE, n_frames, episode_length, flags, TSHAPE_OBS_DIM, len(TSHAPE_METRICS)], dtype=np.int32),
        env_ids=ids,
        env_action_scale=np.array([0.02, 0.02, 0.02, 0.0, 0.0], dtype=np.float32),
        env_ctrl_lo=A["actuator_ctrlrange"][:, 0].astype(np.float32),
        env_ctrl_hi=A["actuator_ctrlrange"][:, 1].astype(np.float32),
        env_reset=reset,
        env_reward=reward,
    )


# ---------------------------------------------------------------------------------------------------------
# Go2 joystick task: reference mujoco_playground/_src/locomotion/go2/joystick.py:13-82 (default_config)
GO2_OBS_DIM = 48
GO2_REWARDS = ("tracking_lin_vel", "tracking_ang_vel", "lin_vel_z", "ang_vel_xy", "orientation", "dof_pos_limits", "pose",
               "termination", "stand_still", "torques", "action_rate", "energy", "feet_clearance", "feet_height", "feet_slip",
               "feet_air_time", "all_feet_air", "symmetric_gait", "lr_symmetry", "fb_symmetry", "feet_off_ground_when_still")
GO2_METRICS = tuple(f"reward/{k}" for k in GO2_REWARDS) + ("swing_peak",)
GO2_INFO_FLOATS = 144
GO2_DEFAULT_CONFIG = dict(
    ctrl_dt=0.02, sim_dt=0.004, episode_length=1000, Kp=60.0, Kd=3.0, action_repeat=1, action_scale=0.5, history_len=1,
    soft_joint_pos_limit_factor=0.95,
    noise_config=dict(level=1.0, scales=dict(joint_pos=0.03, joint_vel=1.5, gyro=0.2, gravity=0.05, linvel=0.1)),
    reward_config=dict(
        scales=dict(tracking_lin_vel=3.0, tracking_ang_vel=1.5, lin_vel_z=-0.5, ang_vel_xy=-0.05, orientation=-3.0,
                    dof_pos_limits=-1.0, pose=0.0, termination=-1.0, stand_still=-1.0, torques=-0.0002, action_rate=-0.01,
                    energy=-0.001, feet_clearance=-2.0, feet_height=-3.5, feet_slip=-0.1, feet_air_time=0.8,
                    all_feet_air=-1.0, symmetric_gait=-0.8, lr_symmetry=-0.8, fb_symmetry=-0.8,
                    feet_off_ground_when_still=-1.0),
        tracking_sigma=0.25, max_foot_height=0.12),
    pert_config=dict(enable=False, velocity_kick=[0.0, 3.0], kick_durations=[0.05, 0.2], kick_wait_times=[1.0, 3.0]),
    command_config=dict(a=[0.8, 0.0, 2.0], b=[0.8, 0.0, 0.8], change_interval=12.0),
    delay_config=dict(action=dict(enable=True, steps=3), imu=dict(enable=True, steps=3)),
)


def _merge(base: dict, over: dict) -> dict:
    out = {k: (dict(v) if isinstance(v, dict) else v) for k, v in base.items()}
    for k, v in (over or {}).items():
        if k not in out:
            raise KeyError(f"unknown Go2 config key {k!r}")     # the reference ConfigDict is locked (_src/mjx_env.py:104-106)
        out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out[k], dict) else v
    return out


def go2_apply_overrides(m: CompiledModel, config: dict) -> CompiledModel:
    """go2/base.py:25-31: timestep, dof_damping[6:] = Kd, position servos gain/bias = +-Kp."""
    A = {k: v.copy() for k, v in m.arrays.items()}
    A["opt_timestep"][0] = config["sim_dt"]
    A["dof_damping"][6:] = config["Kd"]
    A["actuator_gainprm"][:, 0] = config["Kp"]
    A["actuator_biasprm"][:, 1] = -config["Kp"]
    return CompiledModel(name=m.name, arrays=A, names=m.names)


def _subtree_mass(m: CompiledModel, body: int) -> float:
    par, mass = m.arrays["body_parentid"], m.arrays["body_mass"]
    total = 0.0
    for b in range(body, m.nbody):
        a = b
        while a > body:
            a = int(par[a])
        if a == body:
            total += float(mass[b])
    return total


GO2_PRIV_OBS_DIM = 123      # obs["privileged_state"], joystick.py:341-366


def go2_env_fields(m: CompiledModel, config: dict, episode_length: int = 0, auto_reset: bool = False) -> Dict[str, np.ndarray]:
    """`m` must already carry the base.py overrides (go2_apply_overrides)."""
    # (config["action_repeat"] is the trainer's: it reaches the Episode wrapper through wrap_for_brax_training, not the env)
    A = m.arrays
    n_sub = int(round(config["ctrl_dt"] / config["sim_dt"]))      # _src/mjx_env.py:139-142
    home = A["key_qpos"][m.names["key"]["home"]].astype(np.float32)
    lo, hi = A["jnt_range"][1:, 0], A["jnt_range"][1:, 1]
    soft = np.concatenate([lo, hi]).astype(np.float32) * np.float32(config["soft_joint_pos_limit_factor"])
    feet = ["FR", "FL", "RR", "RL"]                                # go2_constants.py:20-31
    ids = np.array([m.id("site", "imu")] + [m.id("site", f) for f in feet] + [m.id("geom", "floor")] +
                   [m.id("geom", f) for f in feet] + [m.id("body", "trunk")], dtype=np.int32)
    nz, rc, cc, pc, dc = (config[k] for k in ("noise_config", "reward_config", "command_config", "pert_config", "delay_config"))
    f = np.array([config["ctrl_dt"], config["action_scale"], nz["level"], nz["scales"]["joint_pos"], nz["scales"]["joint_vel"],
                  nz["scales"]["gyro"], nz["scales"]["gravity"], nz["scales"]["linvel"], rc["tracking_sigma"], rc["max_foot_height"],
                  *cc["a"], *cc["b"], cc["change_interval"], *pc["kick_wait_times"], *pc["kick_durations"], *pc["velocity_kick"],
                  _subtree_mass(m, m.id("body", "trunk"))],      # joystick.py:104 body_subtreemass[torso]
                 dtype=np.float32)
    flags = (WRAP_EPISODE if episode_length > 0 else 0) | (WRAP_AUTORESET if auto_reset else 0)
    return dict(
        env_int=np.array([ENV_GO2, n_sub, episode_length, flags, GO2_OBS_DIM, len(GO2_METRICS)], dtype=np.int32),
        env_ids=ids,
        env_go2f=f,
        env_go2i=np.array([dc["action"]["steps"] if dc["action"]["enable"] else 0,
                           dc["imu"]["steps"] if dc["imu"]["enable"] else 0, int(pc["enable"])], dtype=np.int32),
        env_go2_scales=np.array([rc["scales"][k] for k in GO2_REWARDS], dtype=np.float32),
        env_go2_home=home,
        env_go2_soft=soft,
        # unused by this env kind but looked up by the generic loaders
        env_action_scale=np.zeros(m.nu, np.float32), env_ctrl_lo=A["actuator_ctrlrange"][:, 0].astype(np.float32),
        env_ctrl_hi=A["actuator_ctrlrange"][:, 1].astype(np.float32), env_reset=np.zeros(1, np.float32),
        env_reward=np.zeros(1, np.float32),
    )


# ---------------------------------------------------------------------------------------------------------
# Go2 Handstand / Footstand: reference mujoco_playground/_src/locomotion/go2/handstand.py:13-52 (default_config), :53-118 (_post_init),
# :293-342 (Footstand).  Model: scene_mjx_flat_terrain.xml -> go2_mjx.xml (every collision geom of the robot against the floor:
# 4 spheres, 20 capsules, 6 cylinders; assets/go2_full.npz).
HANDSTAND_OBS_DIM = 45
HANDSTAND_PRIV_OBS_DIM = 94
# config order of reward_config.scales = order of the metrics dict (handstand.py:36-50, :144-146)
HANDSTAND_REWARDS = ("height", "orientation", "contact", "action_rate", "termination", "dof_pos_limits", "torques", "pose",
                     "stay_still", "energy", "dof_acc")
HANDSTAND_METRICS = tuple(f"reward/{k}" for k in HANDSTAND_REWARDS)
HANDSTAND_DEFAULT_CONFIG = dict(
    ctrl_dt=0.02, sim_dt=0.004, episode_length=500, Kp=35.0, Kd=0.5, action_repeat=1, action_scale=0.3,
    soft_joint_pos_limit_factor=0.9, init_from_crouch=0.0, energy_termination_threshold=float("inf"),
    noise_config=dict(level=1.0, scales=dict(joint_pos=0.01, joint_vel=1.5, gyro=0.2, gravity=0.05, linvel=0.1)),
    reward_config=dict(scales=dict(height=1.0, orientation=1.0, contact=-0.1, action_rate=0.0, termination=0.0, dof_pos_limits=-0.5,
                                   torques=0.0, pose=-0.1, stay_still=0.0, energy=0.0, dof_acc=0.0)),
)
_HANDSTAND_VARIANTS = {
    # unwanted-contact geoms, feet geoms of the contact cost, joint ids of the pose cost, desired forward vector, desired height
    "handstand": (("fl_calf1", "fl_calf2", "fr_calf1", "fr_calf2", "fl_thigh1", "fl_thigh2", "fl_thigh3", "fr_thigh1", "fr_thigh2",
                   "fr_thigh3", "fl_hip", "fr_hip"), ("RR", "RL"), (6, 7, 8, 9, 10, 11), (0.0, 0.0, -1.0), 0.55),
    "footstand": (("rl_calf1", "rl_calf2", "rr_calf1", "rr_calf2", "rl_thigh1", "rl_thigh2", "rl_thigh3", "rr_thigh1", "rr_thigh2",
                   "rr_thigh3", "rl_hip", "rr_hip"), ("FR", "FL"), (0, 1, 2, 3, 4, 5), (0.0, 0.0, 1.0), 0.53),
}


def handstand_env_fields(m: CompiledModel, config: dict, episode_length: int = 0, auto_reset: bool = False,
                         variant: str = "handstand") -> Dict[str, np.ndarray]:
    """`m` must already carry the base.py overrides (go2_apply_overrides).
    env_ids: imu site, floor geom, the twelve unwanted-contact geoms, the two feet geoms of the contact cost, trunk body.
    env_go2f: ctrl_dt, action_scale, noise level, scales joint_pos / joint_vel / gyro / gravity / linvel, init_from_crouch,
    energy_termination_threshold, z_des, desired forward vector (3).  env_go2i: joint ids of the pose cost.  env_go2_scales: the eleven
    reward scales in config order.  env_go2_home: home qpos | pre_recovery qpos.  env_go2_soft: soft lower | soft upper limits."""
    # (config["action_repeat"] is the trainer's: it reaches the Episode wrapper through wrap_for_brax_training, not the env)
    unwanted, feet, joint_ids, fwd, z_des = _HANDSTAND_VARIANTS[variant]
    A = m.arrays
    n_sub = int(round(config["ctrl_dt"] / config["sim_dt"]))
    home = A["key_qpos"][m.names["key"]["home"]].astype(np.float32)
    crouch = A["key_qpos"][m.names["key"]["pre_recovery"]].astype(np.float32)
    lo, hi = A["jnt_range"][1:, 0], A["jnt_range"][1:, 1]                      # handstand.py:74-78
    c, r = (lo + hi) / 2, hi - lo
    f = config["soft_joint_pos_limit_factor"]
    soft = np.concatenate([c - 0.5 * r * f, c + 0.5 * r * f]).astype(np.float32)
    ids = np.array([m.id("site", "imu"), m.id("geom", "floor")] + [m.id("geom", g) for g in unwanted] + [m.id("geom", g) for g in feet] +
                   [m.id("body", "trunk")], dtype=np.int32)
    nz, rc = config["noise_config"], config["reward_config"]
    thr = config["energy_termination_threshold"]
    fl = np.array([config["ctrl_dt"], config["action_scale"], nz["level"], nz["scales"]["joint_pos"], nz["scales"]["joint_vel"],
                   nz["scales"]["gyro"], nz["scales"]["gravity"], nz["scales"]["linvel"], config["init_from_crouch"],
                   np.finfo(np.float32).max if not np.isfinite(thr) else thr, z_des, *fwd], dtype=np.float32)
    flags = (WRAP_EPISODE if episode_length > 0 else 0) | (WRAP_AUTORESET if auto_reset else 0)
    return dict(
        env_int=np.array([ENV_GO2_HANDSTAND, n_sub, episode_length, flags, HANDSTAND_OBS_DIM, len(HANDSTAND_METRICS)], dtype=np.int32),
        env_ids=ids,
        env_go2f=fl,
        env_go2i=np.array(joint_ids, dtype=np.int32),
        env_go2_scales=np.array([rc["scales"][k] for k in HANDSTAND_REWARDS], dtype=np.float32),
        env_go2_home=np.concatenate([home, crouch]),
        env_go2_soft=soft,
        env_action_scale=np.zeros(m.nu, np.float32), env_ctrl_lo=A["actuator_ctrlrange"][:, 0].astype(np.float32),
        env_ctrl_hi=A["actuator_ctrlrange"][:, 1].astype(np.float32), env_reset=np.zeros(1, np.float32),
        env_reward=np.zeros(1, np.float32),
    )
