"""Env configuration -> `env_*` blob fields (the constants the reference bakes into its XLA program).

Airbot cube task: constructor defaults and index look-ups of
reference ppo_train/airbot_training/cube_env.py:9-94; reset constants :99-118, :127.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from ..mjcf import CompiledModel

ENV_CUBE, ENV_TSHAPE, ENV_AIRBOT_SF, ENV_GO2 = 0, 1, 2, 3
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
