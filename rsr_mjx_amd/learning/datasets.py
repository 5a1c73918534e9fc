"""The six comma-separated tables an RSR policy-tuning run starts from, and the five transition sets made of them.

Counterpart of reference test/rsr_policy_training.py:70-209 (`load_rsr_datasets`): one directory holds
    real_obs.txt, real_action.txt      the real system's observation / action sequence
    past_sim_obs.txt                   the previous simulator replaying those actions
    current_sim_obs.txt                the current simulator replaying them
    obs.txt, actions.txt               the simulator's own rollout (checked for presence and shape only, as the reference does)
A sequence of T+1 observations and T actions gives T transitions (s_t, a_t, s_{t+1}); every table is cut to the count the
real pair allows (at most `max_transitions`).  Returns numpy float arrays; learning.pipeline moves them to the device.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Tuple

import numpy as np

REQUIRED_DATA_FILES = ("real_obs.txt", "real_action.txt", "past_sim_obs.txt", "current_sim_obs.txt", "obs.txt", "actions.txt")
_OBS_TABLES = ("real_obs.txt", "past_sim_obs.txt", "current_sim_obs.txt", "obs.txt")
_ACTION_TABLES = ("real_action.txt", "actions.txt")


def load_table(path) -> np.ndarray:
    """A comma-separated numeric file as a 2-D array (one row per time step; a single line is one row)."""
    path = Path(path)
    with open(path) as f:
        rows = [ln for ln in f if ln.strip()]
    if not rows:
        raise ValueError(f"{path.name}: no rows of numbers in this file")
    table = np.loadtxt(rows, delimiter=",", ndmin=2)
    return table


def load_rsr_datasets(data_dir, max_transitions: int = 50, verbose: bool = False) -> Tuple[np.ndarray, ...]:
    """(past_states, past_actions, past_next_states_real, past_next_states_sim, current_next_states_sim), the positional
    dataset arguments of `policy_params_training` (rsr_pipeline.py:275-283)."""
    data_dir = Path(data_dir)
    missing = [name for name in REQUIRED_DATA_FILES if not (data_dir / name).is_file()]
    if missing:
        raise FileNotFoundError(f"{data_dir}: missing {', '.join(missing)} (an RSR data directory holds {', '.join(REQUIRED_DATA_FILES)})")
    tables: Dict[str, np.ndarray] = {name: load_table(data_dir / name) for name in REQUIRED_DATA_FILES}

    real_obs, real_action = tables["real_obs.txt"], tables["real_action.txt"]
    count = min(len(real_obs) - 1, len(real_action), int(max_transitions))
    if count <= 0:
        raise ValueError(f"real_obs.txt ({len(real_obs)} rows) and real_action.txt ({len(real_action)} rows) give no transition: "
                         "a transition takes two consecutive observations and the action between them")
    obs_dim, action_dim = real_obs.shape[1], real_action.shape[1]

    for name in _OBS_TABLES[1:]:
        if len(tables[name]) < count + 1:
            raise ValueError(f"{name}: {len(tables[name])} rows, but {count} transitions take {count + 1} observations")
    if len(tables["actions.txt"]) < count:
        raise ValueError(f"actions.txt: {len(tables['actions.txt'])} rows, but {count} transitions take {count} actions")
    for names, width, label in ((_OBS_TABLES, obs_dim, "observation"), (_ACTION_TABLES, action_dim, "action")):
        for name in names:
            if tables[name].shape[1] != width:
                raise ValueError(f"{name}: {tables[name].shape[1]} columns where the real data's {label}s have {width}")

    if verbose:
        print(f"RSR data from {data_dir}: {count} transitions, {obs_dim}-dim observations, {action_dim}-dim actions")
        for name in REQUIRED_DATA_FILES:
            print(f"  {name}: {tables[name].shape[0]} x {tables[name].shape[1]}")
    return (real_obs[:count], real_action[:count], real_obs[1:count + 1],
            tables["past_sim_obs.txt"][1:count + 1], tables["current_sim_obs.txt"][1:count + 1])
