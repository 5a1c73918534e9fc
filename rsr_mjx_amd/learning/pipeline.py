"""RSR policy-tuning pipeline on the torch learners: the counterpart of reference RSR/rsr_pipeline.py:208-435
(`build_policy_rsr_data`, `policy_params_training`).  Step 3 of RSR (`env_params_tuning`) is rsr_mjx_amd/tuning.py."""
from __future__ import annotations

from typing import Any, Callable, Optional

from . import ppo_train, rsr_loss, sac_train


def build_policy_rsr_data(past_states, past_actions, past_next_states_real, past_next_states_sim, current_next_states_sim,
                          num_samples: int = 10, min_val: float = -3.0, max_val: float = 3.0, bandwidth: float = 0.1, seed: int = 0,
                          device=None) -> rsr_loss.RSRData:
    """The fixed RSR statistics shared by PPO and SAC (rsr_pipeline.py:209-272): the (s, a, s') tables of the real system, the
    previous simulator and the current simulator, built from one set of states / actions and three sets of next states."""
    import torch
    cols = {"past_states": past_states, "past_actions": past_actions, "past_next_states_real": past_next_states_real,
            "past_next_states_sim": past_next_states_sim, "current_next_states_sim": current_next_states_sim}
    t = {k: torch.as_tensor(v, dtype=torch.float32, device=device) for k, v in cols.items()}
    shapes = {k: tuple(v.shape) for k, v in t.items()}
    rows = {sh[0] for sh in shapes.values() if len(sh) == 2}
    state_w = shapes["past_states"][1] if len(shapes["past_states"]) == 2 else None
    ok = (all(len(sh) == 2 for sh in shapes.values()) and len(rows) == 1 and rows != {0}
          and all(shapes[k][1] == state_w for k in ("past_next_states_real", "past_next_states_sim", "current_next_states_sim")))
    if not ok:
        raise ValueError("RSR datasets: five non-empty 2-D tables with equal row counts, the three next-state tables as wide as the states; "
                         f"got {shapes}")
    table = lambda nxt: torch.cat([t["past_states"], t["past_actions"], t[nxt]], dim=1)
    return rsr_loss.build_rsr_data(table("past_next_states_real"), table("past_next_states_sim"), table("current_next_states_sim"),
                                   num_samples=num_samples, min_value=min_val, max_value=max_val, bandwidth=bandwidth, seed=seed)


def policy_params_training(env, progress_fn: Optional[Callable[..., None]] = None, past_states: Any = None, past_actions: Any = None,
                           past_next_states_real: Any = None, past_next_states_sim: Any = None, current_next_states_sim: Any = None,
                           algorithm: str = "ppo", num_samples: int = 10, min_val: float = -3.0, max_val: float = 3.0, bandwidth: float = 0.1,
                           rsr_loss_scale: float = 1.0, num_timesteps: int = 5_000_000, num_evals: int = 10, reward_scaling: float = 0.1,
                           episode_length: int = 1200, normalize_observations: bool = True, action_repeat: int = 1, discounting: float = 0.96,
                           learning_rate: float = 1e-4, num_envs: int = 512, batch_size: int = 128, seed: int = 0, num_eval_envs: int = 128,
                           deterministic_eval: bool = False, unroll_length: int = 10, num_minibatches: int = 32, num_updates_per_batch: int = 8,
                           entropy_cost: float = 2e-2, tau: float = 0.005, min_replay_size: int = 0, max_replay_size: Optional[int] = None,
                           grad_updates_per_step: int = 1, randomization_fn=None, wrap_fn: Optional[Callable] = None, device: Optional[str] = "cuda"):
    """Trains an RSR policy with PPO or SAC (rsr_pipeline.py:275-435).  Returns (make_inference_fn, params)."""
    datasets = (past_states, past_actions, past_next_states_real, past_next_states_sim, current_next_states_sim)
    if rsr_loss_scale < 0 or any(d is None for d in datasets):
        raise ValueError(f"policy_params_training needs the five RSR datasets and rsr_loss_scale >= 0 (got scale {rsr_loss_scale}, "
                         f"{sum(d is None for d in datasets)} dataset(s) missing)")
    past_data = build_policy_rsr_data(*datasets, num_samples=num_samples, min_val=min_val, max_val=max_val, bandwidth=bandwidth, seed=seed,
                                      device=device)
    shared = dict(action_repeat=action_repeat, num_envs=num_envs, num_eval_envs=num_eval_envs, learning_rate=learning_rate,
                  discounting=discounting, seed=seed, batch_size=batch_size, num_evals=num_evals,
                  normalize_observations=normalize_observations, reward_scaling=reward_scaling, rsr_loss_scale=rsr_loss_scale,
                  deterministic_eval=deterministic_eval, progress_fn=progress_fn or (lambda *a: None), randomization_fn=randomization_fn,
                  wrap_fn=wrap_fn)
    learners = {
        "ppo": (ppo_train.train, dict(entropy_cost=entropy_cost, unroll_length=unroll_length, num_minibatches=num_minibatches,
                                      num_updates_per_batch=num_updates_per_batch)),
        "sac": (sac_train.train, dict(tau=tau, min_replay_size=min_replay_size, max_replay_size=max_replay_size,
                                      grad_updates_per_step=grad_updates_per_step)),
    }
    name = algorithm.strip().lower()
    if name not in learners:
        raise ValueError(f"algorithm {algorithm!r}: the pipeline has {sorted(learners)}")
    train, own = learners[name]
    make_policy, params, _metrics = train(env, num_timesteps, episode_length, past_data, **shared, **own)
    return make_policy, params
