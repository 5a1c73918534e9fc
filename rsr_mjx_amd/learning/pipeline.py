"""RSR policy-tuning pipeline on the torch learners: the counterpart of reference RSR/rsr_pipeline.py:208-435
(`build_policy_rsr_data`, `policy_params_training`).  Step 3 of RSR (`env_params_tuning`) is rsr_mjx_amd/tuning.py."""
from __future__ import annotations

from typing import Any, Callable, Optional

from . import ppo_train, rsr_loss, sac_train


def build_policy_rsr_data(past_states, past_actions, past_next_states_real, past_next_states_sim, current_next_states_sim,
                          num_samples: int = 10, min_val: float = -3.0, max_val: float = 3.0, bandwidth: float = 0.1, seed: int = 0,
                          device=None) -> rsr_loss.RSRData:
    """Builds the fixed RSR statistics shared by PPO and SAC (rsr_pipeline.py:209-272)."""
    import torch
    arrays = tuple(torch.as_tensor(v, dtype=torch.float32, device=device) for v in
                   (past_states, past_actions, past_next_states_real, past_next_states_sim, current_next_states_sim))
    if any(v.dim() != 2 for v in arrays):
        raise ValueError(f"all RSR datasets must be rank 2, got {tuple(tuple(v.shape) for v in arrays)}")
    if len({v.shape[0] for v in arrays}) != 1:
        raise ValueError(f"RSR datasets must have equal lengths, got {tuple(tuple(v.shape) for v in arrays)}")
    if arrays[0].shape[0] == 0:
        raise ValueError("RSR datasets must not be empty")
    s, a, nr, ns, cs = arrays
    if nr.shape[1] != s.shape[1]:
        raise ValueError("real next-state width must match state width")
    if ns.shape[1] != s.shape[1]:
        raise ValueError("previous sim next-state width must match state width")
    if cs.shape[1] != s.shape[1]:
        raise ValueError("current sim next-state width must match state width")
    return rsr_loss.build_rsr_data(torch.cat([s, a, nr], 1), torch.cat([s, a, ns], 1), torch.cat([s, a, cs], 1), num_samples=num_samples,
                                   min_value=min_val, max_value=max_val, bandwidth=bandwidth, seed=seed)


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
    if rsr_loss_scale < 0:
        raise ValueError(f"rsr_loss_scale must be non-negative, got {rsr_loss_scale}")
    required = (past_states, past_actions, past_next_states_real, past_next_states_sim, current_next_states_sim)
    if any(v is None for v in required):
        raise ValueError("all five RSR policy datasets are required")
    past_data = build_policy_rsr_data(*required, num_samples=num_samples, min_val=min_val, max_val=max_val, bandwidth=bandwidth, seed=seed,
                                      device=device)
    progress_fn = progress_fn or (lambda *a: None)
    algorithm = algorithm.strip().lower()
    if algorithm == "ppo":
        mk, params, _ = ppo_train.train(env, num_timesteps, episode_length, past_data, action_repeat=action_repeat, num_envs=num_envs,
                                        num_eval_envs=num_eval_envs, learning_rate=learning_rate, entropy_cost=entropy_cost,
                                        discounting=discounting, seed=seed, unroll_length=unroll_length, batch_size=batch_size,
                                        num_minibatches=num_minibatches, num_updates_per_batch=num_updates_per_batch, num_evals=num_evals,
                                        normalize_observations=normalize_observations, reward_scaling=reward_scaling,
                                        rsr_loss_scale=rsr_loss_scale, deterministic_eval=deterministic_eval, progress_fn=progress_fn,
                                        randomization_fn=randomization_fn, wrap_fn=wrap_fn)
        return mk, params
    if algorithm == "sac":
        mk, params, _ = sac_train.train(env, num_timesteps, episode_length, past_data, action_repeat=action_repeat, num_envs=num_envs,
                                        num_eval_envs=num_eval_envs, learning_rate=learning_rate, discounting=discounting, seed=seed,
                                        batch_size=batch_size, num_evals=num_evals, normalize_observations=normalize_observations,
                                        reward_scaling=reward_scaling, tau=tau, min_replay_size=min_replay_size,
                                        max_replay_size=max_replay_size, grad_updates_per_step=grad_updates_per_step,
                                        deterministic_eval=deterministic_eval, progress_fn=progress_fn, randomization_fn=randomization_fn,
                                        rsr_loss_scale=rsr_loss_scale, wrap_fn=wrap_fn)
        return mk, params
    raise ValueError(f'unsupported algorithm {algorithm!r}; expected "ppo" or "sac"')
