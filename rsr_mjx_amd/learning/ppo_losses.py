"""GAE and the PPO loss with the RSR term, on torch.  Counterpart of reference RSR/losses.py:39-205 (Brax PPO loss +
`sim2real_loss`) with Brax's NormalTanhDistribution (brax.training.distribution) restated below.

Networks are plain callables: `policy(obs) -> logits [.., 2A]` and `value(obs) -> [..]` (observation normalisation is the
caller's, folded into the callables).  Data is a rollout.Transition with leading dims [B, T] (the reference swaps to
time-major inside the loss, losses.py:141).
"""
from __future__ import annotations

import math
from typing import Any, Callable, Dict, Tuple

from . import rsr_loss as rsr

MIN_STD = 0.001        # brax NormalTanhDistribution(min_std=0.001)


def _split(logits):
    import torch
    loc, raw = torch.chunk(logits, 2, dim=-1)
    return loc, torch.nn.functional.softplus(raw) + MIN_STD


def _log_det_tanh(x):
    import torch
    return 2.0 * (math.log(2.0) - x - torch.nn.functional.softplus(-2.0 * x))


def tanh_normal_log_prob(logits, raw_action):
    """log p(tanh(raw)) summed over action dims: Normal.log_prob(raw) - log|d tanh / d raw|."""
    import torch
    loc, scale = _split(logits)
    lp = -0.5 * ((raw_action - loc) / scale) ** 2 - torch.log(scale) - 0.5 * math.log(2.0 * math.pi)
    return (lp - _log_det_tanh(raw_action)).sum(-1)


def tanh_normal_entropy(logits, noise):
    """brax ParametricDistribution.entropy(params, seed): Normal entropy + log|det J| at ONE sample (loc + scale * noise)."""
    import torch
    loc, scale = _split(logits)
    ent = 0.5 + 0.5 * math.log(2.0 * math.pi) + torch.log(scale)
    return (ent + _log_det_tanh(loc + scale * noise)).sum(-1)


def tanh_normal_mode(logits):
    import torch
    return torch.tanh(_split(logits)[0])


def compute_gae(truncation, termination, rewards, values, bootstrap_value, lambda_: float = 1.0, discount: float = 0.99):
    """losses.py:39-95; all inputs time-major [T, B]; returns (vs, advantages), both detached."""
    import torch
    mask = 1.0 - truncation
    v_tp1 = torch.cat([values[1:], bootstrap_value[None]], dim=0)
    deltas = (rewards + discount * (1.0 - termination) * v_tp1 - values) * mask
    acc = torch.zeros_like(bootstrap_value)
    out = []
    for t in range(values.shape[0] - 1, -1, -1):
        acc = deltas[t] + discount * (1.0 - termination[t]) * mask[t] * lambda_ * acc
        out.append(acc)
    vs = torch.stack(out[::-1], dim=0) + values
    vs_tp1 = torch.cat([vs[1:], bootstrap_value[None]], dim=0)
    adv = (rewards + discount * (1.0 - termination) * vs_tp1 - values) * mask
    return vs.detach(), adv.detach()


def compute_ppo_loss(policy: Callable, value: Callable, data, entropy_noise, past_data: Any = None, entropy_cost: float = 1e-4,
                     discounting: float = 0.9, reward_scaling: float = 1.0, gae_lambda: float = 0.95, clipping_epsilon: float = 0.3,
                     normalize_advantage: bool = True, rsr_loss_scale: float = 1.0) -> Tuple[Any, Dict[str, Any]]:
    """losses.py:98-205.  `entropy_noise`: standard normal [T, B, A] (the reference draws it from `rng`)."""
    import torch
    sw = lambda x: x.transpose(0, 1)                                   # time first
    obs, nobs, reward, discount = sw(data.observation), sw(data.next_observation), sw(data.reward), sw(data.discount)
    truncation = sw(data.extras["state_extras"]["truncation"])
    raw_action = sw(data.extras["policy_extras"]["raw_action"])
    behaviour_lp = sw(data.extras["policy_extras"]["log_prob"])
    logits = policy(obs)
    # asymmetric actor-critic (Playground network_factory value_obs_key="privileged_state"): the critic reads its own stream
    vobs = sw(data.extras["extra_obs"]["value"]) if "extra_obs" in data.extras and "value" in data.extras["extra_obs"] else obs
    vnobs = sw(data.extras["next_extra_obs"]["value"]) if "next_extra_obs" in data.extras and "value" in data.extras["next_extra_obs"] else nobs
    baseline = value(vobs)
    bootstrap = value(vnobs[-1])
    rewards = reward * reward_scaling
    termination = (1.0 - discount) * (1.0 - truncation)
    target_lp = tanh_normal_log_prob(logits, raw_action)
    vs, adv = compute_gae(truncation, termination, rewards, baseline, bootstrap, gae_lambda, discounting)
    if normalize_advantage:
        adv = (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-8)
    # exp overflows to inf for a log-ratio above ~88.7 (a collapsed policy scale), and inf * negative advantage turns the whole
    # update into NaN; capping the exponent at 80 changes the value only where the reference's is inf
    rho = torch.exp((target_lp - behaviour_lp).clamp(max=80.0))
    policy_loss = -torch.minimum(rho * adv, rho.clamp(1.0 - clipping_epsilon, 1.0 + clipping_epsilon) * adv).mean()
    v_err = vs - baseline
    v_loss = (v_err * v_err).mean() * 0.5 * 0.5
    entropy = tanh_normal_entropy(logits, entropy_noise).mean()
    entropy_loss = entropy_cost * -entropy
    task_loss = policy_loss + v_loss + entropy_loss
    # the action of the policy being optimised (not the rollout's): only then does the RSR term carry a policy gradient
    sim2real_loss, distance = rsr.compute_rsr_loss(obs, tanh_normal_mode(logits), nobs, past_data, loss_scale=rsr_loss_scale)
    total = task_loss + sim2real_loss
    return total, {"total_loss": total, "task_loss": task_loss, "policy_loss": policy_loss, "v_loss": v_loss,
                   "entropy_loss": entropy_loss, "sim2real_loss": sim2real_loss, "rsr_distribution_distance": distance}
