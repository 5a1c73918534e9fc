"""SAC losses with the RSR term in the actor objective, on torch.  Counterpart of reference RSR/sac_losses.py:23-130.
`policy(obs) -> logits`, `q(obs, action) -> [.., 2]` (twin critics) are plain callables; noise tensors are standard normal
draws of the action shape (the reference samples them from `key`)."""
from __future__ import annotations

from typing import Any, Callable

from . import rsr_loss as rsr
from .ppo_losses import _split, tanh_normal_log_prob


def make_losses(policy: Callable, q_network: Callable, reward_scaling: float, discounting: float, action_size: int, *,
                past_data: Any = None, rsr_loss_scale: float = 1.0):
    import torch
    target_entropy = -0.5 * action_size

    def _sample(logits, noise):
        loc, scale = _split(logits)
        return loc + scale * noise

    def alpha_loss(log_alpha, transitions, noise):
        """temperature loss (SAC eq. 18), :40-55"""
        with torch.no_grad():
            logits = policy(transitions.observation)
            raw = _sample(logits, noise)
            log_prob = tanh_normal_log_prob(logits, raw)
        return (torch.exp(log_alpha) * (-log_prob - target_entropy)).mean()

    def critic_loss(q, target_q, alpha, transitions, noise):
        """twin-Q Bellman loss, :57-98"""
        old_q = q(transitions.observation, transitions.action)
        with torch.no_grad():
            nlogits = policy(transitions.next_observation)
            nraw = _sample(nlogits, noise)
            nlp = tanh_normal_log_prob(nlogits, nraw)
            next_q = target_q(transitions.next_observation, torch.tanh(nraw))
            next_v = next_q.min(dim=-1).values - alpha * nlp
            tq = transitions.reward * reward_scaling + transitions.discount * discounting * next_v
        err = (old_q - tq[..., None]) * (1.0 - transitions.extras["state_extras"]["truncation"])[..., None]
        return 0.5 * (err * err).mean()

    def actor_loss(q, alpha, transitions, noise):
        """entropy-regularised actor loss + the differentiable RSR penalty, :100-128"""
        logits = policy(transitions.observation)
        raw = _sample(logits, noise)
        log_prob = tanh_normal_log_prob(logits, raw)
        action = torch.tanh(raw)
        base = (alpha * log_prob - q(transitions.observation, action).min(dim=-1).values).mean()
        s2r, _ = rsr.compute_rsr_loss(transitions.observation, action, transitions.next_observation, past_data, loss_scale=rsr_loss_scale)
        return base + s2r

    return alpha_loss, critic_loss, actor_loss
