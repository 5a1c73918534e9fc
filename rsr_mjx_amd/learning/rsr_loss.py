"""RSR transition-distribution loss on torch (differentiable w.r.t. the policy's actions).

Counterpart of reference RSR/rsr_loss.py:15-175 and RSR/dataset_processor.py:17-43:
  grid      = jax.random.uniform(PRNGKey(seed), (num_samples, D), minval, maxval)         (rsr_loss.py:26-40)
  kde(data) = softmax_m( logsumexp_n( -|grid_m - data_n|^2 / (2 h^2) ) - log N )           (dataset_processor.py:17-34)
  KL(p, q)  = sum p log((p + 1e-10) / (q + 1e-10))                                         (:36-38)
  W(p, q)   = sum | cumsum(p) - cumsum(q) |                                                (:40-42)
  loss      = loss_scale * KL(real, previous_sim) * W( kde(reference U online), kde(reference) )   (rsr_loss.py:122-175)
The grid comes from this build's threefry restatement (rsr_mjx_amd/prng.py), so it is the grid jax would draw.
"""
from __future__ import annotations

from typing import Any, NamedTuple, Tuple

import numpy as np

from .. import prng


class RSRData(NamedTuple):
    """Precomputed real/sim distribution statistics used during training (rsr_loss.py:15-23)."""
    divergence: Any
    reference_density: Any
    reference_data: Any
    grid: Any
    bandwidth: float


def make_grid(num_samples: int, dimension: int, min_value: float = -3.0, max_value: float = 3.0, seed: int = 0, device=None, dtype=None):
    import torch
    g = prng.uniform(prng.PRNGKey(seed), (num_samples, dimension), min_value, max_value)      # float32 draws, as jax
    return torch.as_tensor(g, device=device).to(dtype or torch.float32)


def evaluate_kde(data, grid, bandwidth: float = 0.1):
    """[N, D] data, [M, D] grid -> [M] probabilities.  The (M, N) kernel matrix is |g|^2 + |x|^2 - 2 g.x^T: one GEMM on the
    matrix cores instead of the reference's (M, N, D) difference tensor."""
    import torch
    g2 = (grid * grid).sum(-1, keepdim=True)                       # [M, 1]
    x2 = (data * data).sum(-1)[None, :]                            # [1, N]
    sq = (g2 + x2 - 2.0 * grid @ data.T).clamp_min(0.0)
    log_kernel = -sq / (2.0 * bandwidth ** 2)
    log_pdf = torch.logsumexp(log_kernel, dim=-1) - float(np.log(data.shape[0]))
    return torch.softmax(log_pdf, dim=0)


def kl_divergence(p, q):
    return (p * torch_log((p + 1e-10) / (q + 1e-10))).sum()


def torch_log(x):
    import torch
    return torch.log(x)


def wasserstein_distance(p, q):
    import torch
    return (torch.cumsum(p, 0) - torch.cumsum(q, 0)).abs().sum()


def build_rsr_data(real_data, previous_sim_data, current_sim_data, *, num_samples: int = 10, min_value: float = -3.0,
                   max_value: float = 3.0, bandwidth: float = 0.1, seed: int = 0) -> RSRData:
    if real_data.dim() != 2:
        raise ValueError(f"real_data must be rank 2, got shape {tuple(real_data.shape)}")
    if previous_sim_data.shape != real_data.shape:
        raise ValueError(f"previous_sim_data must match real_data: {tuple(previous_sim_data.shape)} != {tuple(real_data.shape)}")
    if current_sim_data.shape != real_data.shape:
        raise ValueError(f"current_sim_data must match real_data: {tuple(current_sim_data.shape)} != {tuple(real_data.shape)}")
    if num_samples <= 0:
        raise ValueError(f"num_samples must be positive, got {num_samples}")
    if bandwidth <= 0:
        raise ValueError(f"bandwidth must be positive, got {bandwidth}")
    grid = make_grid(num_samples, real_data.shape[-1], min_value, max_value, seed, device=real_data.device, dtype=real_data.dtype)
    real_density = evaluate_kde(real_data, grid, bandwidth)
    previous_sim_density = evaluate_kde(previous_sim_data, grid, bandwidth)
    reference_density = evaluate_kde(current_sim_data, grid, bandwidth)
    return RSRData(divergence=kl_divergence(real_density, previous_sim_density), reference_density=reference_density,
                   reference_data=current_sim_data, grid=grid, bandwidth=bandwidth)


def _as_rsr_data(past_data: Any) -> RSRData:
    """the RSRData format and the legacy 3-tuple (KLD, density, reference_data) (rsr_loss.py:92-119)."""
    if isinstance(past_data, RSRData):
        return past_data
    if not isinstance(past_data, (tuple, list)):
        raise TypeError("past_data must be RSRData or a tuple/list")
    if len(past_data) == 5:
        return RSRData(*past_data)
    if len(past_data) != 3:
        raise ValueError("legacy past_data must contain (KLD, density, reference_data)")
    divergence, reference_density, reference_data = past_data
    grid = make_grid(int(reference_density.shape[0]), int(reference_data.shape[-1]), device=reference_data.device, dtype=reference_data.dtype)
    return RSRData(divergence, reference_density, reference_data, grid, 0.1)


def compute_rsr_loss(observations, policy_actions, next_observations, past_data: Any, *, loss_scale: float = 1.0) -> Tuple[Any, Any]:
    """(scaled_loss, distribution_distance); any number of leading dims on the three online tensors."""
    import torch
    if past_data is None or loss_scale == 0.0:
        zero = torch.zeros((), dtype=observations.dtype, device=observations.device)
        return zero, zero
    d = _as_rsr_data(past_data)
    cur = torch.cat([observations.reshape(-1, observations.shape[-1]), policy_actions.reshape(-1, policy_actions.shape[-1]),
                     next_observations.reshape(-1, next_observations.shape[-1])], dim=-1)
    if cur.shape[-1] != d.reference_data.shape[-1]:
        raise ValueError(f"online transition width does not match RSR reference data: {cur.shape[-1]} != {d.reference_data.shape[-1]}")
    augmented = torch.cat([d.reference_data, cur], dim=0)
    density = evaluate_kde(augmented, d.grid, d.bandwidth)
    distance = wasserstein_distance(density, d.reference_density)
    return loss_scale * d.divergence * distance, distance
