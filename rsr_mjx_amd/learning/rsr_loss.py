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
    import torch
    return (p * torch.log((p + 1e-10) / (q + 1e-10))).sum()


def wasserstein_distance(p, q):
    import torch
    return (torch.cumsum(p, 0) - torch.cumsum(q, 0)).abs().sum()


def _same_table(named) -> None:
    """All tensors are [rows, width] tables of one common shape (the three transition sets share rows and columns)."""
    shapes = {name: tuple(t.shape) for name, t in named.items()}
    first = next(iter(shapes.values()))
    if len(first) != 2 or any(sh != first for sh in shapes.values()):
        raise ValueError(f"RSR transition tables must be 2-D and of one shape; got {shapes}")


def build_rsr_data(real_data, previous_sim_data, current_sim_data, *, num_samples: int = 10, min_value: float = -3.0,
                   max_value: float = 3.0, bandwidth: float = 0.1, seed: int = 0) -> RSRData:
    """The fixed part of the loss: KL(real || previous sim) on the grid, and the current sim's density / data as the reference."""
    _same_table({"real_data": real_data, "previous_sim_data": previous_sim_data, "current_sim_data": current_sim_data})
    if not (num_samples > 0 and bandwidth > 0):
        raise ValueError(f"need a positive grid size and bandwidth; got num_samples={num_samples}, bandwidth={bandwidth}")
    grid = make_grid(num_samples, real_data.shape[-1], min_value, max_value, seed, device=real_data.device, dtype=real_data.dtype)
    density = {name: evaluate_kde(table, grid, bandwidth)
               for name, table in (("real", real_data), ("previous", previous_sim_data), ("current", current_sim_data))}
    return RSRData(divergence=kl_divergence(density["real"], density["previous"]), reference_density=density["current"],
                   reference_data=current_sim_data, grid=grid, bandwidth=bandwidth)


def _as_rsr_data(past_data: Any) -> RSRData:
    """RSRData, its five fields as a sequence, or the older three-field form (divergence, density, data) whose grid and bandwidth
    were implicit: the default grid of the density's length over [-3, 3] with seed 0, bandwidth 0.1 (rsr_loss.py:92-119)."""
    if isinstance(past_data, RSRData):
        return past_data
    fields = tuple(past_data) if isinstance(past_data, (tuple, list)) else None
    if fields is not None and len(fields) == len(RSRData._fields):
        return RSRData(*fields)
    if fields is not None and len(fields) == 3:
        divergence, density, data = fields
        grid = make_grid(int(density.shape[0]), int(data.shape[-1]), device=data.device, dtype=data.dtype)
        return RSRData(divergence, density, data, grid, 0.1)
    raise TypeError(f"past_data: expected RSRData, its {len(RSRData._fields)} fields, or (divergence, density, data); got {type(past_data).__name__}"
                    + (f" of length {len(fields)}" if fields is not None else ""))


def compute_rsr_loss(observations, policy_actions, next_observations, past_data: Any, *, loss_scale: float = 1.0) -> Tuple[Any, Any]:
    """(scaled_loss, distribution_distance); any number of leading dims on the three online tensors.  The online transitions
    (s, a, s') join the reference data, the joint density is re-estimated on the grid, and its Wasserstein distance to the
    reference density, weighted by the fixed real-vs-sim divergence, is the loss (rsr_loss.py:122-175)."""
    import torch
    if past_data is None or loss_scale == 0.0:
        zero = torch.zeros((), dtype=observations.dtype, device=observations.device)
        return zero, zero
    d = _as_rsr_data(past_data)
    online = torch.cat([t.reshape(-1, t.shape[-1]) for t in (observations, policy_actions, next_observations)], dim=-1)
    if online.shape[-1] != d.reference_data.shape[-1]:
        raise ValueError(f"online transitions are {online.shape[-1]} wide (obs + action + next obs), the RSR reference data {d.reference_data.shape[-1]}")
    density = evaluate_kde(torch.cat([d.reference_data, online], dim=0), d.grid, d.bandwidth)
    distance = wasserstein_distance(density, d.reference_density)
    return loss_scale * d.divergence * distance, distance
