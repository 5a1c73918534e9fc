"""Multi-GPU plumbing for the env batch: one process per GPU, envs sharded contiguously by index
(reference RSR/train.py:232-235 reshapes key_envs to (devices, num_envs/devices, 2)); no data-path
collective; one all_gather of a small metric vector at the end of a rollout (RCCL over xGMI on GPUs,
gloo in the CPU tests)."""
from __future__ import annotations

from typing import Tuple

import numpy as np

from . import prng


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    if total % world:
        raise ValueError(f"num_envs={total} is not divisible by world_size={world} (RSR/train.py:208)")
    per = total // world
    return rank * per, (rank + 1) * per


def shard_keys(key_env: np.ndarray, total: int, rank: int, world: int) -> np.ndarray:
    """Keys of this rank's envs: split(key_env, total) sliced, so env i gets the same key for any GPU count."""
    lo, hi = shard_range(total, rank, world)
    return prng.split(key_env, total)[lo:hi]


def randomization_keys(key: np.ndarray, total: int, rank: int, world: int, replicated: bool = False) -> np.ndarray:
    """Keys of this rank's domain-randomisation draws.  Default: split(key, total) sliced like the env keys (env i is the same
    env for any GPU count).  replicated: every rank draws split(key, total / world) -- the reference hands every device the
    same randomisation rng (RSR/train.py:212-217: `randomization_rng = split(key_env, num_envs // local_device_count)`)."""
    lo, hi = shard_range(total, rank, world)
    if replicated:
        return prng.split(key, hi - lo)
    return prng.split(key, total)[lo:hi]


def gather_metrics(vec):
    """all_gather of a 1-D float tensor -> [world, k] (identity for a single process)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return vec[None]
    out = [torch.zeros_like(vec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, vec)
    return torch.stack(out)
