"""jax.random-compatible threefry2x32 PRNG on the host (numpy), used for key fan-out and domain
randomisation exactly where the reference calls jax.random (RSR/train.py:198-235,
ppo_train/airbot_training/domain_randomize.py:36-61).  Semantics: SURVEY.md Appendix D
(jax 0.4.29 defaults: threefry2x32, non-partitionable)."""
from __future__ import annotations

import numpy as np

_R = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return (x << np.uint32(r)) | (x >> np.uint32(32 - r))


def threefry2x32(key, c0, c1):
    """key: uint32[..., 2] broadcastable against counters c0, c1 (uint32 arrays). Returns (o0, o1)."""
    key = np.asarray(key, dtype=np.uint32)
    k0, k1 = key[..., 0], key[..., 1]
    ks = (k0, k1, k0 ^ k1 ^ np.uint32(0x1BD11BDA))
    with np.errstate(over="ignore"):
        x0 = np.asarray(c0, dtype=np.uint32) + ks[0]
        x1 = np.asarray(c1, dtype=np.uint32) + ks[1]
        for g in range(5):
            for r in _R[g & 1]:
                x0 = x0 + x1
                x1 = _rotl(x1, r)
                x1 = x1 ^ x0
            x0 = x0 + ks[(g + 1) % 3]
            x1 = x1 + ks[(g + 2) % 3] + np.uint32(g + 1)
    return x0, x1


def PRNGKey(seed: int) -> np.ndarray:
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


def random_bits(key, n: int) -> np.ndarray:
    """bits[..., n] = threefry_2x32(key, iota(n)) with jax's split-the-counters-in-halves layout."""
    key = np.asarray(key, dtype=np.uint32)
    half = (n + 1) // 2
    c0 = np.arange(half, dtype=np.uint32)
    c1 = np.arange(half, 2 * half, dtype=np.uint32)
    c1 = np.where(c1 < n, c1, 0).astype(np.uint32)
    o0, o1 = threefry2x32(key[..., None, :], c0, c1)
    return np.concatenate([o0, o1], axis=-1)[..., :n]


def split(key, num: int = 2) -> np.ndarray:
    """jax.random.split: [..., 2] -> [..., num, 2]."""
    bits = random_bits(key, 2 * num)
    return bits.reshape(bits.shape[:-1] + (num, 2))


def uniform(key, shape=(), minval=0.0, maxval=1.0) -> np.ndarray:
    """jax.random.uniform(key, shape, float32, minval, maxval) for a single key or a batch of keys [..., 2]."""
    key = np.asarray(key, dtype=np.uint32)
    n = int(np.prod(shape)) if shape else 1
    bits = random_bits(key, n)
    f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo = np.broadcast_to(np.asarray(minval, dtype=np.float32), shape).reshape(-1) if shape else np.float32(minval)
    hi = np.broadcast_to(np.asarray(maxval, dtype=np.float32), shape).reshape(-1) if shape else np.float32(maxval)
    v = (f * (hi - lo)).astype(np.float32) + lo
    v = np.maximum(lo, v).astype(np.float32)
    return v.reshape(key.shape[:-1] + tuple(shape))
