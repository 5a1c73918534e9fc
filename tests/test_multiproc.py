"""N>1 path on CPU: two gloo ranks shard the env batch by index, step their shards with the oracle standing
in for the device stepper, and all_gather the end-of-rollout metric vector; the union equals the unsharded run."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, make_blob


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, blob, total, steps, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from rsr_mjx_amd import prng
    from rsr_mjx_amd.distributed import gather_metrics, shard_keys, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
    keys = shard_keys(prng.PRNGKey(11), total, rank, world)
    orc = O.Oracle(blob)
    st = orc.new_state(hi - lo)
    orc.reset(st, keys, 1)
    acts = np.random.default_rng(0).uniform(-1, 1, (steps, total, 5)).astype(np.float32)
    for t in range(steps):
        orc.step(st, acts[t, lo:hi], 1)
    vec = torch.tensor([float(steps * (hi - lo)), float(st["reward"].sum()), float(st["done"].sum())], dtype=torch.float64)
    allm = gather_metrics(vec)
    np.save(os.path.join(out_dir, f"obs{rank}.npy"), st["obs"])
    if rank == 0:
        np.save(os.path.join(out_dir, "metrics.npy"), allm.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(cube_model, oracle_mod, tmp_path):
    import torch.multiprocessing as mp
    from rsr_mjx_amd import prng
    from rsr_mjx_amd.distributed import shard_range
    blob = make_blob(cube_model, episode_length=1200, auto_reset=True)
    total, steps, world = 16, 6, 2
    mp.spawn(_worker, args=(world, _free_port(), blob, total, steps, str(tmp_path)), nprocs=world, join=True)
    orc = oracle_mod.Oracle(blob)
    st = orc.new_state(total)
    orc.reset(st, prng.split(prng.PRNGKey(11), total), 1)
    acts = np.random.default_rng(0).uniform(-1, 1, (steps, total, 5)).astype(np.float32)
    for t in range(steps):
        orc.step(st, acts[t], 1)
    obs = np.concatenate([np.load(tmp_path / f"obs{r}.npy") for r in range(world)])
    np.testing.assert_array_equal(obs, st["obs"])          # env i is the same env for any GPU count
    met = np.load(tmp_path / "metrics.npy")
    assert met.shape == (world, 3) and met[:, 0].sum() == steps * total
    np.testing.assert_allclose(met[:, 1].sum(), st["reward"].sum(), rtol=1e-6)
    assert shard_range(total, 1, world) == (8, 16)
    with pytest.raises(ValueError):
        shard_range(15, 0, 2)
