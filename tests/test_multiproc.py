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


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` outside torch.distributed.run starts two ranks itself (one per GPU on a GPU box; here the
    --dry-run leg: gloo rendezvous, barrier + MAX-over-ranks timing, the metric all_gather) and prints rank 0's one JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "7", "--envs", "48"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["shards"] == [[0, 48], [48, 96]] and line["ranks_seen"] == [0, 1]
    assert line["ms_per_step"] * 7 >= 20.0 - 1e-6          # the slower rank (sleeps 20 ms) sets the time
    # --gpus N with a WORLD_SIZE that disagrees is refused rather than silently run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                         env=dict(env, WORLD_SIZE="1", RANK="0"), timeout=120)
    assert bad.returncode != 0 and "must agree" in (bad.stderr + bad.stdout)


def test_randomization_keys_sliced_or_replicated():
    """Default: one global key fan-out sliced per rank; replicated: every rank draws the same set (RSR/train.py:212-217)."""
    from rsr_mjx_amd import prng
    from rsr_mjx_amd.distributed import randomization_keys
    key = prng.PRNGKey(5)
    full = prng.split(key, 8)
    a, b = randomization_keys(key, 8, 0, 2), randomization_keys(key, 8, 1, 2)
    np.testing.assert_array_equal(np.concatenate([a, b]), full)
    ra, rb = randomization_keys(key, 8, 0, 2, replicated=True), randomization_keys(key, 8, 1, 2, replicated=True)
    np.testing.assert_array_equal(ra, rb)
    np.testing.assert_array_equal(ra, prng.split(key, 4))
