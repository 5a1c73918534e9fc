#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the fused Airbot-cube env step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, one per GPU, before any GPU call)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One process per GPU; every rank owns `--envs` (default 8192) environments, i.e. the env batch is sharded
by index with per-GPU work fixed ("weak" scaling).  A step = one `rsr_step` launch =
AutoReset/Episode wrappers + ctrl shaping + 4 physics substeps + reward/obs for every env of the rank.
Inputs (state records, per-env domain-randomised model leaves, K pre-generated action tensors) are resident
in HBM before the timed region.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_BYTES_S = 8.0e12          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8 TB/s (spec)
BYTES_PER_ENV_STEP_DR = 1220       # algorithmic bytes per env-step, Airbot cube with DR
BYTES_PER_ENV_STEP = 728           # ... without DR
VALU_PEAK_WAVE_INST_S = 256 * 4 * 2.4e9 / 2.0   # 1024 SIMDs, one wave64 VALU instruction per 2 cycles (MI355X_MICROARCH.md constants)


PROFILE_ROUND = "round3"


def csrc_sha16() -> str:
    """SHA-256 (first 16 hex digits) of the kernel sources the running library was built from: rsr_mjx_amd/csrc/*.hip, *.hpp and
    include/rsr_mjx.h, in name order.  tools/rocpd_summary.py stores the same hash in every PMC summary it writes."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "rsr_mjx_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp")))
    files.append(os.path.join(ROOT, "include", "rsr_mjx.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_profile(workload: str, n: int, dr_on: bool, sha: str):
    """HBM traffic and instruction counts per launch from the committed rocprofv3 PMC passes (profiles/round2_pmc_*_<workload>.csv,
    collected by tools/gpu_final_profile.sh on this very command line).  PMC cannot be read from inside the process, so the
    numbers are quoted only when they were measured on this configuration (8192 envs, default DR) AND on this build: every
    summary carries the hash of the kernel sources it was collected on; on a mismatch the fields are null."""
    if n != 8192 or (workload == "cube" and not dr_on):
        return None, "PMC summaries exist for the default configuration only"
    import csv
    vals = {}
    for tag in ("fetch", "write", "inst"):
        path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_{tag}_{workload}.csv")
        if not os.path.exists(path):
            return None, f"no committed PMC summary {os.path.basename(path)}"
        for row in csv.DictReader(open(path)):
            if "step_kernel" in row["kernel"]:
                if row.get("csrc_sha16") != sha:
                    return None, (f"{os.path.basename(path)} was collected on kernel sources {row.get('csrc_sha16')}, this build is {sha}: "
                                  "counters not quoted")
                vals[row["counter"]] = float(row["avg_per_dispatch"])
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None, "PMC summaries lack FETCH_SIZE / WRITE_SIZE"
    return vals, f"profiles/{PROFILE_ROUND}_pmc_{{fetch,write,inst}}_{workload}.csv, collected on kernel sources {sha}"


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 outside torch.distributed.run: start N ranks of this script, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (the reference finds its devices itself, RSR/train.py:170-180).  Runs before
    anything touches the GPU in this process; the children are fresh interpreters.  Rank 0's stdout is ours."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   RSR_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def cpu_baseline(blob: bytes, dr, seconds: float = 12.0, nu: int = 5, act_std: float = 1.0):
    """The CPU oracle (kind "port") timed on all host cores on a bounded sample of the same workload.  The sample is sized to
    the machine -- 64 envs per OpenMP thread, at least 1024 (the oracle hands out chunks of 8 envs dynamically: eight chunks per
    thread to balance over) -- and timed as three windows of seconds / 3 each; the median is reported and the three rates beside
    it (1024 envs on 128 threads, one 12 s window, moved by +-40 % between runs: one chunk per thread leaves the slowest chunk
    of every step in charge)."""
    from oracle import oracle as O
    from rsr_mjx_amd import prng
    O.build()
    orc = O.Oracle(blob)
    threads = orc.max_threads()
    n = max(1024, 64 * threads)
    if dr is not None:
        n = min(n, len(next(iter(dr.values()))))
    sub = None if dr is None else {k: np.ascontiguousarray(v[:n]) for k, v in dr.items()}
    st = orc.new_state(n, sub)
    orc.reset(st, prng.split(prng.PRNGKey(123), n), threads)
    rng = np.random.default_rng(1)
    acts = np.clip(rng.normal(size=(8, n, nu)) * act_std, -1, 1).astype(np.float32)
    for i in range(2):
        orc.step(st, acts[i], threads)
    rates, k, total = [], 0, 0.0
    for w in range(3):
        t0 = time.perf_counter()
        k0 = k
        while time.perf_counter() - t0 < seconds / 3.0:
            orc.step(st, acts[k % 8], threads)
            k += 1
        dt = time.perf_counter() - t0
        total += dt
        rates.append(n * (k - k0) / dt)
    med = float(np.median(rates))
    return dict(value=med, unit="env-steps/s", cores=threads, kind="port",
                windows=[float(r) for r in rates], spread=float((max(rates) - min(rates)) / med),
                sample=f"oracle/rsr_oracle.c (fp32, OpenMP, {threads} threads) on {n} envs ({n // threads} per thread) x {k} steps of the same "
                       f"workload, median of three {seconds / 3.0:.1f}-s windows ({total:.1f} s)")


def sub_batched_rate(envdef, keys, dr, n, ep_len, actions, parts, steps, warmup):
    """The same n envs stepped as `parts` independent sub-batches, each on its own HIP stream, with no lock-step between
    them: while one sub-batch's launch drains (its last waves run alone for about half a wave lifetime, ~0.37 ms of the
    1.84 ms lock-step launch at 8192 envs) the other's next launch fills the idle SIMD slots.  Envs are independent, so the
    results are the same; what changes is that a consumer must also work per sub-batch (double-buffered rollouts).
    Reported beside `value`, never as `value`."""
    import torch
    m = n // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    envs, states = [], []
    for k in range(parts):
        sub = None if dr is None else {f: v[k * m:(k + 1) * m] for f, v in dr.items()}
        e = envdef.batched(m, episode_length=ep_len, auto_reset=True, randomization=sub)
        envs.append(e)
        states.append(e.reset(keys[k * m:(k + 1) * m]))
    npool = actions.shape[0]

    def run(count):
        for i in range(count):
            for k in range(parts):
                with torch.cuda.stream(streams[k]):
                    envs[k].step(states[k], actions[i % npool, k * m:(k + 1) * m])
    torch.cuda.synchronize()
    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"sub_batches": parts, "value": n * steps / dt, "unit": "env-steps/s", "ms_per_step": dt / steps * 1e3,
            "note": f"{parts} x {m} envs on {parts} HIP streams, {steps} steps each, no lock-step between sub-batches; this GPU only"}


def dry_run(args, rank: int, world: int) -> None:
    """The N > 1 plumbing without a GPU: gloo rendezvous, barrier, MAX-over-ranks time, the metric all_gather, one JSON line."""
    import torch
    import torch.distributed as dist
    from rsr_mjx_amd.distributed import gather_metrics, shard_range
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    n = args.envs
    lo, hi = shard_range(n * world, rank, world)
    metrics = torch.tensor([float(args.steps * n), float(lo), float(hi), float(rank)])
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))                  # ranks finish at different times: the reported time is the slowest rank's
    allm = gather_metrics(metrics)
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "dry-run (no GPU work)", "value": float(allm[:, 0].sum()) / float(t.item()), "unit": "env-steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t.item()) / args.steps * 1e3,
                          "scaling": "weak", "shards": [[int(a), int(b)] for a, b in allm[:, 1:3].tolist()],
                          "ranks_seen": [int(r) for r in allm[:, 3].tolist()]}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs", type=int, default=8192, help="environments per GPU")
    ap.add_argument("--no-dr", action="store_true", help="disable domain randomisation")
    ap.add_argument("--workload", default="cube", choices=["cube", "tshape", "go2", "go2rough", "handstand"],
                    help="cube = BASELINE headline (configs[1] family); tshape / go2 / go2rough = configs[2] / [3] / [4] families")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="length of the bounded CPU-oracle sample")
    ap.add_argument("--sub-batches", type=int, default=2,
                    help="also report the rate with the same envs stepped as this many independent sub-batches on their own HIP "
                         "streams (0 = skip); `value` is always the lock-step figure")
    ap.add_argument("--replicated-dr", action="store_true",
                    help="every GPU draws the same domain-randomisation key set, as the reference does (RSR/train.py:212-217); "
                         "default: one global key fan-out sliced per rank, so env i is the same env for any GPU count")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: ranks rendezvous over gloo and run the launcher, the barrier / MAX-over-ranks timing and the "
                         "end-of-rollout gather on synthetic per-rank metrics (CPU test of the N > 1 plumbing)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist
    from rsr_mjx_amd import prng
    from rsr_mjx_amd.distributed import gather_metrics, randomization_keys, shard_keys, shard_range

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the two must agree")
    if args.dry_run:
        return dry_run(args, rank, world)
    from rsr_mjx_amd.envs.airbot import AirbotPlayBase, domain_randomize
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    n = args.envs
    total = n * world
    # global key fan-out, sliced per rank so results do not depend on the GPU count (RSR/train.py:232-235)
    key_env = prng.split(prng.PRNGKey(0), 3)[1]
    lo, hi = shard_range(total, rank, world)
    keys = shard_keys(key_env, total, rank, world)
    if args.workload in ("go2", "go2rough", "handstand"):
        from rsr_mjx_amd.envs import go2
        wl_name = {"go2": "Go2JoystickFlatTerrain", "go2rough": "Go2JoystickRoughTerrain", "handstand": "Go2Handstand"}[args.workload]
        envdef = go2.load(wl_name, device=f"cuda:{local_rank}")
        dr, ep_len, act_std = None, (500 if args.workload == "handstand" else 1000), 0.3
        args.no_dr = True
    elif args.workload == "tshape":
        from rsr_mjx_amd.envs.airbot import AirbotTShape
        envdef = AirbotTShape(device=f"cuda:{local_rank}")
        dr, ep_len, act_std, wl_name = None, 1200, 1.0, "AirbotPlayBase T_shape_env"
        args.no_dr = True
    else:
        envdef = AirbotPlayBase(device=f"cuda:{local_rank}")
        dr = None if args.no_dr else domain_randomize(envdef.sys, randomization_keys(prng.PRNGKey(1), total, rank, world, args.replicated_dr))
        ep_len, act_std, wl_name = 1200, 1.0, "AirbotPlayBase cube_env"
    env = envdef.batched(n, episode_length=ep_len, auto_reset=True, randomization=dr)   # train.py:47-50
    state = env.reset(keys)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    npool = min(args.steps, 256)
    nu = env.action_size
    actions = torch.clamp(torch.randn((npool, n, nu), generator=gen, device=dev) * act_std, -1.0, 1.0)
    metrics = torch.zeros(4, device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def rollout_metrics():
        # end-of-rollout metric gather (the path's only collective): one reduction launch over the batch's records
        # (rsr_rollout_metrics: envs, sum of reward, sum of done, mean episode reward so far), then the all_gather
        return gather_metrics(env.rollout_metrics(metrics))

    for i in range(args.warmup):
        env.step(state, actions[i % npool])
    env.timing_begin(); env.timing_end()        # warm-up covers everything the timed region runs once: the event pair,
    rollout_metrics()                           # the reductions' first-use code-object loads, the collective's set-up
    barrier()
    env.timing_begin()
    t0 = time.perf_counter()
    for i in range(args.steps):
        env.step(state, actions[i % npool])
    kernel_ms, launches = env.timing_end()      # HIP events on the launch stream; also synchronises it
    metrics_all = rollout_metrics()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    env_steps = float(metrics_all[:, 0].sum().item()) * args.steps
    timeouts = env.handoff_timeouts()         # outside the timed region: the work queue's sticky error word (0 on a healthy run)

    if rank == 0:
        bytes_per = {"cube": BYTES_PER_ENV_STEP if args.no_dr else BYTES_PER_ENV_STEP_DR, "tshape": 560, "go2": 2536, "go2rough": 2536, "handstand": 1580}[args.workload]
        avg_launch_s = kernel_ms * 1e-3 / max(launches, 1)
        achieved = bytes_per * n / avg_launch_s
        stats = env.view("stats").float().mean(dim=0).tolist()
        sha = csrc_sha16()
        pmc, pmc_note = pmc_profile(args.workload, n, not args.no_dr, sha)
        # rocprofv3 reports KiB; on gfx950 FETCH_SIZE tallies 128-B fabric read requests at 64 B (MI355X_MICROARCH.md, HBM): doubled
        traffic = None if pmc is None else (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        out = {
            "metric": "env-steps/sec at num_envs=8192, Airbot cube" if args.workload == "cube" else f"env-steps/sec, {wl_name}",
            "value": env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{wl_name}, num_envs={n} per GPU ({total} total), episode_length={ep_len}, "
                            f"auto-reset, domain randomisation {'off' if args.no_dr else 'on'}, {env.dims.n_frames} substeps/env-step, "
                            f"actions N(0,{act_std}) clipped to +-1",
                "num_envs_per_gpu": n, "parallelism": f"env-batch sharded by index over {world} GPU(s), no data-path collective",
                "csrc_sha16": sha, "dr_keys": "replicated per GPU (RSR/train.py:212-217)" if args.replicated_dr else "global fan-out sliced per rank",
                "kernel": {"cube": "rsr::step_kernel<CubeDims, ENV_CUBE>", "tshape": "rsr::step_kernel<TShapeDims, ENV_TSHAPE>",
                           "go2": "rsr::go2_step_kernel<Go2FlatDims>", "go2rough": "rsr::go2_step_kernel<Go2Dims>",
                           "handstand": "rsr::hs_step_kernel<HandDims>"}[args.workload]
                          + " (one wavefront per env" + ("" if args.workload.startswith("go2") or args.workload == "handstand" else "; persistent waves draw (env, substep) work units from a ticket queue") + ")",
                "lds_bytes_per_env": int(env.dims.lds_bytes), "ncon_max": int(env.dims.ncon_max),
                "mean_newton_iters_last_substep": stats[0], "mean_linesearch_iters_last_substep": stats[1],
                "mean_active_contacts": stats[2], "dropped_contacts_mean": stats[3], "handoff_timeouts": timeouts,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BYTES_S / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_BYTES_S, "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_per * n, "avg_launch_ms": avg_launch_s * 1e3,
                "traffic_detail": None if pmc is None else {
                    "fetch_bytes": 2.0 * pmc["FETCH_SIZE"] * 1024.0, "fetch_size_counter_bytes": pmc["FETCH_SIZE"] * 1024.0,
                    "write_bytes": pmc["WRITE_SIZE"] * 1024.0,
                    "source": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per launch; FETCH_SIZE doubled as the guide "
                              "prescribes for gfx950 (exact for 16-B-per-lane reads; dword-per-lane accesses are uncalibrated: "
                              "MI355X_MICROARCH.md, HBM section)"},
                "pmc_source": pmc_note,
                "valu": None if pmc is None or "SQ_INSTS_VALU" not in pmc else {
                    "wave_instructions_per_env_step": pmc["SQ_INSTS_VALU"] / n,
                    "achieved_wave_inst_per_s": pmc["SQ_INSTS_VALU"] / avg_launch_s,
                    "peak_wave_inst_per_s": VALU_PEAK_WAVE_INST_S,
                    "frac": pmc["SQ_INSTS_VALU"] / avg_launch_s / VALU_PEAK_WAVE_INST_S},
                "note": "state stays on-chip for the whole step, so the path is VALU-issue / LDS-latency bound, not HBM bound "
                        "(SURVEY.md 8d); the HBM fraction is reported because the metric names it, the VALU fraction beside it",
            },
        }
        if args.sub_batches > 1 and world == 1 and n % args.sub_batches == 0:      # single-GPU extra; multi-GPU runs stay lean
            out["sub_batched"] = sub_batched_rate(envdef, keys, dr, n, ep_len, actions, args.sub_batches, args.steps, args.warmup)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(env.blob, dr, seconds=args.cpu_seconds, nu=nu, act_std=act_std)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
