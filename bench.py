#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the fused Airbot-cube env step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
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


def pmc_profile(workload: str, n: int, dr_on: bool):
    """HBM traffic and instruction counts per launch from the committed rocprofv3 PMC passes (profiles/round1_final_pmc_*.csv,
    collected by tools/gpu_final_profile.sh on this very command line).  PMC cannot be read from inside the process, so the
    numbers apply only to the configuration they were measured on (headline cube, 8192 envs, DR on); otherwise None."""
    if workload != "cube" or n != 8192 or not dr_on:
        return None
    import csv
    vals = {}
    for tag in ("fetch", "write", "inst"):
        path = os.path.join(ROOT, "profiles", f"round1_final_pmc_{tag}.csv")
        if not os.path.exists(path):
            return None
        for row in csv.DictReader(open(path)):
            if "step_kernel" in row["kernel"]:
                vals[row["counter"]] = float(row["avg_per_dispatch"])
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return vals


def cpu_baseline(blob: bytes, dr, seconds: float = 12.0, nu: int = 5, act_std: float = 1.0):
    """The CPU oracle (kind "port") timed on all host cores on a bounded sample of the same workload."""
    from oracle import oracle as O
    from rsr_mjx_amd import prng
    O.build()
    orc = O.Oracle(blob)
    threads = orc.max_threads()
    n = 1024
    sub = None if dr is None else {k: np.ascontiguousarray(v[:n]) for k, v in dr.items()}
    st = orc.new_state(n, sub)
    orc.reset(st, prng.split(prng.PRNGKey(123), n), threads)
    rng = np.random.default_rng(1)
    acts = np.clip(rng.normal(size=(8, n, nu)) * act_std, -1, 1).astype(np.float32)
    for i in range(3):
        orc.step(st, acts[i], threads)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        orc.step(st, acts[k % 8], threads)
        k += 1
    dt = time.perf_counter() - t0
    return dict(value=n * k / dt, unit="env-steps/s", cores=threads, kind="port",
                sample=f"oracle/rsr_oracle.c (fp32, OpenMP) on {n} envs x {k} steps of the same workload, {dt:.1f} s")


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs", type=int, default=8192, help="environments per GPU")
    ap.add_argument("--no-dr", action="store_true", help="disable domain randomisation")
    ap.add_argument("--workload", default="cube", choices=["cube", "tshape", "go2", "go2rough"],
                    help="cube = BASELINE headline (configs[1] family); tshape / go2 / go2rough = configs[2] / [3] / [4] families")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="length of the bounded CPU-oracle sample")
    ap.add_argument("--sub-batches", type=int, default=2,
                    help="also report the rate with the same envs stepped as this many independent sub-batches on their own HIP "
                         "streams (0 = skip); `value` is always the lock-step figure")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rsr_mjx_amd import prng
    from rsr_mjx_amd.distributed import gather_metrics, shard_keys, shard_range
    from rsr_mjx_amd.envs.airbot import AirbotPlayBase, domain_randomize

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
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
    if args.workload in ("go2", "go2rough"):
        from rsr_mjx_amd.envs import go2
        wl_name = "Go2JoystickFlatTerrain" if args.workload == "go2" else "Go2JoystickRoughTerrain"
        envdef = go2.load(wl_name, device=f"cuda:{local_rank}")
        dr, ep_len, act_std = None, 1000, 0.3
        args.no_dr = True
    elif args.workload == "tshape":
        from rsr_mjx_amd.envs.airbot import AirbotTShape
        envdef = AirbotTShape(device=f"cuda:{local_rank}")
        dr, ep_len, act_std, wl_name = None, 1200, 1.0, "AirbotPlayBase T_shape_env"
        args.no_dr = True
    else:
        envdef = AirbotPlayBase(device=f"cuda:{local_rank}")
        dr = None if args.no_dr else domain_randomize(envdef.sys, prng.split(prng.PRNGKey(1), total)[lo:hi])
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
        # end-of-rollout metric gather (the path's only collective)
        metrics[0] = float(args.steps * n)
        metrics[1] = state.reward.sum()
        metrics[2] = state.done.sum()
        metrics[3] = state.info["episode_metrics"]["sum_reward"].mean()
        return gather_metrics(metrics)

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
    env_steps = float(metrics_all[:, 0].sum().item())

    if rank == 0:
        bytes_per = {"cube": BYTES_PER_ENV_STEP if args.no_dr else BYTES_PER_ENV_STEP_DR, "tshape": 560, "go2": 2536, "go2rough": 2536}[args.workload]
        avg_launch_s = kernel_ms * 1e-3 / max(launches, 1)
        achieved = bytes_per * n / avg_launch_s
        stats = env.view("stats").float().mean(dim=0).tolist()
        pmc = pmc_profile(args.workload, n, not args.no_dr)
        traffic = None if pmc is None else (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0      # rocprofv3 reports KiB
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
                "kernel": {"cube": "rsr::step_kernel<CubeDims, ENV_CUBE>", "tshape": "rsr::step_kernel<TShapeDims, ENV_TSHAPE>",
                           "go2": "rsr::go2_step_kernel<Go2Dims>", "go2rough": "rsr::go2_step_kernel<Go2Dims>"}[args.workload] + " (one wavefront per env)",
                "lds_bytes_per_env": int(env.dims.lds_bytes), "ncon_max": int(env.dims.ncon_max),
                "mean_newton_iters_last_substep": stats[0], "mean_linesearch_iters_last_substep": stats[1],
                "mean_active_contacts": stats[2], "dropped_contacts_mean": stats[3],
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BYTES_S / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_BYTES_S, "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_per * n, "avg_launch_ms": avg_launch_s * 1e3,
                "traffic_detail": None if pmc is None else {
                    "fetch_bytes": pmc["FETCH_SIZE"] * 1024.0, "write_bytes": pmc["WRITE_SIZE"] * 1024.0,
                    "source": "profiles/round1_final_pmc_fetch.csv, _write.csv (separate rocprofv3 --pmc passes, per launch; "
                              "dword-per-lane accesses are uncalibrated on gfx950, MI355X_MICROARCH.md HBM section); writes above "
                              "the record size are register spills to scratch memory (DESIGN.md 4)"},
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
