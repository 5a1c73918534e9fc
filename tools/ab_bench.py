"""A/B of builds on one box, interleaved: python tools/ab_bench.py [--workload W] [--envs N] [--rounds R] [--steps K] [--warmup W] lib1.so lib2.so@VAR=val,VAR2=val ...
(files under rsr_mjx_amd/csrc/; `@VAR=val` sets environment variables for that arm, e.g. RSR_GRID_PER_CU=8 or RSR_UNITS=2)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
wl, rounds, steps, warmup, envs_n = "cube", 2, "150", "20", "8192"
while args and args[0].startswith("--"):
    if args[0] == "--workload": wl = args[1]
    elif args[0] == "--rounds": rounds = int(args[1])
    elif args[0] == "--steps": steps = args[1]
    elif args[0] == "--warmup": warmup = args[1]
    elif args[0] == "--envs": envs_n = args[1]
    args = args[2:]
res = {l: [] for l in args}
info = {}
for rnd in range(rounds):
    for l in args:
        name, _, envs = l.partition("@")
        env = dict(os.environ, RSR_MJX_LIB=os.path.join(ROOT, "rsr_mjx_amd", "csrc", name))
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("="); env[k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--envs", envs_n, "--steps", steps, "--warmup", warmup, "--no-cpu-baseline", "--sub-batches", "0"],
                             env=env, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            res[l].append(j["value"]); info[l] = (j["config"]["lds_bytes_per_env"], j["config"]["ncon_max"], j["roofline"]["avg_launch_ms"])
        except Exception:
            print(l, "FAILED", out.stderr[-400:])
    print(f"round {rnd} done", flush=True)
for l in args:
    print(f"{l}: " + " ".join(f"{x / 1e6:.3f}" for x in res[l]) + f"  M env-steps/s   lds/ncon/kernel ms {info.get(l)}")
