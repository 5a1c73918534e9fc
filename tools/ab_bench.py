"""A/B of builds on one box, interleaved: python tools/ab_bench.py [--workload W] lib1.so lib2.so ...  (files under rsr_mjx_amd/csrc/)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
wl = "cube"
if args and args[0] == "--workload":
    wl = args[1]; args = args[2:]
res = {l: [] for l in args}
info = {}
for rnd in range(2):
    for l in args:
        env = dict(os.environ, RSR_MJX_LIB=os.path.join(ROOT, "rsr_mjx_amd", "csrc", l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "150", "--warmup", "20", "--no-cpu-baseline", "--sub-batches", "0"],
                             env=env, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            res[l].append(j["value"]); info[l] = (j["config"]["lds_bytes_per_env"], j["config"]["ncon_max"], j["roofline"]["avg_launch_ms"])
        except Exception:
            print(l, "FAILED", out.stderr[-400:])
for l in args:
    print(f"{l}: " + " ".join(f"{x / 1e6:.3f}" for x in res[l]) + f"  M env-steps/s   lds/ncon/kernel ms {info.get(l)}")
