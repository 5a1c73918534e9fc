"""A/B of two builds over all four workloads on one box (interleaved)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
for wl in ("cube", "tshape", "go2", "go2rough"):
    res = {l: [] for l in libs}
    for rnd in range(2):
        for l in libs:
            env = dict(os.environ, RSR_MJX_LIB=os.path.join(ROOT, "rsr_mjx_amd", "csrc", l))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "150", "--warmup", "20", "--no-cpu-baseline"],
                                 env=env, capture_output=True, text=True)
            try:
                res[l].append(json.loads(out.stdout.strip().splitlines()[-1])["value"])
            except Exception:
                print(l, wl, "FAILED", out.stderr[-300:])
    print(wl, "  ".join(f"{l}: " + " ".join(f"{x/1e6:.3f}" for x in res[l]) for l in libs))
