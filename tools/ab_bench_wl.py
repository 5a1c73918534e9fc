"""usage: python tools/ab_bench_wl.py <workload> libA.so libB.so ...  (interleaved A/B on one box)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl, libs = sys.argv[1], sys.argv[2:]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, RSR_MJX_LIB=os.path.join(ROOT, "rsr_mjx_amd", "csrc", l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "200", "--warmup", "30", "--no-cpu-baseline", "--sub-batches", "0"],
                             env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1]); res[l].append(d["value"]); lds = d["config"]["lds_bytes_per_env"]
        except Exception:
            print(l, "FAILED", out.stderr[-400:]); lds = None
    
for l in libs:
    v = res[l]
    print(f"{wl} {l:24s} " + " ".join(f"{x/1e6:.3f}" for x in v))
