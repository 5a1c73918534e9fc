"""A/B throughput of several builds of librsrmjx on ONE box, interleaved (guide rule 24).
usage: python tools/ab_bench.py libA.so libB.so ...   (paths relative to rsr_mjx_amd/csrc)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, RSR_MJX_LIB=os.path.join(ROOT, "rsr_mjx_amd", "csrc", l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "150", "--warmup", "20", "--no-cpu-baseline"],
                             env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[l].append(d["value"])
        except Exception:
            print(l, "FAILED", out.stderr[-500:])
for l in libs:
    v = res[l]
    print(f"{l:32s} " + " ".join(f"{x/1e6:.3f}" for x in v) + f"   median {sorted(v)[len(v)//2]/1e6:.3f} M env-steps/s")
