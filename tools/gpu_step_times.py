"""Kernel time of every single rsr_step launch over the first steps after reset (HIP events around each launch) and the mean
solver statistics of that step: shows whether the driver's short run (5 warm-up + 20 timed steps) meets slower launches
than a 300-step run does, and why.  usage: python tools/gpu_step_times.py [--steps 120] [--workload cube]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs.airbot import AirbotPlayBase, AirbotTShape, domain_randomize

steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 120
wl = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "cube"
n = 8192
envdef = AirbotTShape() if wl == "tshape" else AirbotPlayBase()
dr = domain_randomize(envdef.sys, prng.split(prng.PRNGKey(1), n)) if wl == "cube" else None
env = envdef.batched(n, episode_length=1200, auto_reset=True, randomization=dr)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
acts = torch.clamp(torch.randn((steps, n, 5), generator=gen, device="cuda"), -1, 1)
for rep in range(2):
    st = env.reset(prng.split(prng.split(prng.PRNGKey(0), 3)[1], n))
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    stats = []
    ev[0].record()
    for t in range(steps):
        env.step(st, acts[t]); ev[t + 1].record()
        if rep == 1:
            stats.append(env.view("stats").float().mean(dim=0))
    torch.cuda.synchronize()
    ms = np.array([ev[t].elapsed_time(ev[t + 1]) for t in range(steps)])
    if rep == 1:
        S = torch.stack(stats).cpu().numpy()
        print(f"{wl}: per-launch ms (event to event), mean newton iters / ls iters / contacts of the LAST substep")
        for t in range(steps):
            if t < 30 or t % 10 == 0:
                print(f"step {t:4d}  {ms[t]:.4f} ms   niter {S[t,0]:.2f}  ls {S[t,1]:.2f}  ncon {S[t,2]:.2f}")
        print(f"mean of steps 5..24: {ms[5:25].mean():.4f} ms; of steps 50..{steps - 1}: {ms[50:].mean():.4f} ms")
