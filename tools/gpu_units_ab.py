"""Work-queue dispatch: bit-identity of the results across units per env-step, and the step rate for each (cube, 8192 envs, DR on)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs.airbot import AirbotPlayBase, AirbotTShape, domain_randomize

n = 8192
tshape = "--tshape" in sys.argv
envdef = AirbotTShape() if tshape else AirbotPlayBase()
dr = None if tshape else domain_randomize(envdef.sys, prng.split(prng.PRNGKey(1), n))
keys = prng.split(prng.PRNGKey(0), n)
acts = torch.clamp(torch.randn((64, n, 5), device="cuda"), -1, 1)
recs = {}
for units in (1, 2, 4):
    env = envdef.batched(n, episode_length=1200, auto_reset=True, randomization=dr)
    env.set_schedule(units)
    s = env.reset(keys)
    for t in range(40):
        env.step(s, acts[t])
    torch.cuda.synchronize()
    recs[units] = env.record.clone()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(200):
            env.step(s, acts[t % 64])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"units {units}: {n * 200 / dt / 1e6:.3f} M env-steps/s ({dt / 200 * 1e3:.3f} ms/step)", flush=True)
    st = env.view("stats").cpu().numpy()
    print("   stats[3] min (timeouts would be -1):", st[:, 3].min())
for u in (2, 4):
    same = torch.equal(recs[1].view(torch.int32), recs[u].view(torch.int32))
    print(f"records after 40 steps, units {u} vs 1: {'bit-identical' if same else 'DIFFERENT'}")
    if not same:
        d = (recs[1] != recs[u]).nonzero()
        print("   first differences (env, float index):", d[:8].tolist(), "count", len(d))
