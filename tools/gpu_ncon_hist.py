"""Histogram of active contacts per env (last substep) over a rollout of the headline workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs.airbot import AirbotPlayBase, AirbotTShape, domain_randomize
for name, cls in (("cube", AirbotPlayBase), ("tshape", AirbotTShape)):
    n = 8192
    envdef = cls()
    dr = domain_randomize(envdef.sys, prng.split(prng.PRNGKey(1), n)) if name == "cube" else None
    env = envdef.batched(n, episode_length=1200, auto_reset=True, randomization=dr)
    s = env.reset(prng.split(prng.PRNGKey(0), n))
    hist = torch.zeros(64, dtype=torch.long, device="cuda")
    drop = 0
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for t in range(600):
        env.step(s, torch.clamp(torch.randn(n, 5, device="cuda", generator=g), -1, 1))
        st = env.view("stats")
        hist += torch.bincount(st[:, 2].long().clamp(0, 63), minlength=64)
        drop += int(st[:, 3].sum())
    h = hist.cpu().numpy()
    tot = h.sum()
    print(name, "dropped", drop, "max ncon", int(np.nonzero(h)[0].max()))
    print("  cumulative share with ncon <= k:", {k: round(float(h[:k + 1].sum() / tot), 6) for k in (8, 10, 12, 14, 16, 18, 20, 22, 24)})
