"""Hunt for a simulation blow-up (non-finite state): correlated large actions on many envs; on the first hit, save the env's
record BEFORE the bad step, its action and DR leaves, for replay on the CPU oracle (tools/replay_blowup.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs.airbot import AirbotPlayBase, wrap, domain_randomize
n, steps = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 4000
dr = domain_randomize(AirbotPlayBase().sys, prng.split(prng.PRNGKey(1), n))
env = wrap(AirbotPlayBase(), n, episode_length=1200, randomization_fn=lambda s: dr)
st = env.reset(prng.split(prng.PRNGKey(0), n))
g = torch.Generator(device="cuda"); g.manual_seed(0)
a = torch.zeros(n, 5, device="cuda")
hits = 0
prev = None
act = None
for t in range(steps):
    a = 0.9 * a + 0.6 * torch.randn(n, 5, device="cuda", generator=g)         # OU-like: smooth, mostly saturated after clipping
    last_act = act if t > 0 else a.clamp(-1, 1)
    act = a.clamp(-1, 1)
    prev2 = prev if t > 0 else env.record.clone()
    prev = env.record.clone()
    st = env.step(st, act)
    vmax = env.view("qvel")[:, :8].abs().max(dim=1).values
    vprev = prev[:, 22:30].abs().max(dim=1).values
    bad = (vmax > 300.0) & (vprev < 100.0)                                     # onset of the instability, not its end
    if bool(bad.any()):
        idx = int(torch.nonzero(bad)[0])
        hits += 1
        out = os.path.join(ROOT, "gpurun_out", f"blowup_{hits}.npz")
        np.savez(out, record=prev[idx].cpu().numpy(), record2=prev2[idx].cpu().numpy(), prev_action=last_act[idx].cpu().numpy(), after=env.record[idx].cpu().numpy(), action=act[idx].cpu().numpy(), step=t, env=idx,
                 **{k: v[idx] for k, v in dr.items()})
        print("blow-up at step", t, "env", idx, "saved", out, "stats", env.view("stats")[idx].tolist(), flush=True)
        if hits >= 3:
            break
print("done", steps, "steps x", n, "envs; hits", hits)
