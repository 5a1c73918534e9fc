"""Go2: oracle vs HIP, reset + a few steps (debugging aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import oracle as O
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs import go2
np.set_printoptions(precision=5, suppress=True, linewidth=220)
n = 128
env = go2.load("Go2JoystickFlatTerrain").batched(n, episode_length=1000, auto_reset=True)
orc = O.Oracle(env.blob); orc.set_ncon_cap(env.dims.ncon_max)
keys = prng.split(prng.PRNGKey(0), n)
st = orc.new_state(n); orc.reset(st, keys)
s = env.reset(keys); torch.cuda.synchronize()
F = ["qpos", "qvel", "ctrl", "qacc_warmstart", "xpos", "site_xpos", "obs", "reward", "done", "metrics", "info_go2", "info_steps", "first_obs", "first_qpos"]
def cmp(tag):
    print("==", tag)
    for k in F:
        a = env.view(k).cpu().numpy().reshape(st[k].shape).astype(np.float64); b = st[k].astype(np.float64)
        if k == "info_go2": a, b = a[:, :137], b[:, :137]
        print(f"  {k:16s} max_abs {np.abs(a-b).max():.3e}  scaled {(np.abs(a-b).reshape(n,-1)/np.maximum(1,np.abs(b).reshape(n,-1).max(1,keepdims=True))).max():.3e}")
    rg = env.view("info_go2")[:, 137:139].contiguous().view(torch.int32).cpu().numpy().view(np.uint32)
    ro = st["info_go2"][:, 137:139].copy().view(np.uint32)
    print("  rng equal:", bool((rg == ro).all()))
cmp("reset")
rng = np.random.default_rng(0)
for t in range(6):
    for k in F + ["time", "info_truncation", "info_episode_done", "info_episode_metrics"]:
        env.view(k).copy_(torch.from_numpy(st[k].reshape(n, -1)))
    a = np.clip(rng.normal(size=(n, 12)) * 0.5, -1, 1).astype(np.float32)
    orc.step(st, a); env.step(s, a); torch.cuda.synchronize()
    cmp(f"step {t}")
print("stats gpu", env.view("stats")[:4].cpu().numpy().tolist(), "cpu", st["stats"][:4].tolist())
