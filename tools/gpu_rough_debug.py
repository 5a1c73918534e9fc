"""Go2 rough terrain: find the env where HIP and oracle disagree most after one teacher-forced step."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import oracle as O
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs import go2
np.set_printoptions(precision=6, suppress=True, linewidth=220)
n = 512
jenv = go2.load("Go2JoystickRoughTerrain")
dr = go2.domain_randomize(jenv.sys, prng.split(prng.PRNGKey(12), n))
env = go2.wrap_for_brax_training(jenv, n, episode_length=1000, randomization_fn=lambda sys: dr)
odr = {{"actuator_gainprm": "gainprm", "actuator_biasprm": "biasprm"}.get(k, k): v for k, v in dr.items()}
orc = O.Oracle(env.blob); orc.set_ncon_cap(env.dims.ncon_max)
o64 = O.Oracle(env.blob, "f64"); o64.set_ncon_cap(env.dims.ncon_max)
keys = prng.split(prng.PRNGKey(11), n)
st = orc.new_state(n, odr); orc.reset(st, keys)
s = env.reset(keys); torch.cuda.synchronize()
fields = ["qpos", "qvel", "ctrl", "qacc_warmstart", "time", "xpos", "site_xpos", "obs", "reward", "done", "metrics", "info_go2",
          "info_steps", "info_truncation", "info_episode_done", "info_episode_metrics", "first_qpos", "first_qvel", "first_ctrl",
          "first_warmstart", "first_time", "first_xpos", "first_site_xpos", "first_obs"]
rng = np.random.default_rng(11)
a = np.clip(rng.normal(size=(n, 12)) * 0.5, -1, 1).astype(np.float32)
orc.step(st, a)                      # mirrors the test's rng stream: depth 0 consumes one action
for depth in (5,):
    for _ in range(depth):
        orc.step(st, np.clip(rng.normal(size=(n, 12)) * 0.5, -1, 1).astype(np.float32))
    for k in fields:
        env.view(k).copy_(torch.from_numpy(st[k].reshape(n, -1)))
    st0 = {k: (v.copy() if v is not None else None) for k, v in st.items()}
    st64 = {k: (v.copy() if v is not None else None) for k, v in st.items()}
    a = np.clip(rng.normal(size=(n, 12)) * 0.5, -1, 1).astype(np.float32)
    orc.step(st, a); o64.step(st64, a); env.step(s, a); torch.cuda.synchronize()
    g = lambda k: env.view(k).cpu().numpy().reshape(st[k].shape)
    e = np.abs(g("qvel") - st["qvel"]).max(1)
    w = int(np.argmax(e))
    print("worst env", w, "qvel err", e[w], "f32-f64 err", np.abs(st["qvel"][w] - st64["qvel"][w]).max())
    print("stats gpu", env.view("stats")[w].cpu().numpy(), "cpu", st["stats"][w], "f64", st64["stats"][w])
    print("qpos0", st0["qpos"][w])
    print("qvel gpu", g("qvel")[w]); print("qvel cpu", st["qvel"][w]); print("qvel f64", st64["qvel"][w])
    fs = st0["site_xpos"].reshape(n, -1, 3)[w]
    print("site_xpos before", fs)
    print("sorted errs", np.sort(e)[-8:])
