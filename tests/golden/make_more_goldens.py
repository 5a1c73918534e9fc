"""Generates tests/golden/{tshape,go2_flat,go2_rough,go2_handstand}_n4.npz with the CPU oracle (fp32): small-N versions of
BASELINE.json configs[2], [3] and [4] (T-shape, Go2 flat, Go2 rough) and of the Go2 Handstand task.  Keys split(PRNGKey(0), 4);
actions from numpy default_rng(0) (T-shape U(-1,1); Go2 N(0, 0.3) clipped; Handstand N(0, 0.05): its targets integrate the
actions); raw wrappers as the bench uses them (episode 1200 / 1000 / 500, auto-reset).

Like cube_n4_200.npz these are regression fixtures of the oracle, NOT reference outputs (the JAX/MJX reference cannot run
in this pipeline; parity with MJX stays unpinned).  Run: python tests/golden/make_more_goldens.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from rsr_mjx_amd import prng  # noqa: E402
from rsr_mjx_amd.model import model_fields, pack_blob  # noqa: E402

SNAP = (0, 1, 25, 60, 99)
PIPE = ["qpos", "qvel", "ctrl", "qacc_warmstart", "time", "xpos", "site_xpos"]
OUT = ["obs", "reward", "done", "metrics"]


def cases():
    from rsr_mjx_amd.envs.airbot import AirbotTShape
    from rsr_mjx_amd.envs import go2
    t = AirbotTShape.__new__(AirbotTShape)
    from rsr_mjx_amd.mjcf import CompiledModel
    from conftest import make_blob
    tm = CompiledModel.load(os.path.join(ROOT, "rsr_mjx_amd", "assets", "airbot_tshape.npz"))
    yield "tshape_n4", make_blob(tm, "tshape", episode_length=1200, auto_reset=True), 5, 1.0, \
        ["info_target_base_pos", "info_target_vertical_pos", "info_target_w", "info_new_T_pos", "info_T_pos", "info_xita", "info_steps"]
    for name, task in (("go2_flat_n4", "Go2JoystickFlatTerrain"), ("go2_rough_n4", "Go2JoystickRoughTerrain")):
        e = go2.load(task)
        f = model_fields(e.sys); f.update(e._fields_fn(e.sys, 1000, True))
        yield name, pack_blob(f), 12, 0.3, ["info_go2", "info_steps", "priv_obs"]
    e = go2.load("Go2Handstand")
    f = model_fields(e.sys); f.update(e._fields_fn(e.sys, 500, True))
    yield "go2_handstand_n4", pack_blob(f), 12, 0.05, ["info_go2", "info_steps", "priv_obs"]


def main():
    n, steps = 4, 100
    only = sys.argv[1:]
    for name, blob, nu, std, info in cases():
        if only and name not in only:
            continue
        orc = O.Oracle(blob)
        keys = prng.split(prng.PRNGKey(0), n)
        rng = np.random.default_rng(0)
        acts = (rng.uniform(-1, 1, size=(steps, n, nu)) if std == 1.0 else np.clip(rng.normal(size=(steps, n, nu)) * std, -1, 1)).astype(np.float32)
        st = orc.new_state(n)
        orc.reset(st, keys)
        out = dict(keys=keys, actions=acts, **{f"reset_{f}": st[f].copy() for f in PIPE + ["obs"] + info})
        obs, rew, done = [], [], []
        for t in range(steps):
            if t in SNAP:
                for f in PIPE + OUT + info:
                    out[f"pre{t}_{f}"] = st[f].copy()
            orc.step(st, acts[t])
            if t in SNAP:
                for f in PIPE + OUT + info:
                    out[f"post{t}_{f}"] = st[f].copy()
            obs.append(st["obs"].copy()); rew.append(st["reward"].copy()); done.append(st["done"].copy())
        out.update(obs=np.stack(obs), reward=np.stack(rew), done=np.stack(done))
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **out)
        print("wrote", name, "done events", int(np.stack(done).sum()), "bytes", os.path.getsize(os.path.join(ROOT, "tests", "golden", name + ".npz")))


if __name__ == "__main__":
    main()
