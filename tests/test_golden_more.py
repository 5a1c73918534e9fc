"""Committed golden fixtures for the T-shape and Go2 configs (tests/golden/make_more_goldens.py): the oracle must keep
reproducing them (CPU), and the HIP stepper must land on them from the stored snapshots (GPU, teacher-forced)."""
import os

import numpy as np
import pytest

import parity_envelopes as PE
from conftest import make_blob
from rsr_mjx_amd import prng
from rsr_mjx_amd.model import model_fields, pack_blob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SNAP = (0, 1, 25, 60, 99)
PIPE = ["qpos", "qvel", "ctrl", "qacc_warmstart", "time", "xpos", "site_xpos"]
GNAME = {"priv_obs": "privileged_obs"}


def _cases():
    return [("tshape_n4", None), ("go2_flat_n4", "Go2JoystickFlatTerrain"), ("go2_rough_n4", "Go2JoystickRoughTerrain"),
            ("go2_handstand_n4", "Go2Handstand")]


_EPISODE = {"Go2Handstand": 500}


def _blob(name, task, tshape_model=None):
    if task is None:
        return make_blob(tshape_model, "tshape", episode_length=1200, auto_reset=True)
    from rsr_mjx_amd.envs import go2
    e = go2.load(task)
    f = model_fields(e.sys); f.update(e._fields_fn(e.sys, _EPISODE.get(task, 1000), True))
    return pack_blob(f)


@pytest.mark.parametrize("name,task", _cases())
def test_oracle_reproduces_golden(name, task, tshape_model, oracle_mod):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    orc = oracle_mod.Oracle(_blob(name, task, tshape_model))
    n = g["keys"].shape[0]
    st = orc.new_state(n)
    orc.reset(st, g["keys"])
    np.testing.assert_array_equal(st["obs"], g["reset_obs"])
    np.testing.assert_array_equal(st["qpos"], g["reset_qpos"])
    for t in range(g["actions"].shape[0]):
        orc.step(st, g["actions"][t])
        np.testing.assert_array_equal(st["obs"], g["obs"][t], err_msg=f"{name} step {t}")
        np.testing.assert_array_equal(st["reward"], g["reward"][t])
    assert np.isfinite(g["obs"]).all() and g["done"].sum() == (1 if name == "go2_handstand_n4" else 0)      # (the handstand run holds one termination + auto-reset)
    assert np.abs(np.diff(g["obs"], axis=0)).max() > 1e-3                 # the envs move


@pytest.mark.gpu
@pytest.mark.parametrize("name,task", _cases())
def test_hip_lands_on_golden(name, task):
    import torch
    g = np.load(os.path.join(GOLD, name + ".npz"))
    n = g["keys"].shape[0]
    if task is None:
        from rsr_mjx_amd.envs.airbot import AirbotTShape
        env = AirbotTShape().batched(n, episode_length=1200, auto_reset=True)
        info = ["info_target_base_pos", "info_target_vertical_pos", "info_target_w", "info_new_T_pos", "info_T_pos", "info_xita", "info_steps"]
    else:
        from rsr_mjx_amd.envs import go2
        env = go2.load(task).batched(n, episode_length=_EPISODE.get(task, 1000), auto_reset=True)
        info = ["info_go2", "info_steps"]
    env.reset(g["keys"])
    torch.cuda.synchronize()
    err = lambda a, b: (np.abs(a.astype(np.float64) - b).reshape(n, -1) / np.maximum(1.0, np.abs(b.astype(np.float64)).reshape(n, -1).max(1, keepdims=True))).max()
    get = lambda f, like: env.view(GNAME.get(f, f)).cpu().numpy().reshape(like.shape)
    assert err(get("obs", g["reset_obs"]), g["reset_obs"]) <= 1e-5
    assert err(get("qpos", g["reset_qpos"]), g["reset_qpos"]) <= 1e-6
    state = env._state
    for t in SNAP:
        for f in PIPE + ["obs", "reward", "done", "metrics"] + info:
            a = g[f"pre{t}_{f}"]
            env.view(GNAME.get(f, f)).copy_(torch.from_numpy(a.reshape(n, -1)))
        env.step(state, g["actions"][t])
        torch.cuda.synchronize()
        np.testing.assert_array_equal(get("done", g[f"post{t}_done"]), g[f"post{t}_done"])
        kind = {"tshape_n4": "tshape", "go2_flat_n4": "go2", "go2_rough_n4": "go2rough", "go2_handstand_n4": "handstand"}[name]
        for f in ("obs", "reward", "xpos", "qpos", "qvel"):
            want = g[f"post{t}_{f}"]
            tol = PE.bound(kind, "reset" if t == 0 else "rollout", f)       # per field: 3 x the measured maximum (tests/parity_envelopes.py)
            assert err(get(f, want), want) <= tol, (name, t, f, err(get(f, want), want), tol)
