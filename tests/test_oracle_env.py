"""Env algebra and wrapper semantics of the oracle against the reference source text
(ppo_train/airbot_training/cube_env.py) restated independently in numpy, hand-computed cases,
and the committed regression fixture (BASELINE.json configs[0])."""
import os

import numpy as np
import pytest

from conftest import ROOT, make_blob
from rsr_mjx_amd import prng

GOLD = os.path.join(ROOT, "tests", "golden", "cube_n4_200.npz")


def _np_reward_obs(pre, post, W=(6.0, 3.0, 1.0, 0.778)):
    """cube_env.py:164-229 in numpy float32, written from the source text (not from the oracle)."""
    f = np.float32
    tp, cp, sp = pre["info_target_pos"], post["xpos"][:, 13], post["site_xpos"][:, 0]
    btd = np.sqrt(((tp - cp) ** 2).sum(-1, dtype=f)).astype(f)
    btd = np.where(btd < f(0.005), f(0), btd)
    push = (f(1) / (f(1) + f(3) * btd)) * f(W[0])
    site_z = np.where(sp[:, 2] < f(0.82), f(1), f(0))
    dx, dy = tp[:, 0] - cp[:, 0], tp[:, 1] - cp[:, 1]
    ang = np.arctan2(dy, dx + f(0.00001)).astype(f)
    dist = np.sqrt(dx * dx + dy * dy).astype(f) + f(0.04)
    ncp = np.stack([dx - dist * np.cos(ang).astype(f) + cp[:, 0], dy - dist * np.sin(ang).astype(f) + cp[:, 1]], -1)
    s2c = np.sqrt(((sp[:, :2] - pre["info_new_cube_pos"]) ** 2).sum(-1, dtype=f)).astype(f)
    s2c = np.where(s2c < f(0.042), f(0), s2c - f(0.042))
    siet = (f(1) - np.tanh(f(5) * s2c).astype(f)) * f(W[1])
    siet = np.where(btd < f(0.005), f(W[1]), siet)
    health = f(W[2]) * np.abs(np.where(sp[:, 2] < f(W[3]), f(1), f(0)) - f(1))
    reward = np.clip(push + siet + health + site_z, -100, 100)
    done = np.where(cp[:, 2] < f(0.6), f(1), f(0))
    obs = np.concatenate([post["qpos"][:, :6], sp, tp, cp, ncp, tp - cp, cp - sp], -1)
    return reward, done, obs, ncp, push, siet


def _snap(st):
    return {k: (v.copy() if v is not None else None) for k, v in st.items()}


def test_epilogue_matches_numpy_restatement(cube_model, oracle_mod):
    orc = oracle_mod.Oracle(make_blob(cube_model))
    n = 64
    st = orc.new_state(n)
    orc.reset(st, prng.split(prng.PRNGKey(3), n))
    rng = np.random.default_rng(3)
    for t in range(30):
        pre = _snap(st)
        orc.step(st, rng.uniform(-1, 1, (n, 5)).astype(np.float32))
        reward, done, obs, ncp, push, siet = _np_reward_obs(pre, st)
        np.testing.assert_allclose(st["reward"], reward, rtol=2e-6, atol=2e-6)
        np.testing.assert_array_equal(st["done"], done)
        np.testing.assert_allclose(st["obs"], obs, rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(st["info_new_cube_pos"], ncp, rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(st["metrics"][:, 0], push, rtol=2e-6)
        np.testing.assert_allclose(st["metrics"][:, 2], siet, rtol=2e-6, atol=2e-6)
        assert np.all(st["metrics"][:, 1] == 0)          # ctrl_cost is never updated (cube_env.py:203-206)


def test_prologue_ctrl_shaping(cube_model, oracle_mod):
    """ctrl = clip(prev + 0.02*a); ctrl[3] = -(1.57+q1+q2); ctrl[4] = -atan2(dy, dx+1e-5)+ctrl[0]+1.5708 (cube_env.py:146-161)."""
    orc = oracle_mod.Oracle(make_blob(cube_model))
    n = 16
    st = orc.new_state(n)
    orc.reset(st, prng.split(prng.PRNGKey(4), n))
    pre = _snap(st)
    a = np.random.default_rng(4).uniform(-1, 1, (n, 5)).astype(np.float32)
    orc.step(st, a)
    f = np.float32
    want = pre["ctrl"] + np.array([0.02, 0.02, 0.02, 0, 0], f) * a
    want[:, 3] = -((f(1.57) + pre["qpos"][:, 1]) + pre["qpos"][:, 2])
    dx = pre["info_target_pos"][:, 0] - pre["xpos"][:, 13, 0]
    dy = pre["info_target_pos"][:, 1] - pre["xpos"][:, 13, 1]
    want[:, 4] = (-np.arctan2(dy, dx + f(0.00001)).astype(f) + want[:, 0]) + f(1.5708)
    lo = cube_model.arrays["actuator_ctrlrange"][:, 0].astype(f)
    hi = cube_model.arrays["actuator_ctrlrange"][:, 1].astype(f)
    np.testing.assert_allclose(st["ctrl"], np.clip(want, lo, hi), rtol=1e-6, atol=1e-6)


def test_hand_computed_reward_cases():
    """box_target_dis < 0.005 => push_reward = 6 and siet = 3 (cube_env.py:165-167, 194)."""
    f = np.float32
    tp = np.array([[0.45, 0.1, 0.82]], f)
    pre = dict(info_target_pos=tp, info_new_cube_pos=np.array([[0.37, -0.08]], f))
    xpos = np.zeros((1, 14, 3), f); xpos[0, 13] = tp[0] + f(0.001)
    post = dict(xpos=xpos, site_xpos=np.array([[[0.2, 0.0, 0.9]]], f), qpos=np.zeros((1, 22), f))
    reward, done, obs, ncp, push, siet = _np_reward_obs(pre, post)
    assert push[0] == 6.0 and siet[0] == 3.0 and reward[0] == 6.0 + 3.0 + 1.0 + 0.0 and done[0] == 0.0


def test_reset_follows_cube_env(cube_model, oracle_mod):
    orc = oracle_mod.Oracle(make_blob(cube_model))
    keys = prng.split(prng.PRNGKey(9), 8)
    st = orc.new_state(8)
    orc.reset(st, keys)
    A = cube_model.arrays
    for e, key in enumerate(keys):
        ks = prng.split(key, 5)
        q = A["qpos0"].astype(np.float32) + prng.uniform(ks[1], (22,), -0.01, 0.01)
        q[:6] += np.array([0, -0.5422302, 0.45173569, 1.5718, -1.4794435, 1.1731174], np.float32)
        q[7] = -0.033
        tgt = prng.uniform(ks[4], (3,), np.array([0.4364427, 0.07352592, 0.82], np.float32), np.array([0.4864427, 0.12352592, 0.82], np.float32))
        cube = prng.uniform(ks[0], (3,), np.array([0.29, -0.04, 0.82], np.float32), np.array([0.34, 0.01, 0.82], np.float32))
        q[15:18], q[8:11] = cube, tgt
        for a in (11, 18):
            q[a:a + 4] /= np.linalg.norm(q[a:a + 4])           # kinematics writes the normalised quaternion back
        np.testing.assert_allclose(st["qpos"][e], q, rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(st["qvel"][e], prng.uniform(ks[2], (20,), -0.01, 0.01))
        ctrl = np.array([0, -0.73151061, 0.455936904, -1.4794435, 1.1731174], np.float32) + prng.uniform(ks[3], (5,), -0.01, 0.01)
        np.testing.assert_array_equal(st["ctrl"][e], ctrl)
        np.testing.assert_allclose(st["info_target_pos"][e], tgt, atol=1e-7)
    np.testing.assert_array_equal(st["info_new_cube_pos"], np.tile(np.float32([0.37342, -0.07989]), (8, 1)))
    assert np.all(st["reward"] == 0) and np.all(st["done"] == 0)
    assert np.any(st["qacc_warmstart"] != 0)            # qacc of the init forward (ctrl = 0) is the first warm start


def test_episode_and_autoreset_wrappers(cube_model, oracle_mod):
    """EpisodeWrapper truncation at episode_length and AutoResetWrapper restoring the cached first state
    (brax.envs.training.wrap as called at RSR/train.py:224-229)."""
    L = 5
    orc = oracle_mod.Oracle(make_blob(cube_model, episode_length=L, auto_reset=True))
    n = 8
    st = orc.new_state(n)
    orc.reset(st, prng.split(prng.PRNGKey(5), n))
    first = _snap(st)
    rng = np.random.default_rng(5)
    for t in range(1, 2 * L + 2):
        orc.step(st, rng.uniform(-1, 1, (n, 5)).astype(np.float32))
        k = (t - 1) % L + 1
        np.testing.assert_array_equal(st["info_steps"], k)
        if k == L:
            assert np.all(st["done"] == 1) and np.all(st["info_truncation"] == 1)
            for f in ("qpos", "qvel", "ctrl", "qacc_warmstart", "xpos", "site_xpos", "obs"):
                np.testing.assert_array_equal(st[f], first[f])
            # info is NOT reset by AutoResetWrapper
            assert not np.array_equal(st["info_new_cube_pos"], first["info_new_cube_pos"])
        else:
            assert np.all(st["done"] == 0) and np.all(st["info_truncation"] == 0)
    # episode_metrics follow brax: accumulate, then multiply by (1 - previous episode_done), so the first
    # step after a done leaves them at zero and the next one counts 1
    em = st["info_episode_metrics"]
    assert np.all(em[:, 1] == 0.0) and np.all(em[:, 0] == 0.0)
    orc.step(st, rng.uniform(-1, 1, (n, 5)).astype(np.float32))
    assert np.all(em[:, 1] == 1.0)
    np.testing.assert_allclose(em[:, 0], st["reward"], rtol=1e-6)


def test_unwrapped_env_has_no_episode_logic(cube_model, oracle_mod):
    orc = oracle_mod.Oracle(make_blob(cube_model))
    st = orc.new_state(4)
    orc.reset(st, prng.split(prng.PRNGKey(6), 4))
    for _ in range(3):
        orc.step(st, np.zeros((4, 5), np.float32))
    assert np.all(st["info_steps"] == 0) and np.all(st["info_episode_metrics"] == 0)


def test_golden_configs0_regression(cube_model, oracle_mod):
    g = np.load(GOLD)
    orc = oracle_mod.Oracle(make_blob(cube_model))
    st = orc.new_state(4)
    orc.reset(st, g["keys"])
    np.testing.assert_allclose(st["obs"], g["reset_obs"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(st["qacc_warmstart"], g["reset_warm"], rtol=1e-4, atol=1e-3)
    for t in range(200):
        orc.step(st, g["actions"][t])
        if t < 20:
            np.testing.assert_allclose(st["obs"], g["obs"][t], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(st["reward"], g["reward"][t], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(st["done"], g["done"][-1])
    # teacher-forced single steps from the stored snapshots
    for t in (0, 1, 50, 100, 199):
        s2 = orc.new_state(4)
        orc.reset(s2, g["keys"])
        for f in ("qpos", "qvel", "ctrl", "qacc_warmstart", "time", "xpos", "site_xpos", "info_target_pos", "info_new_cube_pos"):
            s2[f][...] = g[f"pre{t}_{f}"]
        orc.step(s2, g["actions"][t])
        np.testing.assert_allclose(s2["obs"], g[f"post{t}_obs"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(s2["qpos"], g[f"post{t}_qpos"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(s2["reward"], g[f"post{t}_reward"], rtol=1e-5, atol=1e-6)


def test_domain_randomisation_inputs(cube_model, oracle_mod):
    from rsr_mjx_amd.envs.airbot import domain_randomize
    n = 16
    dr = domain_randomize(cube_model, prng.split(prng.PRNGKey(7), n))
    A = cube_model.arrays
    s = dr["geom_friction"][:, 16, 0] / np.float32(A["geom_friction"][16, 0])
    assert np.all((s >= 0.68) & (s <= 1.32))
    np.testing.assert_allclose(dr["geom_friction"][:, 16] / A["geom_friction"][16].astype(np.float32), s[:, None].repeat(3, 1), rtol=1e-6)
    ms = dr["body_mass"][:, 13] / np.float32(0.5)
    assert np.all((ms >= 0.84) & (ms <= 1.16)) and np.all(dr["body_mass"][:, :13] == A["body_mass"][:13].astype(np.float32))
    ds = dr["dof_damping"][:, 0] / np.float32(0.2)
    assert np.all((ds >= 0.92) & (ds <= 1.08)) and np.all(dr["dof_damping"][:, 8:] == 0)
    fingers = [10, 11, 12, 13, 14, 15]
    fs = dr["geom_friction"][:, fingers, 0]
    assert np.allclose(fs, fs[:, :1]) and np.all((fs >= 0.76) & (fs <= 1.24))
    # a heavier, grippier cube changes the physics: oracle with DR differs from without
    orc = oracle_mod.Oracle(make_blob(cube_model))
    keys = prng.split(prng.PRNGKey(8), n)
    a, b = orc.new_state(n), orc.new_state(n, dr)
    orc.reset(a, keys); orc.reset(b, keys)
    act = np.random.default_rng(8).uniform(-1, 1, (n, 5)).astype(np.float32)
    for _ in range(5):
        orc.step(a, act); orc.step(b, act)
    assert np.abs(a["qvel"] - b["qvel"]).max() > 1e-6


def test_sf_variant_follows_test_airbot_py(sf_model, oracle_mod):
    """reference test/airbot.py: wrist target held within 3 cm (:180-184), thresholds 0.003, task-complete bonus 5 (:196),
    health from the done chain (:227-233), done = cube at target (:236-237)."""
    f = np.float32
    orc = oracle_mod.Oracle(make_blob(sf_model, kind="sf"))
    n = 32
    st = orc.new_state(n)
    keys = prng.split(prng.PRNGKey(12), n)
    orc.reset(st, keys)
    # reset ranges of the variant (test/airbot.py:32-39)
    assert np.all((st["qpos"][:, 15] >= 0.28) & (st["qpos"][:, 15] <= 0.29)) and np.all((st["qpos"][:, 8] >= 0.5) & (st["qpos"][:, 8] <= 0.51))
    assert np.all(st["info_last_action"] == 0)
    rng = np.random.default_rng(12)
    for t in range(25):
        pre = _snap(st)
        if t == 10:        # teleport half of the cubes next to their targets to exercise the hold / bonus / done branches
            st["qpos"][: n // 2, 15:18] = st["info_target_pos"][: n // 2] + f(0.001)
            st["xpos"][: n // 2, 13] = st["qpos"][: n // 2, 15:18]
            pre = _snap(st)
        a = rng.uniform(-1, 1, (n, 5)).astype(f)
        orc.step(st, a)
        tp, cp0 = pre["info_target_pos"], pre["xpos"][:, 13]
        act0 = pre["ctrl"][:, 0] + f(0.02) * a[:, 0]
        dx, dy = tp[:, 0] - cp0[:, 0], tp[:, 1] - cp0[:, 1]
        free = (-np.arctan2(dy, dx + f(0.00001)).astype(f) + act0) + f(1.5708)
        btd0 = np.sqrt(((tp - cp0) ** 2).sum(-1, dtype=f)).astype(f)
        want_last = np.where(btd0 < f(0.03), pre["info_last_action"], free)
        np.testing.assert_allclose(st["info_last_action"], want_last, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(st["ctrl"][:, 4], np.clip(want_last, -3.14, 3.14), rtol=1e-6, atol=1e-6)
        cp, sp = st["xpos"][:, 13], st["site_xpos"][:, 0]
        btd = np.sqrt(((tp - cp) ** 2).sum(-1, dtype=f)).astype(f)
        btd = np.where(btd < f(0.003), f(0), btd)
        push = (f(1) / (f(1) + f(3) * btd)) * f(6)
        bonus = np.where(btd < f(0.003), f(5), f(0))
        s2c = np.sqrt(((sp[:, :2] - pre["info_new_cube_pos"]) ** 2).sum(-1, dtype=f)).astype(f)
        s2c = np.where(s2c < f(0.042), f(0), s2c - f(0.042))
        siet = np.where(btd < f(0.005), f(3), (f(1) - np.tanh(f(5) * s2c).astype(f)) * f(3))
        hd = (sp[:, 2] < f(0.8)) | (sp[:, 0] > 1.0) | (sp[:, 0] < -0.6) | (sp[:, 1] > 0.3) | (sp[:, 1] < -0.3) | (cp[:, 2] < 0.6)
        health = f(1) * np.abs(hd.astype(f) - f(1))
        reward = np.clip(push + siet + health + bonus + np.where(sp[:, 2] < f(0.82), f(1), f(0)), -100, 100)
        np.testing.assert_allclose(st["reward"], reward, rtol=2e-6, atol=2e-6)
        np.testing.assert_array_equal(st["done"], (btd < f(0.003)).astype(f))
        if t == 10:
            assert st["done"][: n // 2].min() == 1.0 and np.all(st["reward"][: n // 2] >= 6 + 3 + 5)


def test_tshape_env_follows_T_shape_env_py(tshape_model, oracle_mod):
    """reference T_shape_env.py: reset :98-137, ctrl shaping :146-153 (aim at site T_tail), rewards :158-200, obs :223-234,
    restated in numpy from the source text."""
    f = np.float32
    m = tshape_model
    assert (m.nq, m.nv, m.nbody, m.ngeom, m.nsite, m.npair) == (15, 14, 14, 25, 3, 60)
    np.testing.assert_allclose(m.arrays["body_mass"][13], 0.625, rtol=1e-12)       # inertiafromgeom: 0.375 + 0.25 kg
    orc = oracle_mod.Oracle(make_blob(m, kind="tshape"))
    n = 32
    st = orc.new_state(n)
    keys = prng.split(prng.PRNGKey(13), n)
    orc.reset(st, keys)
    for e in range(4):
        ks = prng.split(keys[e], 5)
        q = m.arrays["qpos0"].astype(f) + prng.uniform(ks[1], (15,), -0.01, 0.01)
        q[:6] += np.array([0, -0.57303354, 0.381795, 1.5718, -1.3787, 1.1731174], f)
        q[11:15] /= np.linalg.norm(q[11:15])
        np.testing.assert_allclose(st["qpos"][e], q, rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(st["ctrl"][e], np.array([0, -0.57303354, 0.381795, -1.3787, 1.1731174], f) + prng.uniform(ks[3], (5,), -0.01, 0.01))
    np.testing.assert_allclose(st["info_target_w"], 10 * np.cos(0.785398163 / 2), rtol=1e-6)    # xquat[T_target][0] * 10
    assert np.all(st["info_xita"] == f(0.2876)) and np.all(st["obs"][:, 13] == f(0.2876))
    np.testing.assert_array_equal(st["info_new_T_pos"], np.tile(f([0.24739072, -0.00496255]), (n, 1)))
    rng = np.random.default_rng(13)
    for t in range(20):
        pre = _snap(st)
        a = rng.uniform(-1, 1, (n, 5)).astype(f)
        orc.step(st, a)
        # prologue
        act0 = pre["ctrl"][:, 0] + f(0.02) * a[:, 0]
        dx = pre["site_xpos"][:, 2, 0] - pre["site_xpos"][:, 0, 0]
        dy = pre["site_xpos"][:, 2, 1] - pre["site_xpos"][:, 0, 1]
        want4 = (-np.arctan2(dy, dx + f(0.00001)).astype(f) + act0) + f(1.5708)
        np.testing.assert_allclose(st["ctrl"][:, 4], np.clip(want4, -3.14, 3.14), rtol=1e-6, atol=1e-6)
        # epilogue pieces that only need record fields: obs layout and reward decomposition
        sp = st["site_xpos"][:, 0]
        np.testing.assert_array_equal(st["obs"][:, :6], st["qpos"][:, :6])
        np.testing.assert_array_equal(st["obs"][:, 6], sp[:, 2])
        np.testing.assert_array_equal(st["obs"][:, 13], st["info_xita"])
        np.testing.assert_allclose(st["obs"][:, 14:16], st["info_new_T_pos"] - sp[:, :2], rtol=1e-6, atol=1e-7)
        tail, ttail = st["site_xpos"][:, 2], st["site_xpos"][:, 1]
        ddx, ddy = ttail[:, 0] - tail[:, 0], ttail[:, 1] - tail[:, 1]
        ang = np.arctan2(ddy, ddx + f(0.00001)).astype(f)
        dist = np.sqrt(ddx * ddx + ddy * ddy).astype(f) + f(0.025)
        newT = np.stack([ddx - dist * np.cos(ang).astype(f) + tail[:, 0], ddy - dist * np.sin(ang).astype(f) + tail[:, 1]], -1)
        np.testing.assert_allclose(st["info_new_T_pos"], newT, rtol=2e-6, atol=2e-6)
        gb = pre["info_target_base_pos"] - st["obs"][:, 7:10]          # geom_xpos[base_block] recovered from the obs
        gv = pre["info_target_vertical_pos"] - st["obs"][:, 10:13]
        ba, ta = gv - gb, pre["info_target_vertical_pos"] - pre["info_target_base_pos"]
        c = (ba * ta).sum(-1) / (np.linalg.norm(ba, axis=-1) * np.linalg.norm(ta, axis=-1))
        np.testing.assert_allclose(st["info_xita"], np.arccos(np.clip(c, -1, 1)), rtol=1e-3, atol=2e-4)
        xita = st["info_xita"]
        db = np.linalg.norm(st["obs"][:, 7:10], axis=-1); db = np.where(db < 0.005, 0, db)
        dv = np.linalg.norm(st["obs"][:, 10:13], axis=-1); dv = np.where(dv < 0.005, 0, dv)
        push = (0.1515 / (1 + 10 * db) + 0.1515 / (1 + 10 * dv) + 0.66 / (1 + 6 * xita)) * 10.0
        np.testing.assert_allclose(st["metrics"][:, 0], push, rtol=1e-5)
        s2c = np.linalg.norm(sp[:, :2] - pre["info_new_T_pos"], axis=-1)
        s2c = np.where(s2c < 0.02, 0, s2c - 0.02)
        siet = (1 - np.tanh(5 * s2c)) * 3.0
        np.testing.assert_allclose(st["metrics"][:, 1], siet, rtol=1e-5, atol=1e-6)
        health = np.abs((sp[:, 2] < 0.78).astype(f) - 1)
        site_z = (sp[:, 2] < 0.83).astype(f) + 4.0 / (1 + 3 * np.abs(sp[:, 2] - 0.805))
        np.testing.assert_allclose(st["metrics"][:, 4], site_z, rtol=1e-5)
        np.testing.assert_allclose(st["reward"], np.clip(push + siet + health + site_z, -100, 100), rtol=1e-5)
        assert np.all(st["metrics"][:, 3] == 0)
        np.testing.assert_array_equal(st["done"], (st["xpos"][:, 13, 2] < 0.6).astype(f))


def test_oracle_under_address_and_undefined_sanitizers(tmp_path):
    """SURVEY 5 (race detection / sanitizers): the CPU restatement built with -fsanitize=address,undefined runs reset + steps of
    every env kind without a report (GPU sanitizers are not available on the pool; the kernels own one env per wave)."""
    import subprocess, sys, textwrap
    src = os.path.join(ROOT, "oracle", "rsr_oracle.c")
    so = tmp_path / "liboracle_f32.so"
    subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fopenmp", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-DRSR_REAL=float", "-o", str(so), src, "-lm"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    prog = textwrap.dedent(f"""
        import sys, os
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
        import numpy as np
        import oracle.oracle as OM
        OM._HERE = {str(tmp_path)!r}                      # load the sanitised build instead of oracle/liboracle_f32.so
        from conftest import make_blob, make_go2_blob
        from rsr_mjx_amd import prng
        from rsr_mjx_amd.mjcf import CompiledModel
        from rsr_mjx_amd.envs import config
        A = os.path.join({ROOT!r}, "rsr_mjx_amd", "assets")
        cases = [(make_blob(CompiledModel.load(A + "/airbot_cube.npz"), "cube", episode_length=5, auto_reset=True), 5),
                 (make_blob(CompiledModel.load(A + "/airbot_tshape.npz"), "tshape", episode_length=5, auto_reset=True), 5),
                 (make_go2_blob(config.go2_apply_overrides(CompiledModel.load(A + "/go2_rough.npz"), config.GO2_DEFAULT_CONFIG), 5, True), 12)]
        for blob, nu in cases:
            orc = OM.Oracle(blob, "f32")
            st = orc.new_state(6)
            orc.reset(st, prng.split(prng.PRNGKey(0), 6), 2)
            rng = np.random.default_rng(0)
            for t in range(12):
                orc.step(st, np.clip(rng.normal(size=(6, nu)), -1, 1).astype(np.float32), 2)
            assert np.isfinite(st["obs"]).all()
        assert any({str(so)!r} in line for line in open("/proc/self/maps")), "sanitised build was not the one loaded"
        print("sanitised oracle ok")
    """)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "sanitised oracle ok" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])


def test_numpy_wrappers_match_the_oracle_wrappers(cube_model, go2_model, oracle_mod):
    """tests/wrappers_np.py (the reference side of the action_repeat GPU tests) against the oracle's own C wrappers at
    action_repeat = 1: the numpy restatement around the PLAIN oracle env must reproduce the wrapped oracle env bit for bit -- every
    field, across two truncation boundaries and (Go2) forced falls -- and at action_repeat = 2 an outer step is two plain steps with
    the rewards summed, steps and the episode length advanced by two."""
    from conftest import make_go2_blob
    from wrappers_np import np_repeat_step
    L = 5
    pipeline = ["qpos", "qvel", "ctrl", "qacc_warmstart", "time", "xpos", "site_xpos", "obs"]
    cases = [("cube", make_blob(cube_model, episode_length=L, auto_reset=True), make_blob(cube_model), 5, 1.0, 24, ()),
             ("go2", make_go2_blob(go2_model, episode_length=L, auto_reset=True), make_go2_blob(go2_model), 12, 0.5, 4, [("priv_obs", "first_priv_obs")])]
    for kind, blob_w, blob_p, nu, astd, cap, extra in cases:
        orc_w, orc_p = oracle_mod.Oracle(blob_w), oracle_mod.Oracle(blob_p)
        n = 12
        keys = prng.split(prng.PRNGKey(15), n)
        a_st = orc_w.new_state(n); orc_w.reset(a_st, keys)
        b_st = {k: (v.copy() if v is not None else None) for k, v in a_st.items()}
        rng = np.random.default_rng(15)
        for t in range(1, 2 * L + 3):
            if kind == "go2" and t == 3:        # a fall: trunk upside down on a third of the envs (the env's own termination)
                for s_ in (a_st, b_st):
                    s_["qpos"][::3, 3:7] = np.array([0.0, 1.0, 0.0, 0.0], dtype=np.float32)
            act = np.clip(rng.normal(size=(n, nu)) * astd, -1, 1).astype(np.float32)
            orc_w.step(a_st, act)
            np_repeat_step(orc_p, b_st, act, 1, L, pipeline, extra_restore=extra)
            if kind == "go2":
                b_st["info_go2"][b_st["done"] != 0, 139:142] = 0.0
            for k, v in a_st.items():
                if v is not None and k != "stats":
                    np.testing.assert_array_equal(v.view(np.int32), b_st[k].view(np.int32), err_msg=f"{kind} {k} at step {t}")
        # action_repeat = 2 from here: two plain steps per outer step
        c_st = {k: (v.copy() if v is not None else None) for k, v in b_st.items()}
        act = np.clip(rng.normal(size=(n, nu)) * astd, -1, 1).astype(np.float32)
        steps0, len0, prev = b_st["info_steps"].copy(), b_st["info_episode_metrics"][:, 1].copy(), b_st["info_episode_done"].copy()
        steps0[b_st["done"] != 0] = 0
        np_repeat_step(orc_p, b_st, act, 2, 10 ** 6, pipeline, extra_restore=extra)
        c_st["info_steps"][c_st["done"] != 0] = 0
        orc_p.step(c_st, act); r1 = c_st["reward"].copy()
        orc_p.step(c_st, act); r2 = c_st["reward"].copy()
        np.testing.assert_array_equal(b_st["reward"], (r1 + r2).astype(np.float32))
        np.testing.assert_array_equal(b_st["info_steps"], steps0 + 2)
        np.testing.assert_array_equal(b_st["info_episode_metrics"][:, 1], np.where(prev != 0, 0, len0 + 2))
        np.testing.assert_array_equal(b_st["obs"], c_st["obs"])
