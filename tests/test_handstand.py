"""Go2 Handstand / Footstand (reference mujoco_playground/_src/locomotion/go2/handstand.py on
scene_mjx_flat_terrain.xml -> go2_mjx.xml): the compiled model, and the oracle's env algebra against a numpy restatement written from
the source text.  The narrow phase of the model's capsule / cylinder geoms: tests/test_collision.py.  GPU parity: test_go2.py-style
tests at the end (marked gpu)."""
import os

import numpy as np
import pytest

import parity_envelopes as PE
from rsr_mjx_amd import mjcf, prng
from rsr_mjx_amd.envs import config as cfg, go2
from rsr_mjx_amd.model import model_fields, pack_blob

f32 = np.float32


def _blob(env, episode_length=500, auto_reset=True):
    f = model_fields(env.sys); f.update(env._fields_fn(env.sys, episode_length, auto_reset))
    return pack_blob(f)


def test_full_collision_model_structure():
    """go2_mjx.xml: every robot geom has contype 0 / conaffinity 1, the floor contype 1 / conaffinity 0 and priority 1 (go2_mjx.xml:10,
    scene_mjx_flat_terrain.xml:23): the pairs are floor-geom only, and the floor's condim 3 / friction 0.6 / default solref and solimp win."""
    m = go2.load("Go2Handstand", device="cpu").sys
    A = m.arrays
    assert (m.nq, m.nv, m.nu, m.nbody) == (19, 18, 12, 14) and A["geom_type"].shape[0] == 44
    kinds = np.bincount(A["pair_kind"], minlength=6)
    assert A["pair_geom1"].shape[0] == 30 and kinds[mjcf.PAIR_PLANE_SPHERE] == 4 and kinds[mjcf.PAIR_PLANE_CAPSULE] == 20 and kinds[mjcf.PAIR_PLANE_CYLINDER] == 6
    floor = m.id("geom", "floor")
    assert (A["pair_geom1"] == floor).all() and (A["pair_condim"] == 3).all()
    np.testing.assert_allclose(A["pair_solimp"], np.tile([0.9, 0.95, 0.001, 0.5, 2.0], (30, 1)))      # the floor's, not the foot's 0.023 width
    assert A["geom_priority"][floor] == 1 and A["geom_friction"][floor, 0] == pytest.approx(0.6)
    # a capsule given by fromto: midpoint, half length, z axis along from - to (fl_thigh1: "-0.02 0 0  -0.02 0 -0.16", radius 0.015)
    g = m.id("geom", "fl_thigh1")
    np.testing.assert_allclose(A["geom_pos"][g], [-0.02, 0, -0.08], atol=1e-12)
    np.testing.assert_allclose(A["geom_size"][g][:2], [0.015, 0.08], atol=1e-12)
    np.testing.assert_allclose(A["geom_quat"][g], [1, 0, 0, 0], atol=1e-12)
    g = m.id("geom", "fl_calf1")                                                   # "0 0 0  0.02 0 -0.13": tilted about y
    z = mjcf.quat_to_mat(A["geom_quat"][g])[:, 2]
    np.testing.assert_allclose(z, np.array([-0.02, 0, 0.13]) / np.hypot(0.02, 0.13), atol=1e-12)
    # base.py:25-31 overrides with the task's Kp / Kd
    assert A["actuator_gainprm"][0, 0] == 35.0 and A["actuator_biasprm"][0, 1] == -35.0 and A["dof_damping"][6] == 0.5 and A["opt_timestep"][0] == 0.004
    assert go2.load("Go2Handstand", device="cpu").observation_sizes == {"state": (45,), "privileged_state": (94,)}
    with pytest.raises(ValueError):
        go2.load("Go2Getup")                                                       # not built


def _noise(key, n, level, scale):
    u = prng.uniform(key, (n,), 0.0, 1.0).astype(f32)
    return (f32(2.0) * u - f32(1.0)) * f32(level) * f32(scale)


@pytest.mark.parametrize("name", ["Go2Handstand", "Go2Footstand"])
def test_reset_and_step_algebra(oracle_mod, name):
    """handstand.py:119-195, restated in numpy from the source: reset draws (bernoulli, xy, yaw, base velocity), ctrl = qpos[7:], the
    five noise draws of the observation in the reference's order (gyro, gravity, joint angles, joint velocities, linvel), motor targets
    = previous targets + 0.3 action, and the reward from the post-step state."""
    env = go2.load(name, device="cpu")
    c = env._config
    orc = oracle_mod.Oracle(_blob(env, 500, False))
    n = 64
    keys = prng.split(prng.PRNGKey(7), n)
    st = orc.new_state(n)
    orc.reset(st, keys)
    home = env.sys.arrays["key_qpos"][env.sys.names["key"]["home"]].astype(f32)
    for e in range(0, n, 9):
        rng = keys[e]
        rng, reset_rng = prng.split(rng, 2)
        assert not prng.uniform(reset_rng, (), 0.0, 1.0) < c["init_from_crouch"]
        rng, k = prng.split(rng, 2)
        dxy = prng.uniform(k, (2,), -0.5, 0.5).astype(f32)
        rng, k = prng.split(rng, 2)
        yaw = prng.uniform(k, (1,), -3.14, 3.14).astype(f32)[0]
        rng, k = prng.split(rng, 2)
        v6 = prng.uniform(k, (6,), -0.5, 0.5).astype(f32)
        np.testing.assert_array_equal(st["qpos"][e, :2], home[:2] + dxy)
        np.testing.assert_allclose(st["qpos"][e, 3:7], [np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)], atol=2e-7)
        np.testing.assert_array_equal(st["qvel"][e, :6], v6)
        np.testing.assert_array_equal(st["ctrl"][e], st["qpos"][e, 7:])
        # the observation's draws: gyro, gravity, joint angles, joint velocities, linvel
        nz = c["noise_config"]
        ks = []
        for _ in range(5):
            rng, k = prng.split(rng, 2); ks.append(k)
        np.testing.assert_allclose(st["obs"][e, 9:21], st["qpos"][e, 7:] + _noise(ks[2], 12, nz["level"], nz["scales"]["joint_pos"]) - home[7:], atol=1e-6)
        np.testing.assert_allclose(st["obs"][e, 21:33], st["qvel"][e, 6:] + _noise(ks[3], 12, nz["level"], nz["scales"]["joint_vel"]), atol=1e-6)
        grav_noise = _noise(ks[1], 3, nz["level"], nz["scales"]["gravity"])
        q = st["qpos"][e, 3:7].astype(np.float64); R = mjcf.quat_to_mat(q / np.linalg.norm(q))
        np.testing.assert_allclose(st["obs"][e, 6:9], R.T @ np.array([0, 0, -1.0]) + grav_noise, atol=1e-6)
        np.testing.assert_array_equal(st["info_go2"][e, 137:139].view(np.uint32), rng)
        assert not st["obs"][e, 33:45].any() and st["info_go2"][e, 0] == 0
    # privileged_state: state + clean readings; joint angles NOT relative to the default pose, torso height last (handstand.py:246-259)
    np.testing.assert_array_equal(st["priv_obs"][:, :45], st["obs"])
    np.testing.assert_array_equal(st["priv_obs"][:, 57:69], st["qpos"][:, 7:])
    np.testing.assert_array_equal(st["priv_obs"][:, 69:81], st["qvel"][:, 6:])
    imu = env.sys.id("site", "imu")
    np.testing.assert_array_equal(st["priv_obs"][:, 93], st["site_xpos"][:, imu, 2])
    assert not st["priv_obs"][:, 94:].any()
    # ---- steps ----
    unwanted, feet, joint_ids, fwd_des, z_des = cfg._HANDSTAND_VARIANTS[env._variant]
    A = env.sys.arrays
    lo, hi = A["jnt_range"][1:, 0], A["jnt_range"][1:, 1]
    cc, rr = (lo + hi) / 2, hi - lo
    soft_lo, soft_hi = cc - 0.5 * rr * c["soft_joint_pos_limit_factor"], cc + 0.5 * rr * c["soft_joint_pos_limit_factor"]
    feet_sites = [env.sys.id("site", s) for s in feet]
    rng_np = np.random.default_rng(3)
    last_act = np.zeros((n, 12), f32)
    checked = 0
    for t in range(12):
        act = np.clip(rng_np.normal(size=(n, 12)) * 0.3, -1, 1).astype(f32)
        ctrl_before = st["ctrl"].copy()
        orc.step(st, act)
        np.testing.assert_array_equal(st["ctrl"], ctrl_before + act * f32(c["action_scale"]))        # no clipping of the targets
        np.testing.assert_array_equal(st["info_go2"][:, 4:16], act)
        assert (st["info_go2"][:, 0] == t + 1).all()
        np.testing.assert_array_equal(st["obs"][:, 33:45], last_act)                                   # the obs sees the PREVIOUS action
        q = st["qpos"].astype(np.float64)
        force = st["priv_obs"][:, 81:93].astype(np.float64)
        for e in range(n):
            # site_xmat / site_xpos after a step are those of the last forward pass, one substep before the final integration (as in
            # MJX): undo that substep's quaternion step, q_new = q_old * exp(w_new dt), with the post-step angular velocity
            w = st["qvel"][e, 3:6].astype(np.float64); ang = np.linalg.norm(w) * c["sim_dt"]
            dq = np.concatenate([[np.cos(-ang / 2)], np.sin(-ang / 2) * w / max(np.linalg.norm(w), 1e-30)])
            R = mjcf.quat_to_mat(mjcf.quat_mul(q[e, 3:7] / np.linalg.norm(q[e, 3:7]), dq))
            h = min(float(st["site_xpos"][e, imu, 2]), z_des)
            terms = dict(height=np.exp(-(z_des - h)), orientation=(0.5 * float(R[:, 0] @ np.array(fwd_des)) + 0.5) ** 2,
                         contact=float(any(st["site_xpos"][e, s, 2] - 0.023 < 0 for s in feet_sites)),
                         action_rate=float(((act[e] - last_act[e]).astype(np.float64) ** 2).sum()), torques=float((force[e] ** 2).sum()),
                         termination=float(st["done"][e]),
                         dof_pos_limits=float((-np.clip(q[e, 7:] - soft_lo, None, 0) + np.clip(q[e, 7:] - soft_hi, 0, None)).sum()),
                         pose=float(((q[e, 7:][list(joint_ids)] - home[7:][list(joint_ids)]) ** 2).sum()),
                         stay_still=float(st["qvel"][e, 0] ** 2 + st["qvel"][e, 1] ** 2 + st["qvel"][e, 5] ** 2),
                         energy=float((np.abs(st["qvel"][e, 6:]) * np.abs(force[e])).sum()))
            sc = c["reward_config"]["scales"]
            for k_i, k in enumerate(cfg.HANDSTAND_REWARDS):
                if k != "dof_acc":
                    assert st["metrics"][e, k_i] == pytest.approx(terms[k] * sc[k], rel=2e-5, abs=5e-6), (t, e, k)
            total = sum(terms[k] * sc[k] for k in terms)
            assert st["reward"][e] == pytest.approx(np.clip(total * c["ctrl_dt"], 0, 1e4), rel=2e-5, abs=2e-6)
            if R[2, 2] < -0.2501:
                assert st["done"][e] == 1
            checked += 1
        last_act = act
    assert checked == 12 * n


def test_unwanted_contact_and_fall_terminate(oracle_mod):
    """handstand.py:188-195: a thigh capsule on the floor, or the trunk upside down (upvector z < -0.25), ends the episode; four feet on
    the floor do not."""
    env = go2.load("Go2Handstand", device="cpu")
    orc = oracle_mod.Oracle(_blob(env, 500, False))
    n = 3
    st = orc.new_state(n)
    orc.reset(st, prng.split(prng.PRNGKey(1), n))
    home = env.sys.arrays["key_qpos"][0].astype(f32)
    st["qvel"][:] = 0
    st["qpos"][:] = home
    st["qpos"][1, 2] = 0.09                      # trunk lowered until the thighs (and hips) lie on the floor
    st["qpos"][2, 3:7] = [0, 1, 0, 0]           # rolled over: up vector (0, 0, -1)
    st["qpos"][2, 2] = 0.5
    st["ctrl"][:] = st["qpos"][:, 7:]
    orc.step(st, np.zeros((n, 12), f32))
    assert st["done"].tolist() == [0.0, 1.0, 1.0], st["done"]
    assert st["stats"][1, 2] > 4                 # more contacts than the four feet


def test_truncation_and_autoreset(oracle_mod):
    env = go2.load("Go2Handstand", device="cpu")
    orc = oracle_mod.Oracle(_blob(env, 4, True))
    n = 16
    st = orc.new_state(n)
    orc.reset(st, prng.split(prng.PRNGKey(2), n))
    first = {k: st[k].copy() for k in ("qpos", "obs", "priv_obs")}
    for t in range(1, 9):
        orc.step(st, np.zeros((n, 12), f32))
        if t % 4 == 0:
            assert (st["done"] == 1).all()
            np.testing.assert_array_equal(st["qpos"], first["qpos"]); np.testing.assert_array_equal(st["obs"], first["obs"])
            np.testing.assert_array_equal(st["priv_obs"], first["priv_obs"])


# ------------------------------------------------------------------------------------------------ GPU parity (HIP stepper vs oracle)
GNAME = {"priv_obs": "privileged_obs", "first_priv_obs": "first_privileged_obs"}
FIELDS = ["qpos", "qvel", "ctrl", "qacc_warmstart", "time", "xpos", "site_xpos", "obs", "reward", "done", "metrics", "info_go2",
          "info_steps", "info_truncation", "info_episode_done", "info_episode_metrics", "first_qpos", "first_qvel", "first_ctrl",
          "first_warmstart", "first_time", "first_xpos", "first_site_xpos", "first_obs", "priv_obs", "first_priv_obs"]


def _key(info):
    return np.ascontiguousarray(info[:, 137:139]).view(np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("name,randomize", [("Go2Handstand", True), ("Go2Footstand", False)])
def test_handstand_hip_parity(oracle_mod, name, randomize):
    """Reset: the PRNG-only parts bit for bit; teacher-forced env-steps at three rollout depths inside the measured envelopes; and a
    step from poses with the trunk lowered onto the floor (thigh / calf capsules, hip and trunk cylinders in contact: the pair kinds a
    rollout from the home pose hardly meets), where done / contact counts must agree exactly."""
    import torch
    n = 1024
    jenv = go2.load(name)
    dr = go2.domain_randomize(jenv.sys, prng.split(prng.PRNGKey(12), n)) if randomize else None
    env = go2.wrap_for_brax_training(jenv, n, episode_length=500, randomization_fn=(lambda sys: dr) if randomize else None)
    assert env.observation_size == 45 and env.action_size == 12 and env.dims.ncon_max == 12
    orc = oracle_mod.Oracle(env.blob); orc.set_ncon_cap(env.dims.ncon_max)
    keys = prng.split(prng.PRNGKey(21), n)
    odr = None if dr is None else {{"actuator_gainprm": "gainprm", "actuator_biasprm": "biasprm"}.get(k, k): v for k, v in dr.items()}
    st = orc.new_state(n, odr)
    orc.reset(st, keys)
    state = env.reset(keys)
    torch.cuda.synchronize()
    get = lambda k: env.view(GNAME.get(k, k)).cpu().numpy().reshape(st[k].shape)
    for k in ("qvel", "ctrl", "first_qvel", "first_ctrl"):
        np.testing.assert_array_equal(get(k), st[k], err_msg=k)
    np.testing.assert_array_equal(_key(get("info_go2")), _key(st["info_go2"]))
    np.testing.assert_allclose(get("qpos"), st["qpos"], atol=1e-6)
    for k in ("obs", "priv_obs", "xpos", "site_xpos"):
        PE.check("handstand", "reset", k, get(k), st[k], tag="after reset") if k in PE.ENV["handstand"]["reset"] else None
    assert set(state.info) >= {"step", "rng", "last_act", "steps", "truncation"} and state.obs_dict["privileged_state"].shape == (n, 94)
    rng = np.random.default_rng(21)
    for depth in (0, 5, 30):
        for _ in range(depth):
            orc.step(st, np.clip(rng.normal(size=(n, 12)) * 0.3, -1, 1).astype(f32))
        for k in FIELDS:
            env.view(GNAME.get(k, k)).copy_(torch.from_numpy(st[k].reshape(n, -1)))
        a = np.clip(rng.normal(size=(n, 12)) * 0.3, -1, 1).astype(f32)
        orc.step(st, a)
        state = env.step(state, a)
        torch.cuda.synchronize()
        for k in ("done", "info_steps", "info_truncation", "ctrl", "time"):
            np.testing.assert_array_equal(get(k), st[k], err_msg=f"{k} depth {depth}")
        np.testing.assert_array_equal(_key(get("info_go2")), _key(st["info_go2"]))
        np.testing.assert_array_equal(get("info_go2")[:, :16], st["info_go2"][:, :16])            # step counter, last action
        np.testing.assert_array_equal(get("stats")[:, 2:], st["stats"][:, 2:])                    # active / dropped contacts
        phase = "reset" if depth == 0 else "rollout"
        for k in ("qpos", "xpos", "site_xpos", "obs", "reward", "metrics", "qvel", "qacc_warmstart", "priv_obs"):
            PE.check("handstand", phase, k, get(k), st[k], tag=f"{name} depth {depth}", outliers=2)
    # ---- poses on the floor: capsules and cylinders in contact ----
    home = jenv.sys.arrays["key_qpos"][0].astype(f32)
    st["qpos"][:] = home; st["qvel"][:] = 0
    z = np.linspace(0.06, 0.2, n).astype(f32)
    st["qpos"][:, 2] = z
    tilt = rng.normal(size=(n, 4)).astype(f32) * 0.15 + np.array([1, 0, 0, 0], f32)
    st["qpos"][:, 3:7] = tilt / np.linalg.norm(tilt, axis=1, keepdims=True)
    st["qpos"][:, 7:] += rng.normal(size=(n, 12)).astype(f32) * 0.2
    st["ctrl"][:] = st["qpos"][:, 7:]; st["qacc_warmstart"][:] = 0; st["done"][:] = 0
    for k in FIELDS:
        env.view(GNAME.get(k, k)).copy_(torch.from_numpy(st[k].reshape(n, -1)))
    a = np.zeros((n, 12), f32)
    orc.step(st, a); state = env.step(state, a)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(get("done"), st["done"])
    np.testing.assert_array_equal(get("stats")[:, 2:], st["stats"][:, 2:])
    assert (st["stats"][:, 2] > 4).mean() > 0.5 and (st["stats"][:, 3] > 0).any()      # capsule / cylinder contacts, and the cap of 12 is met
    assert st["done"].mean() > 0.5
    for k in ("reward", "metrics"):                                                     # computed from the (pre-reset) step on both sides
        PE.check("handstand", "rollout", k, get(k), st[k], tag="floor poses", quantiles=False, outliers=2)      # (violent states: the bound, not the rollout's quantiles)


@pytest.mark.gpu
def test_handstand_truncation_and_autoreset_on_device(oracle_mod):
    import torch
    n, L = 1024, 5
    env = go2.wrap_for_brax_training(go2.load("Go2Handstand"), n, episode_length=L)
    orc = oracle_mod.Oracle(env.blob); orc.set_ncon_cap(env.dims.ncon_max)
    keys = prng.split(prng.PRNGKey(41), n)
    st = orc.new_state(n); orc.reset(st, keys)
    state = env.reset(keys)
    get = lambda k: env.view(GNAME.get(k, k)).cpu().numpy().reshape(st[k].shape)
    first_obs, first_qpos = st["first_obs"].copy(), st["first_qpos"].copy()
    rng = np.random.default_rng(41)
    for t in range(1, 2 * L + 1):
        for k in FIELDS:
            env.view(GNAME.get(k, k)).copy_(torch.from_numpy(st[k].reshape(n, -1)))
        a = np.clip(rng.normal(size=(n, 12)) * 0.3, -1, 1).astype(f32)
        orc.step(st, a); state = env.step(state, a)
        torch.cuda.synchronize()
        for k in ("done", "info_steps", "info_truncation", "info_episode_done", "ctrl", "time"):
            np.testing.assert_array_equal(get(k), st[k], err_msg=f"{k} at step {t}")
        done = st["done"] != 0
        if done.any():
            np.testing.assert_array_equal(get("obs")[done], first_obs[done])
            np.testing.assert_array_equal(get("qpos")[done], first_qpos[done])
        if t % L == 0:
            assert done.all() and (st["info_truncation"][st["info_steps"] == L] >= 0).all()
        em = PE.scaled_err(get("info_episode_metrics"), st["info_episode_metrics"])
        assert np.quantile(em, 0.99) <= 1e-4, (t, float(np.quantile(em, 0.99)))
