"""Hand-derived closed forms that pin the two items SURVEY B.10 marks "(?)" -- the per-joint actuator force clamp and the
pyramidal contact row weight -- plus the oracle's line-search switches.  No reference fixture exists for the path (parity with
MJX stays unpinned); these check the restatement against MuJoCo's published formulas on cases simple enough to solve by hand.
"""
import numpy as np
import pytest

from conftest import make_blob


def _zero_state(model):
    qpos = model.arrays["qpos0"].copy().astype(np.float64)
    return qpos, np.zeros(model.nv)


def test_actuator_force_is_clamped_per_actuator_then_per_joint(cube_model, oracle_mod):
    """cube.xml:50-74, 175-180: position servos force = kp (ctrl - q), clamped to forcerange (+-300), and the joint's total
    actuator force clamped to actuatorfrcrange (+-24 for joints 1-3, +-8 for joints 5-6).  Static pushes of three sizes:
    unsaturated (kp e), joint-clamped (kp e = 60 -> 24), and both clamps (kp e = 400 -> 300 -> 24)."""
    A = cube_model.arrays
    orc = oracle_mod.Oracle(make_blob(cube_model), "f64")
    qpos, qvel = _zero_state(cube_model)
    kp = A["actuator_gainprm"][:, 0]
    trn = A["actuator_trnid"].reshape(-1, 2)[:, 0] if A["actuator_trnid"].ndim > 1 else A["actuator_trnid"]
    for u in range(cube_model.nu):
        j = int(trn[u]); dof = int(A["jnt_dofadr"][j]); qa = int(A["jnt_qposadr"][j])
        lim = A["jnt_actfrcrange"][j][1]
        assert A["jnt_actfrclimited"][j] == 1 and A["actuator_forcelimited"][u] == 1 and lim in (24.0, 8.0)
        for target_force in (0.5 * lim, 2.5 * lim, 400.0, -0.5 * lim, -400.0):
            ctrl = qpos[[int(A["jnt_qposadr"][int(t)]) for t in trn]].copy()          # every servo at rest ...
            ctrl[u] = qpos[qa] + target_force / kp[u]                                    # ... but this one: kp e = target_force
            lo, hi = A["actuator_ctrlrange"][u]
            if A["actuator_ctrllimited"][u] and not (lo <= ctrl[u] <= hi):
                continue                                                                 # (ctrl clamp would change the premise)
            orc.forward(qpos, qvel, ctrl)
            expect = np.clip(np.clip(target_force, -300.0, 300.0), -lim, lim)
            got = orc.get("qfrc_actuator")[dof]
            assert abs(got - expect) <= 1e-9 * max(1.0, abs(expect)), (u, target_force, got, expect)
            assert abs(orc.get("actuator_force")[u] - np.clip(target_force, -300.0, 300.0)) <= 1e-9 * 300.0


def test_resting_cube_penetration_matches_the_row_weight_formula(cube_model, oracle_mod):
    """A 0.5 kg box at rest on the table (cube.xml:164: solref 0.01 1, solimp 0.8 1 0.01, condim 4, pyramidal): four corner
    contacts, six pyramid rows each.  At rest qacc = 0 and v = 0, so every row has Jaref = -aref = k imp r (r = dist - margin < 0)
    and carries force D k imp |r| with D = imp / (w (1 - imp)), w = (t + mu^2 t) 2 mu^2 / impratio, t the sum of the two bodies'
    translational invweight0.  The six edges of a contact add up to 6 f along the normal (the tangential parts cancel), so
    24 f = m g fixes |r|: a scalar equation in |r| through imp(|r|).  The oracle, left to settle, must sit at that depth."""
    A = cube_model.arrays
    orc = oracle_mod.Oracle(make_blob(cube_model), "f64")
    n = 1
    st = orc.new_state(n)
    from rsr_mjx_amd import prng
    orc.reset(st, prng.split(prng.PRNGKey(0), n))
    for _ in range(400):                                          # 400 env-steps = 4 s: the arm holds its pose, the box settles
        orc.step(st, np.zeros((n, 5), dtype=np.float32))
    qpos, qvel = st["qpos"][0].astype(np.float64), st["qvel"][0].astype(np.float64) * 0.0
    orc.forward(qpos, qvel, st["ctrl"][0].astype(np.float64), st["qacc_warmstart"][0].astype(np.float64))
    con = orc.get("contacts").reshape(-1, 10)
    cube_b = cube_model.id("body", "cube_for_push")
    mine = con[(con[:, 7] == cube_b) | (con[:, 8] == cube_b)]
    assert len(mine) == 4, mine
    pair = int(mine[0, 9])
    assert (mine[:, 9] == pair).all()
    g1, g2 = int(A["pair_geom1"][pair]), int(A["pair_geom2"][pair])
    b1, b2 = int(A["geom_bodyid"][g1]), int(A["geom_bodyid"][g2])
    solref, solimp = A["pair_solref"][pair].astype(np.float64), A["pair_solimp"][pair].astype(np.float64)
    margin = float(A["pair_margin"][pair] - A["pair_gap"][pair])
    mu = float(max(A["geom_friction"][g1][0], A["geom_friction"][g2][0]))
    t = float(A["body_invweight0"][b1][0] + A["body_invweight0"][b2][0])
    w = (t + mu * mu * t) * 2.0 * mu * mu / float(A["opt_impratio"][0])
    dt = float(A["opt_timestep"][0])
    timeconst, dampratio = max(solref[0], 2 * dt), solref[1]
    dmin, dmax, width, mid, power = solimp
    k = 1.0 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio)
    mass, g = float(A["body_mass"][cube_b]), 9.81

    def imp_of(r):
        x = abs(r) / width
        if x >= 1:
            return dmax
        y = x ** power / mid ** (power - 1) if x <= mid else 1 - (1 - x) ** power / (1 - mid) ** (power - 1)
        return dmin + y * (dmax - dmin)

    def total_normal_force(r):                                    # 4 contacts x 6 rows
        imp = min(max(imp_of(r), 1e-4), 0.9999)
        return 24.0 * (imp / (w * (1.0 - imp))) * k * imp * abs(r)

    lo, hi = 0.0, 0.05
    for _ in range(200):
        mid_r = 0.5 * (lo + hi)
        if total_normal_force(mid_r) < mass * g: lo = mid_r
        else: hi = mid_r
    depth = 0.5 * (lo + hi)
    got = margin - mine[:, 0]                                     # |r| of the four corners
    assert np.abs(st["qvel"][0][14:20]).max() < 1e-3, "the box has come to rest"
    assert np.abs(got - depth).max() <= 0.02 * depth, (got, depth)
    assert abs(got.mean() - depth) <= 5e-3 * depth, (got.mean(), depth)
    # and the rows' forces add up to the weight
    f = orc.get("efc_force"); nefc = len(f)
    Jz = orc.get("efc_J").reshape(nefc, -1)[:, 16]                # dof 16 = z translation of the box's free joint (dofs 14..19)
    assert abs(float(Jz @ f) - mass * g) <= 1e-3 * mass * g


def test_line_search_switches(cube_model, tshape_model, oracle_mod):
    """oracle.set_ls_cycle: the exact cut of the bracket update's limit cycles leaves every result bit-identical and removes the
    50-iteration searches.  oracle.set_ls_rule(2): the fp32 noise-floor stop the HIP kernel uses moves qacc by no more than
    rounding noise.  (Both switches exist so that either side can run either rule; the kernel runs with both on.)"""
    from rsr_mjx_amd import prng
    for model, kind in ((cube_model, "cube"), (tshape_model, "tshape")):
        orc = oracle_mod.Oracle(make_blob(model, kind, episode_length=1200, auto_reset=True))
        n = 48
        keys = prng.split(prng.PRNGKey(9), n)
        rng = np.random.default_rng(9)
        acts = np.clip(rng.normal(size=(12, n, 5)), -1, 1).astype(np.float32)

        def run(rule, cycle):
            orc.set_ls_rule(rule, 1.0); orc.set_ls_cycle(cycle); orc.ls_counters(reset=True)
            st = orc.new_state(n); orc.reset(st, keys)
            for a in acts:
                orc.step(st, a)
            calls, iters, hist = orc.ls_counters()
            return st, calls, iters, hist
        try:
            base, c0, i0, h0 = run(0, False)
            cut, c1, i1, h1 = run(0, True)
            for k in ("qpos", "qvel", "qacc_warmstart", "obs", "reward"):
                np.testing.assert_array_equal(base[k], cut[k], err_msg=f"{kind} {k}: the cycle cut must not change a bit")
            assert c0 == c1 and h0[40:].sum() > 0.02 * c0 and h1[20:].sum() <= 0.002 * c1 and i1 < 0.6 * i0, (kind, c0, i0, i1)   # (periods above 8 run on)
            kern, c2, i2, h2 = run(2, True)
            assert i2 < 0.6 * i1
            # same trajectories to rounding noise over 12 contact-rich steps (obs is what the learner sees)
            err = np.abs(kern["obs"].astype(np.float64) - base["obs"]).max(axis=1) / np.maximum(1.0, np.abs(base["obs"]).max(axis=1))
            assert np.quantile(err, 0.9) <= 1e-5, (kind, float(np.quantile(err, 0.9)))
        finally:
            orc.set_ls_rule(0); orc.set_ls_cycle(False)
