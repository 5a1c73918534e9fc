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


def test_standing_go2_foot_forces_match_the_row_weight_formula(go2_model, oracle_mod):
    """The same closed form on the condim-3 path (Go2 feet: spheres on the floor, four pyramid rows per contact,
    go2_mjx_feetonly.xml): the robot left standing on its home pose comes to rest on four contacts whose rows carry
    f = D k imp |r|, D = imp / (w (1 - imp)), w = (t + mu^2 t) 2 mu^2 / impratio; the four edges of a contact add up to 4 f along
    the normal, and the four feet together carry the robot's weight."""
    from conftest import make_go2_blob
    from rsr_mjx_amd import prng
    A = go2_model.arrays
    orc = oracle_mod.Oracle(make_go2_blob(go2_model), "f64")
    st = orc.new_state(1)
    orc.reset(st, prng.split(prng.PRNGKey(0), 1))
    for _ in range(150):                                          # 3 s on the home pose
        orc.step(st, np.zeros((1, 12), dtype=np.float32))
    assert np.abs(st["qvel"][0]).max() < 2e-2, "the robot has come to rest"
    qpos = st["qpos"][0].astype(np.float64)
    orc.forward(qpos, np.zeros(go2_model.nv), st["ctrl"][0].astype(np.float64), st["qacc_warmstart"][0].astype(np.float64))
    con = orc.get("contacts").reshape(-1, 10)
    con = con[con[:, 0] < 0]                                      # penetrating
    assert len(con) == 4 and len(set(con[:, 9].astype(int))) == 4, con
    f = orc.get("efc_force"); nefc = len(f)
    J = orc.get("efc_J").reshape(nefc, -1)
    total_mass = float(A["body_mass"].sum())
    assert abs(float(J[:, 2] @ f) - total_mass * 9.81) <= 0.01 * total_mass * 9.81      # dof 2: world z of the floating base
    dt = float(A["opt_timestep"][0])
    rows0 = nefc - 4 * len(con)                                   # the pyramid rows are the last 4 per contact, in contact order
    # contacts are listed in the order of their constraint rows (pair order)
    order = np.argsort(con[:, 9], kind="stable")
    carried = 0.0
    for slot, ci in enumerate(order):
        c = con[ci]; pair = int(c[9])
        g1, g2 = int(A["pair_geom1"][pair]), int(A["pair_geom2"][pair])
        b1, b2 = int(A["geom_bodyid"][g1]), int(A["geom_bodyid"][g2])
        solref, solimp = A["pair_solref"][pair].astype(np.float64), A["pair_solimp"][pair].astype(np.float64)
        margin = float(A["pair_margin"][pair] - A["pair_gap"][pair])
        # contact friction: the higher-priority geom's, else the maximum (the floor has priority 1 in the scene)
        p1, p2 = int(A["geom_priority"][g1]), int(A["geom_priority"][g2])
        mu = float(A["geom_friction"][g1][0] if p1 > p2 else (A["geom_friction"][g2][0] if p2 > p1 else max(A["geom_friction"][g1][0], A["geom_friction"][g2][0])))
        t = float(A["body_invweight0"][b1][0] + A["body_invweight0"][b2][0])
        w = (t + mu * mu * t) * 2.0 * mu * mu / float(A["opt_impratio"][0])
        timeconst, dampratio = max(solref[0], 2 * dt), solref[1]
        dmin, dmax, width, mid, power = solimp
        k = 1.0 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio)
        r = abs(margin - c[0])
        x = r / width
        y = 1.0 if x >= 1 else (x ** power / mid ** (power - 1) if x <= mid else 1 - (1 - x) ** power / (1 - mid) ** (power - 1))
        imp = min(max(dmin + y * (dmax - dmin), 1e-4), 0.9999)
        per_row = (imp / (w * (1.0 - imp))) * k * imp * r
        rows = f[rows0 + 4 * slot: rows0 + 4 * slot + 4]
        # at rest the four edges carry the same force up to the (small) tangential load of the stance
        assert abs(rows.sum() - 4.0 * per_row) <= 0.02 * 4.0 * per_row, (slot, rows, per_row)
        carried += rows.sum()
    assert abs(carried - total_mass * 9.81) <= 0.02 * total_mass * 9.81


def test_box_slides_with_the_coulomb_acceleration(cube_model, oracle_mod):
    """Friction saturation by hand: with gravity tilted by theta about x (equivalent to tilting the table), the 0.5 kg box on
    the table (sliding friction set to mu = 0.3: the model's own mu = 1 would tip the cube before it slides) stays put while
    tan(theta) < mu and starts to slide along y with a = g (sin(theta) - mu cos(theta)) beyond it.  The pyramid's edges lie
    along the contact frame's axes, so along an axis the pyramidal cone carries exactly mu N.  Only the first env-steps are
    compared: a box sliding on four soft corner contacts starts to rock after ~60 ms (and the restated MJX solver, whose line
    search can stall on the friction-loss kinks of the arm's rows, then leaves the solve early: DESIGN.md 2)."""
    import copy
    from rsr_mjx_amd import prng
    A = cube_model.arrays
    mu, g = 0.3, 9.81

    def box_velocity_y(tan_theta, steps):
        m = copy.copy(cube_model); m.arrays = dict(cube_model.arrays)
        fr = np.array(A["geom_friction"]).copy(); fr[:, 0] = mu
        m.arrays["geom_friction"] = fr
        th = np.arctan(tan_theta)
        m.arrays["opt_gravity"] = np.array([0.0, g * np.sin(th), -g * np.cos(th)], dtype=A["opt_gravity"].dtype)
        orc = oracle_mod.Oracle(make_blob(m), "f64")
        st = orc.new_state(1)
        orc.reset(st, prng.split(prng.PRNGKey(0), 1))
        st["qpos"][0][18:22] = [1, 0, 0, 0]; st["qpos"][0][17] = 0.8199; st["qvel"][0][14:20] = 0       # the box level and at rest
        out = []
        for _ in range(steps):
            orc.step(st, np.zeros((1, 5), dtype=np.float32))
            out.append(float(st["qvel"][0][15]))                  # dof 15: y translation of the box's free joint (dofs 14..19)
        return np.array(out), th

    dt_env = float(A["opt_timestep"][0]) * 4                       # 4 substeps per env-step
    v, th = box_velocity_y(0.5 * mu, 6)                            # below the threshold: soft-constraint creep, no acceleration
    assert np.abs(v).max() < 2e-3 and np.abs(np.diff(v)).max() / dt_env < 0.02, v
    v, th = box_velocity_y(1.5 * mu, 4)                            # above it: the Coulomb acceleration
    acc = np.diff(np.concatenate([[0.0], v])) / dt_env
    want = g * (np.sin(th) - mu * np.cos(th))
    assert np.abs(acc[1:] - want).max() <= 0.04 * want and acc.mean() <= want, (acc, want)
