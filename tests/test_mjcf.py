"""Model compiler: structure of the compiled Airbot cube model and blob packing."""
import os

import numpy as np
import pytest

from conftest import CUBE_XML, make_blob
from rsr_mjx_amd import mjcf
from rsr_mjx_amd.model import topology_tables, unpack_blob


def test_cube_model_dimensions_and_layout(cube_model):
    m = cube_model
    assert (m.nq, m.nv, m.nu, m.nbody, m.njnt, m.ngeom, m.nsite, m.npair) == (22, 20, 5, 14, 10, 23, 1, 45)
    A = m.arrays
    # qpos addresses used by the env (cube_env.py:66-82)
    assert [int(A["jnt_qposadr"][m.id("joint", f"joint{i}")]) for i in range(1, 7)] == [0, 1, 2, 3, 4, 5]
    assert int(A["jnt_qposadr"][m.id("joint", "endright")]) == 6 and int(A["jnt_qposadr"][m.id("joint", "endleft")]) == 7
    assert m.id("body", "target_pos") == 12 and m.id("body", "cube_for_push") == 13
    assert m.id("geom", "ground") == 0 and m.id("geom", "table-b") == 16 and m.id("geom", "geom_for_push") == 22
    # pair kinds: 15 plane-box + 30 box-box, all condim 4
    assert int((A["pair_kind"] == mjcf.PAIR_PLANE_BOX).sum()) == 15 and int((A["pair_kind"] == mjcf.PAIR_BOX_BOX).sum()) == 30
    assert set(A["pair_condim"].tolist()) == {4}
    # options (cube.xml:20) and MuJoCo defaults
    assert A["opt_timestep"][0] == 0.0025 and A["opt_iterations"][0] == 20 and A["opt_integrator"][0] == mjcf.INT_IMPLICITFAST
    assert A["opt_ls_iterations"][0] == 50 and A["opt_tolerance"][0] == 1e-8
    # free bodies: invweight = 1/m and 1/I
    np.testing.assert_allclose(A["body_invweight0"][13], [2.0, 1.0 / 0.0005333], rtol=1e-9)
    np.testing.assert_allclose(A["dof_invweight0"][14:17], 2.0, rtol=1e-9)
    # position servos: gain kp, bias (0, -kp, 0)
    np.testing.assert_allclose(A["actuator_gainprm"][:, 0], [1000, 1000, 1000, 350, 100])
    np.testing.assert_allclose(A["actuator_biasprm"][:, 1], [-1000, -1000, -1000, -350, -100])
    # contact mixing: cube (solref .01) on table (.01) -> .01 ; arm default (.02) on table -> .015
    p = [i for i in range(m.npair) if A["pair_geom1"][i] == 16 and A["pair_geom2"][i] == 22][0]
    np.testing.assert_allclose(A["pair_solref"][p], [0.01, 1.0])
    p = [i for i in range(m.npair) if A["pair_geom1"][i] == 3 and A["pair_geom2"][i] == 16][0]
    np.testing.assert_allclose(A["pair_solref"][p], [0.015, 1.0])


@pytest.mark.skipif(not os.path.exists(CUBE_XML), reason="reference tree not present (GPU box)")
def test_committed_asset_is_the_compiler_output(cube_model):
    fresh = mjcf.compile_mjcf(CUBE_XML)
    assert set(fresh.arrays) == set(cube_model.arrays)
    for k, v in fresh.arrays.items():
        np.testing.assert_allclose(v, cube_model.arrays[k], rtol=1e-12, atol=1e-14, err_msg=k)
    assert fresh.names == cube_model.names


def test_blob_roundtrip_and_topology(cube_model):
    blob = make_blob(cube_model, episode_length=1200, auto_reset=True)
    f = unpack_blob(blob)
    np.testing.assert_array_equal(f["dims"], [22, 20, 5, 14, 10, 23, 1, 1, 45])
    np.testing.assert_allclose(f["body_mass"], cube_model.arrays["body_mass"].astype(np.float32))
    assert f["env_int"].tolist() == [0, 4, 1200, 3, 23, 3]
    t = topology_tables(cube_model)
    anc = t["dof_ancmask"].view(np.uint32)
    assert anc[7] == 0b10111111            # endleft hangs off link6, not off endright
    assert anc[19] == 0b111111 << 14
    sub = t["body_submask"].view(np.uint32)
    assert sub[8] == (1 << 8) | (1 << 9) | (1 << 10)
    assert t["fric_dofs"].tolist() == list(range(8)) and t["limit_jnts"].tolist() == list(range(8))


def test_euler_and_inertia_helpers():
    q = mjcf.euler_to_quat([0, 0, 1.5708])
    np.testing.assert_allclose(q, [np.cos(0.7854), 0, 0, np.sin(0.7854)], atol=1e-12)
    w, iq = mjcf.MjcfCompiler._principal(np.diag([1.0, 3.0, 2.0]))
    np.testing.assert_allclose(w, [3, 2, 1])
    R = mjcf.quat_to_mat(iq)
    np.testing.assert_allclose(R @ np.diag(w) @ R.T, np.diag([1.0, 3.0, 2.0]), atol=1e-12)


def test_lane_records_restate_the_model_tables(cube_model):
    """model.lane_records: the per-lane quads the kernel's stages fetch must carry exactly what the per-field tables hold
    (layout documented in lane_records; the device side is enum LaneQuad in csrc/rsr_device.hpp, whose count the library
    checks against the blob at rsr_model_create)."""
    import re
    from rsr_mjx_amd import model as M
    from rsr_mjx_amd.model import LANE_QUADS, lane_records, topology_tables
    m = cube_model
    A, topo = m.arrays, topology_tables(m)
    rec = lane_records(m, topo)
    fv = rec.view(np.float32)
    assert rec.shape == (LANE_QUADS, 64, 4) and rec.dtype == np.int32
    f32 = lambda x: np.asarray(x, dtype=np.float32)
    for b in range(m.nbody):
        assert list(rec[0, b]) == [A["body_parentid"][b], topo["body_depth"][b], topo["body_jtype"][b], topo["body_qposadr"][b]]
        np.testing.assert_array_equal(fv[2, b], f32(A["body_quat"][b]))
        np.testing.assert_array_equal([fv[4, b, 2], fv[4, b, 3], fv[6, b, 0]], f32(A["body_ipos"][b]))
        assert rec[6, b, 1] == A["body_rootid"][b] and np.uint32(rec[6, b, 2]) == np.uint32(topo["body_submask"][b])
        np.testing.assert_array_equal(fv[7, b, :3], f32(A["body_inertia"][b]))
    for j in range(m.njnt):
        jb = int(A["jnt_bodyid"][j])
        assert list(rec[8, j, :3]) == [jb, A["body_parentid"][jb], A["jnt_type"][j]]
        np.testing.assert_array_equal([fv[11, j, 2], fv[11, j, 3], fv[12, j, 0]], f32(A["jnt_axis"][j]))
        assert list(rec[12, j, 1:3]) == [A["jnt_qposadr"][j], A["jnt_dofadr"][j]]
    for i in range(m.nv):
        u = int(topo["dof_act"][i])
        assert rec[19, i, 0] == u and np.uint32(rec[18, i, 1]) == np.uint32(topo["dof_ancmask"][i])
        if u >= 0:
            np.testing.assert_array_equal([fv[20, i, 2], fv[20, i, 3], fv[21, i, 0], fv[21, i, 1]],
                                          f32([A["actuator_gainprm"][u][0], *A["actuator_biasprm"][u][:3]]))
    for l, j in enumerate(topo["limit_jnts"]):
        assert list(rec[26, l, :2]) == [A["jnt_qposadr"][j], A["jnt_dofadr"][j]] and rec[29, l, 1] == j
        np.testing.assert_array_equal(fv[26, l, 2:], f32(A["jnt_range"][j]))
        np.testing.assert_array_equal([*fv[28, l], fv[29, l, 0]], M.impedance_consts(A["jnt_solimp"][j]))
    for q in range(m.npair):
        g1, g2 = int(A["pair_geom1"][q]), int(A["pair_geom2"][q])
        assert list(rec[30, q, :3]) == [g1, g2, A["pair_kind"][q]]
        np.testing.assert_array_equal(fv[32, q, :3], f32(A["geom_size"][g2]))
        p1, p2 = A["geom_priority"][g1], A["geom_priority"][g2]
        assert rec[32, q, 3] == (0 if p1 == p2 else (1 if p1 > p2 else 2))
        np.testing.assert_array_equal([*fv[34, q, 2:], *fv[35, q, :3]], M.impedance_consts(A["pair_solimp"][q]))
    # (solimp travels as the impedance function consumes it: MuJoCo's clamps applied, the width as its reciprocal)
    np.testing.assert_array_equal(M.impedance_consts([0.9, 0.95, 0.001, 0.5, 2.0]), f32([0.9, 0.95, np.float32(1.0) / np.float32(0.001), 0.5, 2.0]))
    np.testing.assert_array_equal(M.impedance_consts([0.0, 1.5, 0.0, 2.0, 0.5]), f32([0.0001, 0.9999, np.float32(1.0) / np.float32(1e-15), 0.9999, 1.0]))
    # the rows' stiffness / damping (k, b) stand where solref stood: SURVEY B.10's formulas in double, to float32 rounding
    dt = float(A["opt_timestep"][0])
    def kb64(solref, solimp):
        tc, dr = max(float(solref[0]), 2 * dt), float(solref[1])
        dmax = min(max(float(solimp[1]), 1e-4), 0.9999)
        k = 1.0 / (dmax * dmax * tc * tc * dr * dr) if solref[0] > 0 else -float(solref[0]) / (dmax * dmax)      # (direct form: solref <= 0)
        b = 2.0 / (dmax * tc) if solref[1] > 0 else -float(solref[1]) / dmax
        return k, b
    for q in range(m.npair):
        np.testing.assert_allclose(fv[34, q, :2], kb64(A["pair_solref"][q], A["pair_solimp"][q]), rtol=1e-6)
    for l, j in enumerate(topo["limit_jnts"]):
        np.testing.assert_allclose(fv[27, l, 2:], kb64(A["jnt_solref"][j], A["jnt_solimp"][j]), rtol=1e-6)
    for l, i in enumerate(topo["fric_dofs"]):
        np.testing.assert_allclose(fv[23, l, 2:], kb64(A["dof_solref"][int(i)], A["dof_solimp"][int(i)]), rtol=1e-6)
    # lanes past a role's count are zero, and the device enum names as many quads as the host packs
    assert not rec[0:8, m.nbody:].any() and not rec[30:36, m.npair:].any()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "rsr_mjx_amd", "csrc", "rsr_device.hpp")).read()
    enum = re.search(r"enum LaneQuad \{(.*?)\};", src, re.S).group(1)
    names = [t.strip().split("=")[0].strip() for t in enum.replace("\n", " ").split(",") if t.strip()]
    assert names[-1] == "LQ_COUNT" and len(names) - 1 == LANE_QUADS


def test_geom_slots_and_pair_records(go2_model, cube_model):
    """model.geom_slots: the Go2 feet-only model keeps frames for the five geoms of its contact pairs only (floor + four feet of
    39), the Airbot models for every geom; the lane records' geom rows and pair rows go by slot."""
    from rsr_mjx_amd.model import geom_slots, lane_records, topology_tables
    sl = geom_slots(go2_model)
    A = go2_model.arrays
    assert len(sl) == 5 and sorted(set(A["pair_geom1"].tolist()) | set(A["pair_geom2"].tolist())) == sl.tolist()
    rec = lane_records(go2_model, topology_tables(go2_model), sl)
    fv = rec.view(np.float32)
    for k, g in enumerate(sl):
        assert rec[13, k, 0] == A["geom_bodyid"][g]
        np.testing.assert_array_equal(fv[13, k, 1:], A["geom_pos"][g].astype(np.float32))
        np.testing.assert_array_equal(fv[14, k], A["geom_quat"][g].astype(np.float32))
    assert not rec[13:15, len(sl):].any()                                    # lanes past the slots read zeros
    for q in range(go2_model.npair):
        assert sl[rec[30, q, 0]] == A["pair_geom1"][q] and sl[rec[30, q, 1]] == A["pair_geom2"][q]
        np.testing.assert_array_equal(fv[32, q, :3], A["geom_size"][A["pair_geom2"][q]].astype(np.float32))
    np.testing.assert_array_equal(geom_slots(cube_model), np.arange(cube_model.ngeom))
