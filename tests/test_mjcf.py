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
