"""Box-box / plane-box narrow phase of the oracle against brute force."""
import ctypes as C

import numpy as np
import pytest

from rsr_mjx_amd import mjcf


def _call_box_box(lib, pa, Ra, sa, pb, Rb, sb):
    arrs = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in (pa, Ra, sa, pb, Rb, sb)]
    out, nrm = np.zeros(16), np.zeros(3)
    n = lib.oracle_box_box(*[a.ctypes.data for a in arrs], out.ctypes.data, nrm.ctypes.data)
    return n, out.reshape(4, 4)[:n], nrm


def _rand_rot(rng, scale=1.0):
    q = np.array([1.0, 0, 0, 0]) + scale * rng.normal(size=4)
    return mjcf.quat_to_mat(q / np.linalg.norm(q))


def _inside(p, pos, R, size, tol):
    loc = R.T @ (p - pos)
    return np.all(np.abs(loc) <= size + tol)


@pytest.fixture(scope="module")
def lib64(oracle_mod):
    return oracle_mod._load("f64")


def test_separated_boxes_give_no_contact(lib64):
    rng = np.random.default_rng(0)
    for _ in range(200):
        Ra, Rb = _rand_rot(rng), _rand_rot(rng)
        sa, sb = rng.uniform(0.02, 0.2, 3), rng.uniform(0.02, 0.2, 3)
        d = rng.normal(size=3)
        d = d / np.linalg.norm(d) * (np.linalg.norm(sa) + np.linalg.norm(sb) + 0.01)
        n, _, _ = _call_box_box(lib64, np.zeros(3), Ra, sa, d, Rb, sb)
        assert n == 0


def test_stacked_boxes_four_point_manifold(lib64):
    # small box resting 1 mm inside a big slab: 4 corner contacts, normal +z, depth 1 mm, points mid-way
    sa, sb = np.array([0.8, 0.3, 0.01]), np.array([0.04, 0.04, 0.04])
    pb = np.array([0.1, 0.05, 0.01 + 0.04 - 0.001])
    n, pts, nrm = _call_box_box(lib64, np.zeros(3), np.eye(3), sa, pb, np.eye(3), sb)
    assert n == 4
    np.testing.assert_allclose(nrm, [0, 0, 1], atol=1e-12)
    np.testing.assert_allclose(pts[:, 0], -0.001, atol=1e-12)
    np.testing.assert_allclose(pts[:, 3], 0.01 - 0.0005, atol=1e-12)
    corners = {(round(x, 6), round(y, 6)) for x, y in pts[:, 1:3]}
    assert corners == {(0.14, 0.09), (0.06, 0.09), (0.14, 0.01), (0.06, 0.01)}


def test_penetrating_boxes_contacts_are_consistent(lib64):
    """Random overlapping boxes: every contact point lies (within its depth) inside both boxes, the normal is
    unit and points from A to B, and translating B by depth along the normal separates that point."""
    rng = np.random.default_rng(1)
    hits = 0
    for _ in range(400):
        Ra, Rb = _rand_rot(rng, 0.4), _rand_rot(rng, 0.4)
        sa, sb = rng.uniform(0.03, 0.15, 3), rng.uniform(0.03, 0.15, 3)
        pb = rng.normal(size=3) * 0.08
        n, pts, nrm = _call_box_box(lib64, np.zeros(3), Ra, sa, pb, Rb, sb)
        if n == 0:
            continue
        hits += 1
        assert abs(np.linalg.norm(nrm) - 1) < 1e-9
        assert nrm @ pb > -1e-9
        for dist, *p in pts:
            p = np.array(p)
            assert dist < 0
            assert _inside(p, np.zeros(3), Ra, sa, -dist + 1e-9)
            assert _inside(p, pb, Rb, sb, -dist + 1e-9)
    assert hits > 100


def test_sat_depth_matches_sampled_support(lib64):
    """Face-contact depth equals the SAT penetration along the reported normal (brute-force support functions)."""
    rng = np.random.default_rng(2)
    checked = 0
    for _ in range(300):
        # B is small and sits well inside A's top face, so its deepest vertex survives the clipping
        Ra, Rb = _rand_rot(rng, 0.02), _rand_rot(rng, 0.02)
        sa, sb = rng.uniform(0.2, 0.3, 3), rng.uniform(0.03, 0.06, 3)
        pb = Ra @ np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), sa[2] + sb[2] - rng.uniform(0.001, 0.004)])
        n, pts, nrm = _call_box_box(lib64, np.zeros(3), Ra, sa, pb, Rb, sb)
        if n == 0:
            continue
        corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=float)
        va = (corners * sa) @ Ra.T
        vb = (corners * sb) @ Rb.T + pb
        overlap = (va @ nrm).max() - (vb @ nrm).min()      # penetration along the normal
        assert overlap > 0
        assert -pts[:, 0].min() <= overlap + 1e-9
        if n >= 3:
            assert abs(-pts[:, 0].min() - overlap) < 1e-6
            checked += 1
    assert checked > 50


def test_plane_box_deepest_vertices(lib64):
    rng = np.random.default_rng(3)
    for _ in range(100):
        Rb = _rand_rot(rng, 0.3)
        size = rng.uniform(0.02, 0.1, 3)
        bp = np.array([0, 0, rng.uniform(0.0, 0.1)])
        args = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in (np.zeros(3), np.eye(3), bp, Rb, size)]
        out, nrm = np.zeros(16), np.zeros(3)
        n = lib64.oracle_plane_box(*[a.ctypes.data for a in args], out.ctypes.data, nrm.ctypes.data)
        corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=float)
        z = ((corners * size) @ Rb.T + bp)[:, 2]
        if z.min() >= 0:
            assert n == 0
            continue
        pts = out.reshape(4, 4)[:n]
        assert n >= 1 and abs(pts[:, 0].min() - z.min()) < 1e-12
        assert np.all(pts[:, 0] < 0) and np.all(pts[:, 0] <= z.min() + 1e-3 + 1e-12)
