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


def _hfield_sphere(lib64, hsize, data, spos, radius, hpos=(0, 0, 0), hmat=np.eye(3)):
    nrow, ncol = data.shape
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (hpos, hmat, spos, [radius])]
    hs = np.ascontiguousarray(hsize, dtype=np.float32)
    dat = np.ascontiguousarray(data, dtype=np.float32)
    out, nrm = np.zeros(4), np.zeros(3)
    n = lib64.oracle_hfield_sphere(a[0].ctypes.data, a[1].ctypes.data, hs.ctypes.data, nrow, ncol, dat.ctypes.data, a[2].ctypes.data,
                                   a[3].ctypes.data, out.ctypes.data, nrm.ctypes.data)
    return n, out[0], out[1:4].copy(), nrm.copy()


def _surface_samples(hsize, data, cx, cy, span, k=24):
    """dense point samples of the triangulated surface (diagonal (c,r)-(c+1,r+1)) around (cx, cy)"""
    nrow, ncol = data.shape
    sx, sy, sz = hsize[:3]
    dx, dy = 2 * sx / (ncol - 1), 2 * sy / (nrow - 1)
    pts = []
    c0, r0 = int(np.floor((cx + sx) / dx)), int(np.floor((cy + sy) / dy))
    u = np.linspace(0, 1, k)
    U, V = np.meshgrid(u, u)
    for r in range(max(r0 - span, 0), min(r0 + span + 1, nrow - 1)):
        for c in range(max(c0 - span, 0), min(c0 + span + 1, ncol - 1)):
            z00, z10, z01, z11 = (data[r, c] * sz, data[r, c + 1] * sz, data[r + 1, c] * sz, data[r + 1, c + 1] * sz)
            Z = np.where(U >= V, z00 + (z10 - z00) * U + (z11 - z10) * V, z00 + (z11 - z01) * U + (z01 - z00) * V)
            pts.append(np.stack([-sx + dx * (c + U), -sy + dy * (r + V), Z], -1).reshape(-1, 3))
    return np.concatenate(pts)


def test_hfield_sphere_closest_surface_point(lib64):
    """Sphere vs height field: one contact at the closest point of the triangulated surface (brute-force samples);
    flat field reduces to the plane-sphere closed form; centre below the surface gives the perpendicular depth."""
    rng = np.random.default_rng(0)
    hsize = np.array([1.0, 1.5, 0.3, 0.1])
    data = rng.uniform(0, 1, size=(21, 17))
    for _ in range(60):
        c = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-1.2, 1.2), rng.uniform(0.0, 0.45)])
        r = 0.04
        n, dist, pos, nrm = _hfield_sphere(lib64, hsize, data, c, r)
        assert n == 1 and abs(np.linalg.norm(nrm) - 1) < 1e-9
        S = _surface_samples(hsize, data, c[0], c[1], 2)
        dmin = np.linalg.norm(S - c, axis=1).min()
        # height of the surface under the centre, from the same samples
        near = S[np.argmin(np.linalg.norm(S[:, :2] - c[:2], axis=1))]
        if c[2] > near[2] + 0.01:
            assert abs(dist - (dmin - r)) < 4e-3, (dist, dmin - r)
            q = pos - nrm * (0.5 * dist)                       # the surface point
            assert np.linalg.norm(S - q, axis=1).min() < 6e-3
            np.testing.assert_allclose(c - nrm * (dist + r), q, atol=1e-9)
        elif c[2] < near[2] - 0.01:
            assert dist < -r and nrm[2] > 0
    flat = np.full((9, 9), 0.5)
    n, dist, pos, nrm = _hfield_sphere(lib64, hsize, flat, [0.13, -0.2, 0.25], 0.05)
    np.testing.assert_allclose([dist, *nrm], [0.25 - 0.15 - 0.05, 0, 0, 1], atol=1e-7)      # size is stored as float32
    np.testing.assert_allclose(pos, [0.13, -0.2, 0.15 + 0.5 * dist], atol=1e-7)
    n, dist, pos, nrm = _hfield_sphere(lib64, hsize, flat, [0.13, -0.2, 0.10], 0.05)      # centre under the surface
    np.testing.assert_allclose([dist, *nrm], [-0.05 - 0.05, 0, 0, 1], atol=1e-7)
    assert _hfield_sphere(lib64, hsize, flat, [1.2, 0, 0.2], 0.05)[0] == 0               # beyond the field: no contact
    # a rotated, shifted field gives the rotated, shifted answer
    ang = 0.4
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    c = np.array([0.21, 0.33, 0.31])
    n0, d0, p0, n0v = _hfield_sphere(lib64, hsize, data, c, 0.04)
    off = np.array([0.5, -0.25, 0.125])
    n1, d1, p1, n1v = _hfield_sphere(lib64, hsize, data, off + Rz @ c, 0.04, hpos=off, hmat=Rz)
    np.testing.assert_allclose([d1, *p1, *n1v], [d0, *(off + Rz @ p0), *(Rz @ n0v)], atol=1e-12)


# ---- plane-capsule / plane-cylinder (the Go2 full-collision model of the Handstand task: go2_mjx.xml) ----
def _surface_samples_capsule(rng, pos, R, radius, halflen, n=60000):
    """points on a capsule's surface: the cylinder wall and the two end caps"""
    t = rng.uniform(-halflen, halflen, n); a = rng.uniform(0, 2 * np.pi, n)
    wall = np.stack([radius * np.cos(a), radius * np.sin(a), t], 1)
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    caps = v * radius + np.where(v[:, 2:3] > 0, 1.0, -1.0) * np.array([0, 0, halflen])
    return np.concatenate([wall, caps]) @ R.T + pos


def _surface_samples_cylinder(rng, pos, R, radius, halflen, n=60000):
    t = rng.uniform(-halflen, halflen, n); a = rng.uniform(0, 2 * np.pi, n)
    wall = np.stack([radius * np.cos(a), radius * np.sin(a), t], 1)
    rim = np.stack([radius * np.cos(a), radius * np.sin(a), np.where(rng.uniform(size=n) > 0.5, halflen, -halflen)], 1)
    return np.concatenate([wall, rim]) @ R.T + pos


def test_plane_capsule_against_surface_samples(lib64):
    """Two contacts, the end spheres of the segment: each one's depth is its sphere's, the deeper one is the deepest point of the
    whole capsule surface, the points lie mid-way between sphere and plane, and the frame's first tangent is the capsule axis projected
    into the plane."""
    rng = np.random.default_rng(5)
    for _ in range(60):
        Rp, Rc = _rand_rot(rng, 0.3), _rand_rot(rng)
        pp, cp = rng.normal(size=3) * 0.1, rng.normal(size=3) * 0.1
        radius, halflen = rng.uniform(0.01, 0.05), rng.uniform(0.03, 0.15)
        out, nrm, frame = np.zeros(8), np.zeros(3), np.zeros(9)
        n = lib64.oracle_plane_capsule(pp.ctypes.data, np.ascontiguousarray(Rp).ctypes.data, cp.ctypes.data, np.ascontiguousarray(Rc).ctypes.data,
                                       radius, halflen, out.ctypes.data, nrm.ctypes.data, frame.ctypes.data)
        assert n == 2
        nn = Rp[:, 2]
        np.testing.assert_allclose(nrm, nn, atol=1e-12)
        pts = _surface_samples_capsule(rng, cp, Rc, radius, halflen)
        h = (pts - pp) @ nn                                   # signed height of the surface samples above the plane
        axis = Rc[:, 2]
        for i, sg in enumerate((1.0, -1.0)):                  # +axis end first
            end = cp + sg * axis * halflen
            dist, pos = out[4 * i], out[4 * i + 1:4 * i + 4]
            assert abs(dist - ((end - pp) @ nn - radius)) < 1e-12
            np.testing.assert_allclose(pos, end - nn * (radius + 0.5 * dist), atol=1e-12)
        assert abs(min(out[0], out[4]) - h.min()) < 2e-3 * radius + 1e-6
        f = frame.reshape(3, 3)
        np.testing.assert_allclose(f[0], nn, atol=1e-12)
        b = axis - nn * (nn @ axis)
        if np.linalg.norm(b) >= 0.5:
            np.testing.assert_allclose(f[1], b / np.linalg.norm(b), atol=1e-12)
            np.testing.assert_allclose(f @ f.T, np.eye(3), atol=1e-12)
        np.testing.assert_allclose(f[2], np.cross(f[0], f[1]), atol=1e-12)


def test_plane_cylinder_against_surface_samples(lib64):
    """Three contacts on the rim of the disk facing the plane (or on both disks when the cylinder lies flat): the first is the deepest
    point of the whole surface, every contact point projects onto a point of the cylinder's surface (pos + n * dist / 2 is ON the
    surface), and its dist is that surface point's height."""
    rng = np.random.default_rng(6)
    flat_seen = 0
    for trial in range(80):
        Rp = _rand_rot(rng, 0.3)
        Rc = _rand_rot(rng)
        if trial % 4 == 0:                                     # lying flat: axis in the plane
            nn = Rp[:, 2]
            ax = np.cross(nn, rng.normal(size=3)); ax /= np.linalg.norm(ax)
            x = np.cross(ax, nn)
            Rc = np.stack([x, np.cross(ax, x), ax], 1)
            flat_seen += 1
        pp, cp = rng.normal(size=3) * 0.1, rng.normal(size=3) * 0.1
        radius, halflen = rng.uniform(0.02, 0.06), rng.uniform(0.01, 0.13)
        out, nrm = np.zeros(12), np.zeros(3)
        n = lib64.oracle_plane_cylinder(pp.ctypes.data, np.ascontiguousarray(Rp).ctypes.data, cp.ctypes.data, np.ascontiguousarray(Rc).ctypes.data,
                                        radius, halflen, out.ctypes.data, nrm.ctypes.data)
        assert n == 3
        nn = Rp[:, 2]
        pts = _surface_samples_cylinder(rng, cp, Rc, radius, halflen, n=150000)
        h = (pts - pp) @ nn
        assert abs(out[0] - h.min()) < 3e-3 * radius + 1e-6, (trial, out[0], h.min())
        for i in range(3):
            dist, pos = out[4 * i], out[4 * i + 1:4 * i + 4]
            surf = pos + nn * dist * 0.5                       # the surface point the contact stands for
            loc = Rc.T @ (surf - cp)
            assert abs(np.hypot(loc[0], loc[1]) - radius) < 1e-9 * 1e3 or np.hypot(loc[0], loc[1]) <= radius + 1e-9, (trial, i)
            assert abs(abs(loc[2]) - halflen) < 1e-9                                    # on a disk's plane
            assert abs((surf - pp) @ nn - dist) < 1e-9                                   # dist = that point's height
        # the two side points are 120 degrees from the deepest one on the same rim (not in the flat case's second point)
        l0, l2 = Rc.T @ (out[1:4] + nn * out[0] * 0.5 - cp), Rc.T @ (out[9:12] + nn * out[8] * 0.5 - cp)
        c = (l0[:2] @ l2[:2]) / (np.linalg.norm(l0[:2]) * np.linalg.norm(l2[:2]))
        assert abs(c + 0.5) < 1e-6, (trial, c)
    assert flat_seen >= 10
