"""Packing of a CompiledModel + env configuration into the flat "RSRM" blob.

The blob is the only thing that crosses the C ABI at model-creation time
(`rsr_model_create(const void* blob, size_t nbytes, ...)`, include/rsr_mjx.h):
a header, a table of named fields and the field data, float32 / int32 only.

  header   : char magic[4]="RSRM"; int32 version; int32 nfields; int32 total_bytes
  field[i] : char name[40]; int32 dtype (0=f32, 1=i32); int32 count; int32 offset_bytes; int32 reserved
  data     : each field 16-byte aligned

Both the HIP library and the CPU oracle look fields up by name, so neither
depends on the other's headers.
"""
from __future__ import annotations

import struct
from typing import Dict

import numpy as np

from .mjcf import CompiledModel

MAGIC = b"RSRM"
VERSION = 2            # 2: lane records carry solimp as impedance_consts() returns it (the kernels do not clamp it again)
_NAME_LEN = 40
_ENTRY = struct.Struct(f"<{_NAME_LEN}siiii")
_HEADER = struct.Struct("<4siii")

# model arrays that go to the device / oracle (everything numeric)
_SKIP_PREFIX = ("__",)


def pack_blob(fields: Dict[str, np.ndarray]) -> bytes:
    names = sorted(fields)
    table_bytes = _HEADER.size + _ENTRY.size * len(names)
    off = (table_bytes + 15) & ~15
    entries, chunks = [], []
    for n in names:
        a = np.ascontiguousarray(fields[n])
        if a.dtype.kind == "f":
            a = a.astype(np.float32)
            dt = 0
        elif a.dtype.kind in "iub":
            a = a.astype(np.int32)
            dt = 1
        else:
            raise TypeError(f"field {n}: unsupported dtype {a.dtype}")
        raw = a.tobytes()
        if len(n.encode()) >= _NAME_LEN:
            raise ValueError(f"field name too long: {n}")
        entries.append(_ENTRY.pack(n.encode(), dt, a.size, off, 0))
        pad = (-len(raw)) & 15
        chunks.append(raw + b"\0" * pad)
        off += len(raw) + pad
    head = _HEADER.pack(MAGIC, VERSION, len(names), off)
    body = head + b"".join(entries)
    body += b"\0" * (((table_bytes + 15) & ~15) - len(body))
    return body + b"".join(chunks)


def unpack_blob(blob: bytes) -> Dict[str, np.ndarray]:
    magic, ver, n, total = _HEADER.unpack_from(blob, 0)
    if magic != MAGIC or ver != VERSION:
        raise ValueError(f"not an RSRM v{VERSION} blob")
    out = {}
    for i in range(n):
        name, dt, cnt, off, _ = _ENTRY.unpack_from(blob, _HEADER.size + i * _ENTRY.size)
        name = name.rstrip(b"\0").decode()
        dtype = np.float32 if dt == 0 else np.int32
        out[name] = np.frombuffer(blob, dtype=dtype, count=cnt, offset=off).copy()
    return out


def model_fields(m: CompiledModel) -> Dict[str, np.ndarray]:
    f = {}
    for k, v in m.arrays.items():
        if k.startswith(_SKIP_PREFIX):
            continue
        f[k] = np.asarray(v)
    f["dims"] = np.array([m.nq, m.nv, m.nu, m.nbody, m.njnt, m.ngeom, m.nsite,
                          int(m.arrays["eq_obj1id"].shape[0]), m.npair], dtype=np.int32)
    topo = topology_tables(m)
    f.update(topo)
    f["geom_slot_ids"] = geom_slots(m)
    f["lane_rec"] = lane_records(m, topo, f["geom_slot_ids"])
    return f


def geom_slots(m: CompiledModel) -> np.ndarray:
    """Geom ids the kernel keeps world frames and friction for, one per "geom slot".  Every geom, in order -- unless fewer than
    half of the geoms appear in a contact pair (Go2 feet-only: floor or height field + four feet of 39 geoms), in which case
    only those, in geom order: the per-env LDS image then holds 5 frames instead of 39.  The kernel's Dims::NGA must match
    (checked at rsr_model_create)."""
    A = m.arrays
    used = sorted(set(int(g) for g in A["pair_geom1"]) | set(int(g) for g in A["pair_geom2"]))
    if 2 * len(used) < m.ngeom:
        return np.asarray(used, dtype=np.int32)
    return np.arange(m.ngeom, dtype=np.int32)


def flattened_tables(m: CompiledModel, body_dofmask: np.ndarray) -> Dict[str, np.ndarray]:
    """One-level look-up tables for the kernel's hot loops (a chain pair -> geom -> body -> mask of dependent global
    loads costs three L1/L2 round trips per trip of the loop; these tables make it one).

      pair_b1/b2, pair_root1/2, pair_mask1/2 : bodies, tree roots and chain dof masks of the two geoms of a pair
      pair_tw    : body_invweight0[b1,0] + body_invweight0[b2,0] ; pair_incl : margin - gap
      dof_rootid, dof_jtype, dof_k            : tree root of the dof's body, joint type, index of the dof inside its joint
      dof_act                                 : actuator driving the dof (-1: none; at most one per dof)
      dof_afl, dof_afrange                    : joint actuatorfrcrange, per dof
      body_jtype, body_qposadr, body_jpos, body_jaxis : the (single) joint of a body, -1 type when it has none
    """
    A = m.arrays
    nv, nb, npair = m.nv, m.nbody, m.npair
    gb = A["geom_bodyid"]
    b1, b2 = gb[A["pair_geom1"]], gb[A["pair_geom2"]]
    out = dict(
        pair_b1=b1.astype(np.int32), pair_b2=b2.astype(np.int32),
        pair_root1=A["body_rootid"][b1].astype(np.int32), pair_root2=A["body_rootid"][b2].astype(np.int32),
        pair_mask1=body_dofmask[b1].astype(np.uint32).view(np.int32), pair_mask2=body_dofmask[b2].astype(np.uint32).view(np.int32),
        pair_tw=(A["body_invweight0"][b1, 0] + A["body_invweight0"][b2, 0]).astype(np.float32),
        pair_incl=(A["pair_margin"] - A["pair_gap"]).astype(np.float32),
    )
    jid = A["dof_jntid"]
    out["dof_rootid"] = A["body_rootid"][A["dof_bodyid"]].astype(np.int32)
    out["dof_jtype"] = A["jnt_type"][jid].astype(np.int32)
    out["dof_k"] = (np.arange(nv) - A["jnt_dofadr"][jid]).astype(np.int32)
    act = np.full(nv, -1, dtype=np.int32)
    for u in range(m.nu):
        d = int(A["jnt_dofadr"][A["actuator_trnid"][u]])
        if act[d] >= 0:
            raise NotImplementedError("two actuators on one dof")
        act[d] = u
    out["dof_act"] = act
    out["dof_afl"] = A["jnt_actfrclimited"][jid].astype(np.int32)
    out["dof_afrange"] = A["jnt_actfrcrange"][jid].astype(np.float32).reshape(nv, 2)
    bj = np.full(nb, -1, dtype=np.int32)
    bq = np.zeros(nb, dtype=np.int32)
    bjp, bja = np.zeros((nb, 3), np.float32), np.zeros((nb, 3), np.float32)
    for b in range(nb):
        if A["body_jntnum"][b] > 0:
            j = int(A["body_jntadr"][b])
            bj[b], bq[b], bjp[b], bja[b] = A["jnt_type"][j], A["jnt_qposadr"][j], A["jnt_pos"][j], A["jnt_axis"][j]
    out.update(body_jtype=bj, body_qposadr=bq, body_jpos=bjp, body_jaxis=bja)
    return out


# quads (16-byte groups) of the per-lane records; keep in step with enum LaneQuad in csrc/rsr_device.hpp
LANE_QUADS = 41


def stiffness_damping(solref, solimp, timestep, disable_refsafe) -> tuple:
    """(k, b) of a constraint row's reference acceleration aref = -b v - k imp r, from its solref / solimp: they
    depend on model constants only, so the host computes them once -- in float32, operation by operation as the kernel did per
    row and per substep (two to four correctly rounded divisions each) -- and the lane records carry them in place of solref."""
    f = np.float32
    sr0, sr1 = f(solref[0]), f(solref[1])
    timeconst, dampratio = sr0, sr1
    if not disable_refsafe:
        timeconst = max(timeconst, f(f(2.0) * f(timestep)))
    dmax = min(max(f(solimp[1]), f(0.0001)), f(0.9999))
    k = f(1.0) / f(f(f(f(f(dmax * dmax) * timeconst) * timeconst) * dampratio) * dampratio)
    b = f(2.0) / f(dmax * timeconst)
    if sr0 <= 0:
        k = f(-sr0) / f(dmax * dmax)
    if sr1 <= 0:
        b = f(-sr1) / dmax
    return f(k), f(b)


def impedance_consts(solimp) -> np.ndarray:
    """solimp as the kernel's impedance function consumes it: (d0, d_width, 1 / width, midpoint, power) with MuJoCo's clamps applied
    (getimpedance: d in [mjMINIMP, mjMAXIMP], width >= mjMINVAL, power >= 1) -- model constants, so the host clamps and takes the
    reciprocal once, in float32, instead of every row of every substep."""
    f = np.float32
    clampi = lambda x: min(max(f(x), f(0.0001)), f(0.9999))
    width = max(f(solimp[2]), f(1e-15))
    return np.array([clampi(solimp[0]), clampi(solimp[1]), f(1.0) / width, clampi(solimp[3]), max(f(solimp[4]), f(1.0))], dtype=np.float32)


def lane_records(m: CompiledModel, topo: Dict[str, np.ndarray], geom_slot_ids=None) -> np.ndarray:
    """Per-lane constant records for the tree stages of the kernel: int32 [LANE_QUADS][64][4] (floats stored by bit pattern).

    In the kernel lane l plays body l, joint l, geom l, site l and dof l.  Fetching that lane's model constants from
    the per-field tables costs a dependent chain of scalar pointer load + vector load per field, a few hundred cycles
    each, in every substep.  Here everything a role needs is gathered (indirections resolved on the host) into quads
    laid out [quad][lane], so a stage issues all of its constant loads back to back, coalesced, and waits once.
    Lanes beyond a role's count hold zeros.  Layout (i = integer word):
      body : 0 (i parentid, i depth, i jtype, i qposadr)  1 (pos xyz, qpos0[qposadr])  2 quat  3 (jpos xyz, jaxis x)
             4 (jaxis yz, ipos xy)  5 iquat  6 (ipos z, i rootid, i submask, i dofmask)  7 (inertia xyz, 0)
      joint: 8 (i bodyid, i parent of that body, i type, 0)  9 body_quat[bodyid]  10 (body_pos[bodyid] xyz, jnt_pos x)
             11 (jnt_pos yz, jnt_axis xy)  12 (jnt_axis z, 0, 0, 0)
      geom : 13 (i bodyid, pos xyz)  14 quat  (indexed by geom SLOT, see geom_slots)      site : 15 (i bodyid, pos xyz)  16 quat
      dof  : 17 (i jntid, i bodyid, i jtype, i k)  18 (i rootid, i ancmask, i velmask, armature)
             19 (i actuator or -1, gear, i qposadr of the joint, i ctrllimited)  20 (ctrl lo, ctrl hi, gainprm0, biasprm0)
             21 (biasprm1, biasprm2, i forcelimited, force lo)  22 (force hi, i actfrclimited, actfrc lo, actfrc hi)
    and, indexed by the constraint-side roles (friction row l, limit slot l, geom pair l, equality l):
      (every `solref 0 1` below is stored as the row's stiffness / damping (k, b): stiffness_damping())
      fric : 23 (i dof, invweight0, solref 0 1)  24 (solimp 0..3)  25 (solimp 4, 0, 0, 0)      [solimp: as impedance_consts() returns it]
      limit: 26 (i qposadr, i dofadr, range lo hi)  27 (margin, invweight0, solref 0 1)  28 (solimp 0..3)  29 (solimp 4, i joint, 0, 0)
      pair : 30 (i slot of geom1, i slot of geom2, i kind, margin - gap)  31 (size[geom1] xyz, tw)  32 (size[geom2] xyz, i friction rule: 0 max,
             1 geom1's, 2 geom2's)  33 (i mask1, i mask2, i root1, i root2)  34 (solref 0 1, solimp 0 1)  35 (solimp 2 3 4, 0)
      eq   : 36 (i active, i qposadr1, i dofadr1, invweight0 sum)  37 (i has obj2, i qposadr2, i dofadr2, 0)  38 (data 0..3)
             39 (data 4, solref 0 1, solimp 0)  40 (solimp 1..4)
    The joint quad 12 also carries (jnt_axis z, i qposadr, i dofadr, 0) for the integrator.
    """
    A = m.arrays
    kb = lambda solref, solimp: stiffness_damping(solref, solimp, A["opt_timestep"][0], int(A["opt_disable_refsafe"][0]) != 0)
    rec = np.zeros((LANE_QUADS, 64, 4), dtype=np.int32)
    fv = rec.view(np.float32)
    if max(m.nbody, m.njnt, m.ngeom, m.nsite, m.nv) > 64:
        raise ValueError("lane records need at most 64 bodies / joints / geoms / sites / dofs")
    u32 = lambda x: np.uint32(int(x) & 0xFFFFFFFF).view(np.int32)
    for b in range(m.nbody):
        qa = int(topo["body_qposadr"][b])
        rec[0, b] = [A["body_parentid"][b], topo["body_depth"][b], topo["body_jtype"][b], qa]
        fv[1, b] = [*A["body_pos"][b], A["qpos0"][qa] if topo["body_jtype"][b] >= 0 else 0.0]
        fv[2, b] = A["body_quat"][b]
        fv[3, b] = [*topo["body_jpos"][b], topo["body_jaxis"][b][0]]
        fv[4, b] = [topo["body_jaxis"][b][1], topo["body_jaxis"][b][2], A["body_ipos"][b][0], A["body_ipos"][b][1]]
        fv[5, b] = A["body_iquat"][b]
        fv[6, b, 0] = A["body_ipos"][b][2]
        rec[6, b, 1:] = [A["body_rootid"][b], u32(topo["body_submask"][b]), u32(topo["body_dofmask"][b])]
        fv[7, b, :3] = A["body_inertia"][b]
    for j in range(m.njnt):
        jb = int(A["jnt_bodyid"][j])
        rec[8, j] = [jb, A["body_parentid"][jb], A["jnt_type"][j], 0]
        fv[9, j] = A["body_quat"][jb]
        fv[10, j] = [*A["body_pos"][jb], A["jnt_pos"][j][0]]
        fv[11, j] = [A["jnt_pos"][j][1], A["jnt_pos"][j][2], A["jnt_axis"][j][0], A["jnt_axis"][j][1]]
        fv[12, j, 0] = A["jnt_axis"][j][2]
        rec[12, j, 1:3] = [A["jnt_qposadr"][j], A["jnt_dofadr"][j]]
    slots = np.arange(m.ngeom) if geom_slot_ids is None else np.asarray(geom_slot_ids)
    slot_of = {int(g): k for k, g in enumerate(slots)}             # geom rows and the pair rows' geom references go by slot
    for k, g in enumerate(slots):
        rec[13, k, 0] = A["geom_bodyid"][g]
        fv[13, k, 1:] = A["geom_pos"][g]
        fv[14, k] = A["geom_quat"][g]
    for k in range(m.nsite):
        rec[15, k, 0] = A["site_bodyid"][k]
        fv[15, k, 1:] = A["site_pos"][k]
        fv[16, k] = A["site_quat"][k]
    for i in range(m.nv):
        jid = int(A["dof_jntid"][i])
        rec[17, i] = [jid, A["dof_bodyid"][i], topo["dof_jtype"][i], topo["dof_k"][i]]
        rec[18, i, :3] = [topo["dof_rootid"][i], u32(topo["dof_ancmask"][i]), u32(topo["dof_velmask"][i])]
        fv[18, i, 3] = A["dof_armature"][i]
        u = int(topo["dof_act"][i])
        rec[19, i, 0] = u
        rec[19, i, 2] = A["jnt_qposadr"][jid]
        if u >= 0:
            fv[19, i, 1] = A["actuator_gear"][u] if np.ndim(A["actuator_gear"][u]) == 0 else A["actuator_gear"][u][0]
            rec[19, i, 3] = A["actuator_ctrllimited"][u]
            fv[20, i] = [*A["actuator_ctrlrange"][u], A["actuator_gainprm"][u][0], A["actuator_biasprm"][u][0]]
            fv[21, i, :2] = A["actuator_biasprm"][u][1:3]
            rec[21, i, 2] = A["actuator_forcelimited"][u]
            fv[21, i, 3] = A["actuator_forcerange"][u][0]
            fv[22, i, 0] = A["actuator_forcerange"][u][1]
        rec[22, i, 1] = topo["dof_afl"][i]
        fv[22, i, 2:] = topo["dof_afrange"][i]
    if max(len(topo["fric_dofs"]), len(topo["limit_jnts"]), m.npair, int(A["eq_obj1id"].shape[0])) > 64:
        raise ValueError("lane records need at most 64 friction rows / limited joints / geom pairs / equalities")
    for l, i in enumerate(topo["fric_dofs"]):
        i = int(i)
        rec[23, l, 0] = i
        fv[23, l, 1:] = [A["dof_invweight0"][i], *kb(A["dof_solref"][i], A["dof_solimp"][i])]
        si = impedance_consts(A["dof_solimp"][i])
        fv[24, l] = si[:4]
        fv[25, l, 0] = si[4]
    for l, j in enumerate(topo["limit_jnts"]):
        j = int(j)
        rec[26, l, :2] = [A["jnt_qposadr"][j], A["jnt_dofadr"][j]]
        fv[26, l, 2:] = A["jnt_range"][j]
        fv[27, l] = [A["jnt_margin"][j], A["dof_invweight0"][int(A["jnt_dofadr"][j])], *kb(A["jnt_solref"][j], A["jnt_solimp"][j])]
        si = impedance_consts(A["jnt_solimp"][j])
        fv[28, l] = si[:4]
        fv[29, l, 0] = si[4]
        rec[29, l, 1] = j
    for q in range(m.npair):
        g1, g2 = int(A["pair_geom1"][q]), int(A["pair_geom2"][q])
        rec[30, q, :3] = [slot_of[g1], slot_of[g2], A["pair_kind"][q]]
        fv[30, q, 3] = topo["pair_incl"][q]
        fv[31, q] = [*A["geom_size"][g1], topo["pair_tw"][q]]
        fv[32, q, :3] = A["geom_size"][g2]
        p1, p2 = int(A["geom_priority"][g1]), int(A["geom_priority"][g2])
        rec[32, q, 3] = 0 if p1 == p2 else (1 if p1 > p2 else 2)
        rec[33, q] = [topo["pair_mask1"][q], topo["pair_mask2"][q], topo["pair_root1"][q], topo["pair_root2"][q]]
        si = impedance_consts(A["pair_solimp"][q])
        fv[34, q] = [*kb(A["pair_solref"][q], A["pair_solimp"][q]), *si[:2]]
        fv[35, q, :3] = si[2:5]
    for e in range(int(A["eq_obj1id"].shape[0])):
        j1, j2 = int(A["eq_obj1id"][e]), int(A["eq_obj2id"][e])
        d1 = int(A["jnt_dofadr"][j1])
        invw = float(np.float32(A["dof_invweight0"][d1]))
        rec[36, e, :3] = [A["eq_active0"][e], A["jnt_qposadr"][j1], d1]
        if j2 >= 0:
            d2 = int(A["jnt_dofadr"][j2])
            invw = float(np.float32(invw) + np.float32(A["dof_invweight0"][d2]))     # the kernel's fp32 sum
            rec[37, e, :3] = [1, A["jnt_qposadr"][j2], d2]
        fv[36, e, 3] = invw
        fv[38, e] = A["eq_data"][e][:4]
        si = impedance_consts(A["eq_solimp"][e])
        fv[39, e] = [A["eq_data"][e][4], *kb(A["eq_solref"][e], A["eq_solimp"][e]), si[0]]
        fv[40, e] = si[1:5]
    return rec


def topology_tables(m: CompiledModel) -> Dict[str, np.ndarray]:
    """Bitmask tables that let the kernel replace the tree recursions of the reference's
    scan.body_tree passes by independent per-lane loops (all models here have nv, nbody <= 32).

      dof_ancmask[i]  : dofs j that are ancestors of dof i, i included        (mass matrix, Jacobians)
      dof_velmask[i]  : dofs whose cdof*qvel make up the body velocity seen by cdof_dot[i]
                        (all strict ancestors for hinge/slide; for the rotational dofs of a free joint
                        only that joint's three translational dofs; empty for free translations)
      body_dofmask[b] : dofs on the chain from the root to body b, b's own included
      body_submask[b] : bodies in the subtree of b, b included
      body_depth[b]   : tree depth (world = 0); kinematics composes poses level by level
      fric_dofs       : dofs with frictionloss > 0 (constraint rows exist for them)
      limit_jnts      : limited hinge/slide joints
    """
    A = m.arrays
    nv, nb = m.nv, m.nbody
    if nv > 32 or nb > 32:
        raise ValueError("topology bitmasks need nv, nbody <= 32")
    anc = np.zeros(nv, dtype=np.int64)
    for i in range(nv):
        j = i
        while j >= 0:
            anc[i] |= 1 << j
            j = int(A["dof_parentid"][j])
    vel = np.zeros(nv, dtype=np.int64)
    for i in range(nv):
        jid = int(A["dof_jntid"][i])
        da = int(A["jnt_dofadr"][jid])
        if A["jnt_type"][jid] == 0:           # free
            vel[i] = 0 if i < da + 3 else (0b111 << da)
        else:
            vel[i] = anc[i] & ~(1 << i)
    bdof = np.zeros(nb, dtype=np.int64)
    for b in range(1, nb):
        k = b
        while k > 0 and A["body_dofnum"][k] == 0:
            k = int(A["body_parentid"][k])
        if k > 0:
            last = int(A["body_dofadr"][k] + A["body_dofnum"][k] - 1)
            bdof[b] = anc[last]
    sub = np.zeros(nb, dtype=np.int64)
    for b in range(nb - 1, -1, -1):
        sub[b] |= 1 << b
        if b > 0:
            sub[int(A["body_parentid"][b])] |= sub[b]
    fric = np.array([i for i in range(nv) if A["dof_frictionloss"][i] > 0], dtype=np.int32)
    lim = np.array([j for j in range(m.njnt) if A["jnt_limited"][j] and A["jnt_type"][j] in (2, 3)], dtype=np.int32)
    depth = np.zeros(nb, dtype=np.int32)
    for b in range(1, nb):
        depth[b] = depth[int(A["body_parentid"][b])] + 1
    as_i32 = lambda a: a.astype(np.uint32).view(np.int32)
    flat = flattened_tables(m, bdof)
    return dict(**flat, dof_ancmask=as_i32(anc), dof_velmask=as_i32(vel), body_dofmask=as_i32(bdof),
                body_submask=as_i32(sub), fric_dofs=fric, limit_jnts=lim, body_depth=depth,
                counts2=np.array([len(fric), len(lim), int(depth.max()), int(A["body_jntnum"].max()),
                                  # longest bit lists the kernel walks: subtree bodies (bodies >= 1), chain dofs
                                  max([bin(int(x)).count("1") for x in sub[1:]] + [0]),
                                  max([bin(int(x)).count("1") for x in list(bdof) + list(anc) + list(vel)] + [0])], dtype=np.int32))
