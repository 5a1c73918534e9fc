"""MJCF-subset model compiler (host side, numpy fp64).

The reference builds its models with MuJoCo's C compiler, once, at env
construction (`mujoco.MjModel.from_xml_path`, reference
ppo_train/airbot_training/cube_env.py:37-38, go2/base.py:25-27) and hands the
result to MJX.  Neither MuJoCo nor MJX exists on the target, so this module
restates the part of the MJCF compilation the hot-path models need
(SURVEY.md Appendix A.4): defaults classes, body/joint/geom/site/actuator/
equality/exclude elements, explicit or geom-derived inertials, euler->quat,
the static collision pair list with MuJoCo's contact-parameter mixing, and
the compile-time constants `dof_invweight0`, `body_invweight0`,
`stat.meaninertia` (M^-1 at qpos0).

Field names follow MuJoCo's mjModel so the constants can be read side by side
with the MuJoCo documentation.  Everything is computed in float64 and cast to
float32 only when packed for the device (model.py).
"""
from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

# ---- enums (values follow MuJoCo's mjtJoint / mjtGeom / mjtIntegrator) ----
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
INT_EULER, INT_RK4, INT_IMPLICIT, INT_IMPLICITFAST = 0, 1, 2, 3
# pair kinds understood by the stepper
PAIR_PLANE_BOX, PAIR_BOX_BOX, PAIR_PLANE_SPHERE, PAIR_HFIELD_SPHERE, PAIR_PLANE_CAPSULE, PAIR_PLANE_CYLINDER = 0, 1, 2, 3, 4, 5

MJ_MINVAL = 1e-15

_GEOM_TYPES = {
    "plane": GEOM_PLANE, "hfield": GEOM_HFIELD, "sphere": GEOM_SPHERE,
    "capsule": GEOM_CAPSULE, "ellipsoid": GEOM_ELLIPSOID,
    "cylinder": GEOM_CYLINDER, "box": GEOM_BOX, "mesh": GEOM_MESH,
}
_INTEGRATORS = {"euler": INT_EULER, "rk4": INT_RK4, "implicit": INT_IMPLICIT,
                "implicitfast": INT_IMPLICITFAST}


# --------------------------------------------------------------------------
# small math helpers (float64)
# --------------------------------------------------------------------------
def _vec(s: Optional[str], n: Optional[int] = None, default=None) -> np.ndarray:
    if s is None:
        return None if default is None else np.array(default, dtype=np.float64)
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None and v.size != n:
        raise ValueError(f"expected {n} numbers, got {v.size}: {s!r}")
    return v


def quat_mul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
    ])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def mat_to_quat(m):
    """Rotation matrix -> unit quaternion (w,x,y,z), w >= 0 branch preferred."""
    t = np.trace(m)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = np.array([(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s])
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = np.array([(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s])
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = np.array([(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s])
    return q / np.linalg.norm(q)


def rotate(v, q):
    return quat_to_mat(q) @ v


def euler_to_quat(e, seq="xyz"):
    """MuJoCo convention: lower-case axes are intrinsic (post-multiply)."""
    q = np.array([1.0, 0, 0, 0])
    for i in range(3):
        r = np.array([math.cos(e[i] / 2), 0.0, 0.0, 0.0])
        sa = math.sin(e[i] / 2)
        ax = seq[i].lower()
        r[1 + "xyz".index(ax)] = sa
        q = quat_mul(q, r) if seq[i].islower() else quat_mul(r, q)
    return q


def axisangle_to_quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if n < MJ_MINVAL:
        return np.array([1.0, 0, 0, 0])
    s = math.sin(angle / 2)
    return np.concatenate([[math.cos(angle / 2)], axis / n * s])


# --------------------------------------------------------------------------
# compiled model container
# --------------------------------------------------------------------------
@dataclass
class CompiledModel:
    """Flat model constants.  Arrays are float64/int32 numpy arrays."""
    name: str = ""
    arrays: Dict[str, np.ndarray] = field(default_factory=dict)
    names: Dict[str, Dict[str, int]] = field(default_factory=dict)

    def __getattr__(self, k):
        arrays = self.__dict__.get("arrays", {})
        if k in arrays:
            return arrays[k]
        raise AttributeError(k)

    # sizes
    @property
    def nq(self): return int(self.arrays["qpos0"].shape[0])
    @property
    def nv(self): return int(self.arrays["dof_bodyid"].shape[0])
    @property
    def nu(self): return int(self.arrays["actuator_trnid"].shape[0])
    @property
    def nbody(self): return int(self.arrays["body_parentid"].shape[0])
    @property
    def njnt(self): return int(self.arrays["jnt_type"].shape[0])
    @property
    def ngeom(self): return int(self.arrays["geom_type"].shape[0])
    @property
    def nsite(self): return int(self.arrays["site_bodyid"].shape[0])
    @property
    def npair(self): return int(self.arrays["pair_geom1"].shape[0])

    def id(self, kind: str, name: str) -> int:
        return self.names[kind][name]

    # ---- serialisation (npz, no pickle) ----
    def save(self, path: str) -> None:
        out = dict(self.arrays)
        for kind, d in self.names.items():
            keys = sorted(d, key=lambda k: d[k])
            out[f"__names_{kind}"] = np.array(keys, dtype="U64")
            out[f"__ids_{kind}"] = np.array([d[k] for k in keys], dtype=np.int32)
        out["__model_name"] = np.array([self.name], dtype="U64")
        np.savez_compressed(path, **out)

    @staticmethod
    def load(path: str) -> "CompiledModel":
        z = np.load(path, allow_pickle=False)
        m = CompiledModel()
        for k in z.files:
            if k.startswith("__names_"):
                kind = k[len("__names_"):]
                m.names[kind] = {str(n): int(i) for n, i in zip(z[k], z[f"__ids_{kind}"])}
            elif k.startswith("__ids_"):
                continue
            elif k == "__model_name":
                m.name = str(z[k][0])
            else:
                m.arrays[k] = z[k]
        return m


# --------------------------------------------------------------------------
# defaults handling
# --------------------------------------------------------------------------
_DEFAULT_ELEMS = ("joint", "geom", "site", "position", "motor", "general", "velocity", "mesh", "material",
                  "camera", "light", "pair", "equality", "tendon")


class _Defaults:
    def __init__(self):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {"main": {e: {} for e in _DEFAULT_ELEMS}}

    def parse(self, node: ET.Element, parent: Optional[str] = None):
        cname = node.get("class")
        if parent is None:
            cname = cname or "main"
        if cname is None:
            raise ValueError("nested <default> needs a class")
        if cname == "main":
            cur = self.classes["main"]
        else:
            base = self.classes[parent or "main"]
            cur = {e: dict(base.get(e, {})) for e in _DEFAULT_ELEMS}
            self.classes[cname] = cur
        for ch in node:
            if ch.tag != "default" and ch.tag in cur:
                cur[ch.tag].update(ch.attrib)
        for ch in node:
            if ch.tag == "default":
                self.parse(ch, parent=cname)

    def resolve(self, elem: str, attrib: Dict[str, str], childclass: Optional[str]) -> Dict[str, str]:
        cname = attrib.get("class") or childclass or "main"
        if cname not in self.classes:
            raise ValueError(f"unknown default class {cname!r}")
        out = dict(self.classes[cname].get(elem, {}))
        out.update(attrib)
        return out


# --------------------------------------------------------------------------
# the compiler
# --------------------------------------------------------------------------
class _Body:
    def __init__(self):
        self.name = ""
        self.parent = -1
        self.pos = np.zeros(3)
        self.quat = np.array([1.0, 0, 0, 0])
        self.joints: List[dict] = []
        self.geoms: List[dict] = []
        self.sites: List[dict] = []
        self.inertial: Optional[dict] = None
        self.gravcomp = 0.0


class MjcfCompiler:
    def __init__(self, path: str):
        self.path = path
        self.dir = os.path.dirname(os.path.abspath(path))
        self.root = self._load(path)
        self.defaults = _Defaults()
        self.angle_deg = True
        self.eulerseq = "xyz"
        self.autolimits = True
        self.inertiafromgeom = "auto"
        self.inertiagrouprange = (0, 5)
        self.boundmass = 0.0
        self.boundinertia = 0.0
        self.settotalmass = -1.0
        self.meshdir = ""
        self.opt = dict(timestep=0.002, gravity=np.array([0, 0, -9.81]), iterations=100, ls_iterations=50,
                        tolerance=1e-8, ls_tolerance=0.01, impratio=1.0, integrator=INT_EULER,
                        cone="pyramidal", solver="newton", disable_eulerdamp=False, disable_refsafe=False,
                        disable_warmstart=False, disable_filterparent=False, disable_frictionloss=False,
                        disable_limit=False, disable_contact=False, disable_equality=False,
                        disable_gravity=False, disable_clampctrl=False, disable_actuation=False,
                        disable_passive=False)
        self.bodies: List[_Body] = []
        self.hfields: Dict[str, dict] = {}

    # ---- XML loading with <include> ----
    def _load(self, path: str) -> ET.Element:
        root = ET.parse(path).getroot()
        self._expand_includes(root, os.path.dirname(os.path.abspath(path)))
        return root

    def _expand_includes(self, node: ET.Element, base: str):
        i = 0
        children = list(node)
        for ch in children:
            if ch.tag == "include":
                inc = ET.parse(os.path.join(base, ch.get("file"))).getroot()
                self._expand_includes(inc, base)
                idx = list(node).index(ch)
                node.remove(ch)
                for k, sub in enumerate(list(inc)):
                    node.insert(idx + k, sub)
            else:
                self._expand_includes(ch, base)
            i += 1

    # ---- orientation of an element ----
    def _orient(self, a: Dict[str, str]) -> np.ndarray:
        if "quat" in a:
            q = _vec(a["quat"], 4)
            return q / np.linalg.norm(q)
        if "euler" in a:
            e = _vec(a["euler"], 3)
            if self.angle_deg:
                e = np.deg2rad(e)
            return euler_to_quat(e, self.eulerseq)
        if "axisangle" in a:
            v = _vec(a["axisangle"], 4)
            ang = math.radians(v[3]) if self.angle_deg else v[3]
            return axisangle_to_quat(v[:3], ang)
        if "zaxis" in a:
            z = _vec(a["zaxis"], 3)
            z = z / np.linalg.norm(z)
            a0 = np.array([0, 0, 1.0])
            axis = np.cross(a0, z)
            s = np.linalg.norm(axis)
            ang = math.atan2(s, float(a0 @ z))
            if s < 1e-10:
                axis = np.array([1.0, 0, 0])
            return axisangle_to_quat(axis, ang)
        if "xyaxes" in a:
            v = _vec(a["xyaxes"], 6)
            x = v[:3] / np.linalg.norm(v[:3])
            y = v[3:] - x * (x @ v[3:])
            y = y / np.linalg.norm(y)
            z = np.cross(x, y)
            return mat_to_quat(np.stack([x, y, z], axis=1))
        return np.array([1.0, 0, 0, 0])

    # ---- top-level sections ----
    def _parse_compiler_option(self):
        for c in self.root.iter("compiler"):
            if "angle" in c.attrib:
                self.angle_deg = c.get("angle") == "degree"
            self.eulerseq = c.get("eulerseq", self.eulerseq)
            if "autolimits" in c.attrib:
                self.autolimits = c.get("autolimits") == "true"
            self.inertiafromgeom = c.get("inertiafromgeom", self.inertiafromgeom)
            if "inertiagrouprange" in c.attrib:
                r = _vec(c.get("inertiagrouprange"), 2)
                self.inertiagrouprange = (int(r[0]), int(r[1]))
            self.boundmass = float(c.get("boundmass", self.boundmass))
            self.boundinertia = float(c.get("boundinertia", self.boundinertia))
            self.meshdir = c.get("meshdir", self.meshdir)
        for o in self.root.iter("option"):
            a = o.attrib
            for k in ("timestep", "tolerance", "ls_tolerance", "impratio"):
                if k in a:
                    self.opt[k] = float(a[k])
            for k in ("iterations", "ls_iterations"):
                if k in a:
                    self.opt[k] = int(a[k])
            if "gravity" in a:
                self.opt["gravity"] = _vec(a["gravity"], 3)
            if "integrator" in a:
                self.opt["integrator"] = _INTEGRATORS[a["integrator"].lower()]
            if "cone" in a:
                self.opt["cone"] = a["cone"].lower()
            if "solver" in a:
                self.opt["solver"] = a["solver"].lower()
            for f in o.iter("flag"):
                for k, v in f.attrib.items():
                    key = f"disable_{k}"
                    if key in self.opt:
                        self.opt[key] = (v == "disable")
        if self.opt["cone"] != "pyramidal":
            raise NotImplementedError("only pyramidal friction cones are supported")
        if self.opt["solver"] != "newton":
            raise NotImplementedError("only the Newton solver is supported")

    def _parse_assets(self):
        for asset in self.root.iter("asset"):
            for h in asset.iter("hfield"):
                self.hfields[h.get("name")] = dict(h.attrib)

    def _parse_body(self, node: ET.Element, parent: int, childclass: Optional[str]):
        b = _Body()
        b.name = node.get("name", f"body{len(self.bodies)}")
        b.parent = parent
        b.pos = _vec(node.get("pos"), 3, default=[0, 0, 0])
        b.quat = self._orient(node.attrib)
        b.gravcomp = float(node.get("gravcomp", 0))
        childclass = node.get("childclass", childclass)
        bid = len(self.bodies)
        self.bodies.append(b)
        self._parse_body_children(node, bid, childclass)
        return bid

    def _parse_body_children(self, node: ET.Element, bid: int, childclass: Optional[str]):
        b = self.bodies[bid]
        for ch in node:
            if ch.tag == "inertial":
                b.inertial = dict(ch.attrib)
            elif ch.tag in ("joint", "freejoint"):
                if ch.tag == "freejoint":
                    a = dict(ch.attrib)
                    a["type"] = "free"
                    # freejoint does not take defaults
                    j = a
                else:
                    j = self.defaults.resolve("joint", ch.attrib, childclass)
                b.joints.append(j)
            elif ch.tag == "geom":
                b.geoms.append(self.defaults.resolve("geom", ch.attrib, childclass))
            elif ch.tag == "site":
                b.sites.append(self.defaults.resolve("site", ch.attrib, childclass))
        for ch in node:
            if ch.tag == "body":
                self._parse_body(ch, bid, childclass)

    # ---- inertia from geoms ----
    def _geom_mass_inertia(self, g: dict):
        """Mass and inertia (about geom centre, geom frame) of a primitive geom."""
        gtype = _GEOM_TYPES[g.get("type", "sphere")]
        size = _vec(g.get("size"), None, default=[0, 0, 0])
        density = float(g.get("density", 1000.0))
        if gtype == GEOM_BOX:
            vol = 8 * size[0] * size[1] * size[2]
            unit = np.array([size[1] ** 2 + size[2] ** 2, size[0] ** 2 + size[2] ** 2,
                             size[0] ** 2 + size[1] ** 2]) / 3.0
        elif gtype == GEOM_SPHERE:
            vol = 4.0 / 3.0 * math.pi * size[0] ** 3
            unit = np.full(3, 2.0 / 5.0 * size[0] ** 2)
        elif gtype == GEOM_CYLINDER:
            r, h = size[0], size[1]
            vol = math.pi * r * r * 2 * h
            unit = np.array([(3 * r * r + 4 * h * h) / 12.0] * 2 + [r * r / 2.0])
        elif gtype == GEOM_ELLIPSOID:
            vol = 4.0 / 3.0 * math.pi * size[0] * size[1] * size[2]
            unit = np.array([size[1] ** 2 + size[2] ** 2, size[0] ** 2 + size[2] ** 2,
                             size[0] ** 2 + size[1] ** 2]) / 5.0
        else:
            raise NotImplementedError(f"inertia of geom type {g.get('type')}")
        mass = float(g["mass"]) if "mass" in g else density * vol
        return mass, unit * mass

    def _body_inertial(self, b: _Body):
        """Returns (mass, ipos, iquat, diaginertia) in the body frame."""
        use_geoms = (self.inertiafromgeom == "true") or (self.inertiafromgeom == "auto" and b.inertial is None)
        if use_geoms:
            # MuJoCo only overrides the <inertial> element when the body has geoms in the inertia group range
            has = any(self.inertiagrouprange[0] <= int(g.get("group", 0)) <= self.inertiagrouprange[1]
                      and _GEOM_TYPES[g.get("type", "sphere")] not in (GEOM_PLANE, GEOM_HFIELD, GEOM_MESH) for g in b.geoms)
            if not has and b.inertial is not None:
                use_geoms = False
        if not use_geoms:
            if b.inertial is None:
                return 0.0, np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3)
            a = b.inertial
            mass = float(a["mass"])
            ipos = _vec(a.get("pos"), 3, default=[0, 0, 0])
            iquat = self._orient(a)
            if "diaginertia" in a:
                inertia = _vec(a["diaginertia"], 3)
            elif "fullinertia" in a:
                f = _vec(a["fullinertia"], 6)
                full = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                inertia, iquat2 = self._principal(full)
                iquat = quat_mul(iquat, iquat2)
            else:
                inertia = np.zeros(3)
            return mass, ipos, iquat, inertia
        # accumulate geoms in the inertia group range
        geoms = [g for g in b.geoms
                 if self.inertiagrouprange[0] <= int(g.get("group", 0)) <= self.inertiagrouprange[1]
                 and _GEOM_TYPES[g.get("type", "sphere")] not in (GEOM_PLANE, GEOM_HFIELD, GEOM_MESH)]
        if not geoms:
            return 0.0, np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3)
        masses, poss, full_list = [], [], []
        for g in geoms:
            m, diag = self._geom_mass_inertia(g)
            gp = _vec(g.get("pos"), 3, default=[0, 0, 0])
            gq = self._orient(g)
            R = quat_to_mat(gq)
            masses.append(m)
            poss.append(gp)
            full_list.append(R @ np.diag(diag) @ R.T)
        mass = float(sum(masses))
        if mass < MJ_MINVAL:
            return 0.0, np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3)
        com = sum(m * p for m, p in zip(masses, poss)) / mass
        full = np.zeros((3, 3))
        for m, p, I in zip(masses, poss, full_list):
            d = p - com
            full += I + m * ((d @ d) * np.eye(3) - np.outer(d, d))
        inertia, iquat = self._principal(full)
        return mass, com, iquat, inertia

    @staticmethod
    def _principal(full: np.ndarray):
        """Eigen-decomposition sorted by decreasing eigenvalue, right-handed frame (MuJoCo mju_eig3 order)."""
        w, v = np.linalg.eigh(full)
        order = np.argsort(-w)
        w, v = w[order], v[:, order]
        if np.linalg.det(v) < 0:
            v[:, 2] = -v[:, 2]
        return w, mat_to_quat(v)

    # ---- main ----
    def compile(self) -> CompiledModel:
        self._parse_compiler_option()
        for d in self.root.findall("default"):
            self.defaults.parse(d)
        self._parse_assets()

        world = _Body()
        world.name = "world"
        self.bodies.append(world)
        for wb in self.root.findall("worldbody"):
            self._parse_body_children(wb, 0, None)

        nbody = len(self.bodies)
        A: Dict[str, np.ndarray] = {}
        names: Dict[str, Dict[str, int]] = {k: {} for k in ("body", "joint", "geom", "site", "actuator", "sensor")}

        # ---------------- bodies, joints, dofs ----------------
        body_parentid = np.array([max(b.parent, 0) for b in self.bodies], dtype=np.int32)
        body_pos = np.stack([b.pos for b in self.bodies])
        body_quat = np.stack([b.quat for b in self.bodies])
        body_mass = np.zeros(nbody)
        body_ipos = np.zeros((nbody, 3))
        body_iquat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
        body_inertia = np.zeros((nbody, 3))
        body_jntnum = np.zeros(nbody, dtype=np.int32)
        body_jntadr = np.full(nbody, -1, dtype=np.int32)
        body_dofnum = np.zeros(nbody, dtype=np.int32)
        body_dofadr = np.full(nbody, -1, dtype=np.int32)

        jnt = dict(type=[], qposadr=[], dofadr=[], bodyid=[], pos=[], axis=[], limited=[], range=[],
                   actfrclimited=[], actfrcrange=[], solref=[], solimp=[], margin=[], stiffness=[])
        dof = dict(bodyid=[], jntid=[], armature=[], damping=[], frictionloss=[], solref=[], solimp=[])
        qpos0: List[float] = []

        def solimp5(s, default=(0.9, 0.95, 0.001, 0.5, 2.0)):
            v = list(default)
            if s is not None:
                x = _vec(s)
                v[:x.size] = x
            return np.array(v)

        def solref2(s, default=(0.02, 1.0)):
            v = list(default)
            if s is not None:
                x = _vec(s)
                v[:x.size] = x
            return np.array(v)

        def limited_flag(attr: Optional[str], has_range: bool) -> bool:
            if attr in ("true", "false"):
                return attr == "true"
            return has_range if self.autolimits else False

        for bid, b in enumerate(self.bodies):
            names["body"][b.name] = bid
            if bid == 0:
                continue
            mass, ipos, iquat, inertia = self._body_inertial(b)
            body_mass[bid], body_ipos[bid], body_iquat[bid], body_inertia[bid] = mass, ipos, iquat, inertia
            body_jntnum[bid] = len(b.joints)
            dof_start = len(dof["bodyid"])
            if b.joints:
                body_jntadr[bid] = len(jnt["type"])
                body_dofadr[bid] = len(dof["bodyid"])
            for j in b.joints:
                jt = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}[j.get("type", "hinge")]
                jid = len(jnt["type"])
                names["joint"][j.get("name", f"joint{jid}")] = jid
                jnt["type"].append(jt)
                jnt["qposadr"].append(len(qpos0))
                jnt["dofadr"].append(len(dof["bodyid"]))
                jnt["bodyid"].append(bid)
                jnt["pos"].append(_vec(j.get("pos"), 3, default=[0, 0, 0]))
                axis = _vec(j.get("axis"), 3, default=[0, 0, 1])
                jnt["axis"].append(axis / max(np.linalg.norm(axis), MJ_MINVAL))
                rng = _vec(j.get("range"), 2, default=[0, 0])
                if self.angle_deg and jt == JNT_HINGE:
                    rng = np.deg2rad(rng)
                jnt["range"].append(rng)
                jnt["limited"].append(limited_flag(j.get("limited"), "range" in j) and jt in (JNT_SLIDE, JNT_HINGE, JNT_BALL))
                afr = _vec(j.get("actuatorfrcrange"), 2, default=[0, 0])
                jnt["actfrcrange"].append(afr)
                jnt["actfrclimited"].append(limited_flag(j.get("actuatorfrclimited"), "actuatorfrcrange" in j))
                jnt["solref"].append(solref2(j.get("solreflimit")))
                jnt["solimp"].append(solimp5(j.get("solimplimit")))
                jnt["margin"].append(float(j.get("margin", 0)))
                jnt["stiffness"].append(float(j.get("stiffness", 0)))
                ndof = {JNT_FREE: 6, JNT_BALL: 3, JNT_SLIDE: 1, JNT_HINGE: 1}[jt]
                for _ in range(ndof):
                    dof["bodyid"].append(bid)
                    dof["jntid"].append(jid)
                    dof["armature"].append(float(j.get("armature", 0)))
                    dof["damping"].append(float(j.get("damping", 0)))
                    dof["frictionloss"].append(float(j.get("frictionloss", 0)))
                    dof["solref"].append(solref2(j.get("solreffriction")))
                    dof["solimp"].append(solimp5(j.get("solimpfriction")))
                if jt == JNT_FREE:
                    qpos0.extend(list(b.pos) + list(b.quat))
                elif jt == JNT_BALL:
                    qpos0.extend([1.0, 0, 0, 0])
                else:
                    ref = float(j.get("ref", 0))
                    if self.angle_deg and jt == JNT_HINGE:
                        ref = math.radians(ref)
                    qpos0.append(ref)
            body_dofnum[bid] = len(dof["bodyid"]) - dof_start

        nv = len(dof["bodyid"])
        njnt = len(jnt["type"])
        # root / weld ids
        body_rootid = np.zeros(nbody, dtype=np.int32)
        body_weldid = np.zeros(nbody, dtype=np.int32)
        for bid in range(1, nbody):
            p = body_parentid[bid]
            body_rootid[bid] = bid if p == 0 else body_rootid[p]
            body_weldid[bid] = bid if body_jntnum[bid] > 0 else body_weldid[p]
        # dof parent: previous dof of the same body, else last dof of nearest ancestor with dofs
        dof_parentid = np.full(nv, -1, dtype=np.int32)
        for d in range(nv):
            bid = dof["bodyid"][d]
            if d > body_dofadr[bid]:
                dof_parentid[d] = d - 1
            else:
                p = body_parentid[bid]
                while p > 0 and body_dofnum[p] == 0:
                    p = body_parentid[p]
                if p > 0:
                    dof_parentid[d] = body_dofadr[p] + body_dofnum[p] - 1

        A.update(
            body_parentid=body_parentid, body_rootid=body_rootid, body_weldid=body_weldid,
            body_jntnum=body_jntnum, body_jntadr=body_jntadr, body_dofnum=body_dofnum, body_dofadr=body_dofadr,
            body_pos=body_pos, body_quat=body_quat, body_ipos=body_ipos, body_iquat=body_iquat,
            body_mass=body_mass, body_inertia=body_inertia,
            jnt_type=np.array(jnt["type"], dtype=np.int32), jnt_qposadr=np.array(jnt["qposadr"], dtype=np.int32),
            jnt_dofadr=np.array(jnt["dofadr"], dtype=np.int32), jnt_bodyid=np.array(jnt["bodyid"], dtype=np.int32),
            jnt_pos=np.array(jnt["pos"]).reshape(njnt, 3), jnt_axis=np.array(jnt["axis"]).reshape(njnt, 3),
            jnt_limited=np.array(jnt["limited"], dtype=np.int32), jnt_range=np.array(jnt["range"]).reshape(njnt, 2),
            jnt_actfrclimited=np.array(jnt["actfrclimited"], dtype=np.int32),
            jnt_actfrcrange=np.array(jnt["actfrcrange"]).reshape(njnt, 2),
            jnt_solref=np.array(jnt["solref"]).reshape(njnt, 2), jnt_solimp=np.array(jnt["solimp"]).reshape(njnt, 5),
            jnt_margin=np.array(jnt["margin"]), jnt_stiffness=np.array(jnt["stiffness"]),
            dof_bodyid=np.array(dof["bodyid"], dtype=np.int32), dof_jntid=np.array(dof["jntid"], dtype=np.int32),
            dof_parentid=dof_parentid, dof_armature=np.array(dof["armature"]), dof_damping=np.array(dof["damping"]),
            dof_frictionloss=np.array(dof["frictionloss"]),
            dof_solref=np.array(dof["solref"]).reshape(nv, 2), dof_solimp=np.array(dof["solimp"]).reshape(nv, 5),
            qpos0=np.array(qpos0),
        )

        # ---------------- geoms / sites ----------------
        g_rows = dict(type=[], bodyid=[], contype=[], conaffinity=[], condim=[], priority=[], size=[], pos=[],
                      quat=[], friction=[], solmix=[], solref=[], solimp=[], margin=[], gap=[], hfield=[])
        s_rows = dict(bodyid=[], pos=[], quat=[])
        for bid, b in enumerate(self.bodies):
            for g in b.geoms:
                gid = len(g_rows["type"])
                names["geom"][g.get("name", f"geom{gid}")] = gid
                gt = _GEOM_TYPES[g.get("type", "sphere")]
                g_rows["type"].append(gt)
                g_rows["bodyid"].append(bid)
                g_rows["contype"].append(int(g.get("contype", 1)))
                g_rows["conaffinity"].append(int(g.get("conaffinity", 1)))
                g_rows["condim"].append(int(g.get("condim", 3)))
                g_rows["priority"].append(int(g.get("priority", 0)))
                size = np.zeros(3)
                sv = _vec(g.get("size"), None, default=[0, 0, 0])
                size[:min(3, sv.size)] = sv[:3]
                gpos, gquat = _vec(g.get("pos"), 3, default=[0, 0, 0]), self._orient(g)
                if g.get("fromto") is not None:
                    # MuJoCo's compiler: the geom sits at the segment's midpoint, its z axis along from - to, and the half length
                    # is the segment's (capsule / cylinder: size[1]; box / ellipsoid: size[2])
                    ft = _vec(g.get("fromto"), 6)
                    vec = ft[:3] - ft[3:]
                    ln = float(np.linalg.norm(vec))
                    if gt in (GEOM_CAPSULE, GEOM_CYLINDER):
                        size[1] = ln / 2
                    elif gt in (GEOM_BOX, GEOM_ELLIPSOID):
                        size[2] = ln / 2
                    else:
                        raise NotImplementedError(f"fromto on geom type {g.get('type')}")
                    gpos = 0.5 * (ft[:3] + ft[3:])
                    gquat = self._orient({"zaxis": " ".join(repr(float(x)) for x in vec)})
                g_rows["size"].append(size)
                g_rows["pos"].append(gpos)
                g_rows["quat"].append(gquat)
                fr = np.array([1.0, 0.005, 0.0001])
                fv = _vec(g.get("friction"))
                if fv is not None:
                    fr[:fv.size] = fv
                g_rows["friction"].append(fr)
                g_rows["solmix"].append(float(g.get("solmix", 1.0)))
                g_rows["solref"].append(solref2(g.get("solref")))
                g_rows["solimp"].append(solimp5(g.get("solimp")))
                g_rows["margin"].append(float(g.get("margin", 0)))
                g_rows["gap"].append(float(g.get("gap", 0)))
                g_rows["hfield"].append(g.get("hfield", ""))
            for s in b.sites:
                sid = len(s_rows["bodyid"])
                names["site"][s.get("name", f"site{sid}")] = sid
                s_rows["bodyid"].append(bid)
                s_rows["pos"].append(_vec(s.get("pos"), 3, default=[0, 0, 0]))
                s_rows["quat"].append(self._orient(s))
        ngeom = len(g_rows["type"])
        nsite = len(s_rows["bodyid"])
        A.update(
            geom_type=np.array(g_rows["type"], dtype=np.int32), geom_bodyid=np.array(g_rows["bodyid"], dtype=np.int32),
            geom_contype=np.array(g_rows["contype"], dtype=np.int32),
            geom_conaffinity=np.array(g_rows["conaffinity"], dtype=np.int32),
            geom_condim=np.array(g_rows["condim"], dtype=np.int32),
            geom_priority=np.array(g_rows["priority"], dtype=np.int32),
            geom_size=np.array(g_rows["size"]).reshape(ngeom, 3), geom_pos=np.array(g_rows["pos"]).reshape(ngeom, 3),
            geom_quat=np.array(g_rows["quat"]).reshape(ngeom, 4),
            geom_friction=np.array(g_rows["friction"]).reshape(ngeom, 3), geom_solmix=np.array(g_rows["solmix"]),
            geom_solref=np.array(g_rows["solref"]).reshape(ngeom, 2),
            geom_solimp=np.array(g_rows["solimp"]).reshape(ngeom, 5),
            geom_margin=np.array(g_rows["margin"]), geom_gap=np.array(g_rows["gap"]),
            site_bodyid=np.array(s_rows["bodyid"], dtype=np.int32),
            site_pos=np.array(s_rows["pos"]).reshape(nsite, 3), site_quat=np.array(s_rows["quat"]).reshape(nsite, 4),
        )
        self._geom_hfield = g_rows["hfield"]

        # ---------------- equality ----------------
        eq_rows = dict(obj1=[], obj2=[], data=[], solref=[], solimp=[], active=[])
        for eq in self.root.findall("equality"):
            for e in eq:
                if e.tag != "joint":
                    raise NotImplementedError(f"equality type {e.tag}")
                eq_rows["obj1"].append(names["joint"][e.get("joint1")])
                eq_rows["obj2"].append(names["joint"][e.get("joint2")] if e.get("joint2") else -1)
                pc = np.array([0.0, 1.0, 0, 0, 0])
                pv = _vec(e.get("polycoef"))
                if pv is not None:
                    pc[:pv.size] = pv
                eq_rows["data"].append(pc)
                eq_rows["solref"].append(solref2(e.get("solref")))
                eq_rows["solimp"].append(solimp5(e.get("solimp")))
                eq_rows["active"].append(e.get("active", "true") == "true")
        neq = len(eq_rows["obj1"])
        A.update(
            eq_obj1id=np.array(eq_rows["obj1"], dtype=np.int32), eq_obj2id=np.array(eq_rows["obj2"], dtype=np.int32),
            eq_data=np.array(eq_rows["data"]).reshape(neq, 5), eq_solref=np.array(eq_rows["solref"]).reshape(neq, 2),
            eq_solimp=np.array(eq_rows["solimp"]).reshape(neq, 5), eq_active0=np.array(eq_rows["active"], dtype=np.int32),
        )

        # ---------------- actuators ----------------
        act = dict(trnid=[], gear=[], gainprm=[], biasprm=[], ctrllimited=[], ctrlrange=[], forcelimited=[],
                   forcerange=[])
        for sec in self.root.findall("actuator"):
            for a0 in sec:
                if a0.tag not in ("position", "motor", "general", "velocity"):
                    raise NotImplementedError(f"actuator {a0.tag}")
                # actuators take defaults from their class only (no childclass)
                a = self.defaults.resolve(a0.tag, a0.attrib, None)
                if "joint" not in a:
                    raise NotImplementedError("only joint transmissions are supported")
                aid = len(act["trnid"])
                names["actuator"][a.get("name", f"actuator{aid}")] = aid
                jid = names["joint"][a["joint"]]
                act["trnid"].append(jid)
                gear = _vec(a.get("gear"), None, default=[1.0])
                act["gear"].append(float(gear[0]))
                gain = np.zeros(3)
                bias = np.zeros(3)
                if a0.tag == "position":
                    kp = float(a.get("kp", 1.0))
                    kv = float(a.get("kv", 0.0))
                    gain[0] = kp
                    bias[1], bias[2] = -kp, -kv
                elif a0.tag == "velocity":
                    kv = float(a.get("kv", 1.0))
                    gain[0] = kv
                    bias[2] = -kv
                elif a0.tag == "motor":
                    gain[0] = 1.0
                else:
                    gv = _vec(a.get("gainprm"), None, default=[1.0])
                    bv = _vec(a.get("biasprm"), None, default=[0.0])
                    gain[:min(3, gv.size)] = gv[:3]
                    bias[:min(3, bv.size)] = bv[:3]
                act["gainprm"].append(gain)
                act["biasprm"].append(bias)
                cr = _vec(a.get("ctrlrange"), 2, default=[0, 0])
                has_cr = "ctrlrange" in a
                if a.get("inheritrange") is not None and float(a.get("inheritrange")) > 0 and not has_cr:
                    ir = float(a["inheritrange"])
                    jr = A["jnt_range"][jid]
                    mean, rad = 0.5 * (jr[0] + jr[1]), 0.5 * (jr[1] - jr[0]) * ir
                    cr = np.array([mean - rad, mean + rad])
                    has_cr = True
                act["ctrlrange"].append(cr)
                act["ctrllimited"].append(limited_flag(a.get("ctrllimited"), has_cr))
                fr = _vec(a.get("forcerange"), 2, default=[0, 0])
                act["forcerange"].append(fr)
                act["forcelimited"].append(limited_flag(a.get("forcelimited"), "forcerange" in a))
        nu = len(act["trnid"])
        A.update(
            actuator_trnid=np.array(act["trnid"], dtype=np.int32), actuator_gear=np.array(act["gear"]),
            actuator_gainprm=np.array(act["gainprm"]).reshape(nu, 3), actuator_biasprm=np.array(act["biasprm"]).reshape(nu, 3),
            actuator_ctrllimited=np.array(act["ctrllimited"], dtype=np.int32),
            actuator_ctrlrange=np.array(act["ctrlrange"]).reshape(nu, 2),
            actuator_forcelimited=np.array(act["forcelimited"], dtype=np.int32),
            actuator_forcerange=np.array(act["forcerange"]).reshape(nu, 2),
        )

        # ---------------- keyframes ----------------
        key_qpos, key_ctrl = [], []
        names["key"] = {}
        for sec in self.root.findall("keyframe"):
            for k in sec.findall("key"):
                names["key"][k.get("name", f"key{len(key_qpos)}")] = len(key_qpos)
                key_qpos.append(_vec(k.get("qpos"), None, default=A["qpos0"]))
                key_ctrl.append(_vec(k.get("ctrl"), None, default=np.zeros(nu)))
        A["key_qpos"] = np.array(key_qpos).reshape(len(key_qpos), len(qpos0))
        A["key_ctrl"] = np.array(key_ctrl).reshape(len(key_ctrl), nu)

        # ---------------- exclude list & pair list ----------------
        excl = set()
        for sec in self.root.findall("contact"):
            for e in sec.findall("exclude"):
                b1, b2 = names["body"][e.get("body1")], names["body"][e.get("body2")]
                excl.add((min(b1, b2) << 16) + max(b1, b2))
        self._build_pairs(A, excl)

        # ---------------- options ----------------
        o = self.opt
        A["opt_timestep"] = np.array([o["timestep"]])
        A["opt_gravity"] = np.array(o["gravity"], dtype=np.float64)
        A["opt_tolerance"] = np.array([o["tolerance"]])
        A["opt_ls_tolerance"] = np.array([o["ls_tolerance"]])
        A["opt_impratio"] = np.array([o["impratio"]])
        A["opt_iterations"] = np.array([o["iterations"]], dtype=np.int32)
        A["opt_ls_iterations"] = np.array([o["ls_iterations"]], dtype=np.int32)
        A["opt_integrator"] = np.array([o["integrator"]], dtype=np.int32)
        A["opt_disable_eulerdamp"] = np.array([int(o["disable_eulerdamp"])], dtype=np.int32)
        A["opt_disable_refsafe"] = np.array([int(o["disable_refsafe"])], dtype=np.int32)

        # ---------------- heightfields ----------------
        self._load_hfields(A)

        # ---------------- constants at qpos0 ----------------
        m = CompiledModel(name=self.root.get("model", ""), arrays=A, names=names)
        set_const(m)
        return m

    # ---- collision pair enumeration (MJX collision_driver rule) ----
    def _build_pairs(self, A, excl):
        nbody = A["body_parentid"].shape[0]
        gb = A["geom_bodyid"]
        ct, ca, gt = A["geom_contype"], A["geom_conaffinity"], A["geom_type"]
        weld, par = A["body_weldid"], A["body_parentid"]
        body_geoms = [[g for g in range(gb.shape[0]) if gb[g] == b and (ct[g] | ca[g])] for b in range(nbody)]
        P = dict(g1=[], g2=[], kind=[], condim=[], friction=[], solref=[], solimp=[], margin=[], gap=[])
        for b1 in range(nbody):
            if not body_geoms[b1]:
                continue
            w1 = weld[b1]
            w1p = weld[par[w1]]
            for b2 in range(b1, nbody):
                if not body_geoms[b2]:
                    continue
                if ((b1 << 16) + b2) in excl:
                    continue
                w2 = weld[b2]
                if w1 == w2:
                    continue
                w2p = weld[par[w2]]
                if (not self.opt["disable_filterparent"]) and w1 != 0 and w2 != 0 and (w1 == w2p or w2 == w1p):
                    continue
                for g1 in body_geoms[b1]:
                    for g2 in body_geoms[b2]:
                        a, b = (g1, g2) if gt[g1] <= gt[g2] else (g2, g1)
                        ta, tb = gt[a], gt[b]
                        if (ta, tb) in ((GEOM_PLANE, GEOM_PLANE), (GEOM_PLANE, GEOM_HFIELD)):
                            continue
                        if not ((ct[a] & ca[b]) | (ct[b] & ca[a])):
                            continue
                        if (ta, tb) == (GEOM_PLANE, GEOM_BOX):
                            kind = PAIR_PLANE_BOX
                        elif (ta, tb) == (GEOM_BOX, GEOM_BOX):
                            kind = PAIR_BOX_BOX
                        elif (ta, tb) == (GEOM_PLANE, GEOM_SPHERE):
                            kind = PAIR_PLANE_SPHERE
                        elif (ta, tb) == (GEOM_HFIELD, GEOM_SPHERE):
                            kind = PAIR_HFIELD_SPHERE
                        elif (ta, tb) == (GEOM_PLANE, GEOM_CAPSULE):
                            kind = PAIR_PLANE_CAPSULE
                        elif (ta, tb) == (GEOM_PLANE, GEOM_CYLINDER):
                            kind = PAIR_PLANE_CYLINDER
                        else:
                            raise NotImplementedError(f"collision pair of geom types {ta},{tb}")
                        P["g1"].append(a)
                        P["g2"].append(b)
                        P["kind"].append(kind)
                        self._mix(A, a, b, P)
        n = len(P["g1"])
        A["pair_geom1"] = np.array(P["g1"], dtype=np.int32)
        A["pair_geom2"] = np.array(P["g2"], dtype=np.int32)
        A["pair_kind"] = np.array(P["kind"], dtype=np.int32)
        A["pair_condim"] = np.array(P["condim"], dtype=np.int32)
        A["pair_solref"] = np.array(P["solref"]).reshape(n, 2)
        A["pair_solimp"] = np.array(P["solimp"]).reshape(n, 5)
        A["pair_margin"] = np.array(P["margin"]).reshape(n)
        A["pair_gap"] = np.array(P["gap"]).reshape(n)

    @staticmethod
    def _mix(A, g1, g2, P):
        """Contact parameter mixing for a geom pair; friction is mixed at run time because
        domain randomisation overrides geom_friction per env (domain_randomize.py:63-66)."""
        p1, p2 = A["geom_priority"][g1], A["geom_priority"][g2]
        s1, s2 = A["geom_solmix"][g1], A["geom_solmix"][g2]
        if s1 >= MJ_MINVAL and s2 >= MJ_MINVAL:
            mix = s1 / (s1 + s2)
        elif s1 < MJ_MINVAL and s2 < MJ_MINVAL:
            mix = 0.5
        elif s1 < MJ_MINVAL:
            mix = 0.0
        else:
            mix = 1.0
        r1, r2 = A["geom_solref"][g1], A["geom_solref"][g2]
        i1, i2 = A["geom_solimp"][g1], A["geom_solimp"][g2]
        if p1 == p2:
            if r1[0] > 0 and r2[0] > 0:
                solref = mix * r1 + (1 - mix) * r2
            else:
                solref = np.minimum(r1, r2)
            solimp = mix * i1 + (1 - mix) * i2
            condim = max(A["geom_condim"][g1], A["geom_condim"][g2])
        else:
            w = g1 if p1 > p2 else g2
            solref, solimp, condim = A["geom_solref"][w], A["geom_solimp"][w], A["geom_condim"][w]
        P["condim"].append(int(condim))
        P["solref"].append(solref)
        P["solimp"].append(solimp)
        P["margin"].append(max(A["geom_margin"][g1], A["geom_margin"][g2]))
        P["gap"].append(max(A["geom_gap"][g1], A["geom_gap"][g2]))

    def _load_hfields(self, A):
        A["hfield_size"] = np.zeros((0, 4))
        A["hfield_nrow"] = np.zeros(0, dtype=np.int32)
        A["hfield_ncol"] = np.zeros(0, dtype=np.int32)
        A["hfield_data"] = np.zeros(0)
        used = [h for h in self._geom_hfield if h]
        if not used:
            return
        from .png import read_png_gray  # local tiny PNG reader (no PIL on the target)
        sizes, nrow, ncol, data = [], [], [], []
        for hname in dict.fromkeys(used):
            h = self.hfields[hname]
            # the reference loads assets through a name-keyed dict (base.py:18-24), so `file` may or may not repeat meshdir
            cands = [os.path.join(self.dir, self.meshdir, h["file"]), os.path.join(self.dir, h["file"]),
                     os.path.join(self.dir, self.meshdir, os.path.basename(h["file"]))]
            path = next((c for c in cands if os.path.exists(c)), cands[0])
            img = read_png_gray(path).astype(np.float64)
            # MuJoCo flips the image vertically and normalises to [0, 1]
            img = img[::-1, :]
            lo, hi = img.min(), img.max()
            img = (img - lo) / (hi - lo) if hi > lo else np.zeros_like(img)
            sizes.append(_vec(h["size"], 4))
            nrow.append(img.shape[0])
            ncol.append(img.shape[1])
            data.append(img.reshape(-1))
        A["hfield_size"] = np.array(sizes).reshape(len(sizes), 4)
        A["hfield_nrow"] = np.array(nrow, dtype=np.int32)
        A["hfield_ncol"] = np.array(ncol, dtype=np.int32)
        A["hfield_data"] = np.concatenate(data)


# --------------------------------------------------------------------------
# kinematics / mass matrix at a configuration (float64) -- used for the
# compile-time constants and by the host-side tests
# --------------------------------------------------------------------------
def forward_kinematics(m: CompiledModel, qpos: np.ndarray):
    """Returns dict with xpos, xquat, xmat, xipos, ximat, xanchor, xaxis (world frame)."""
    A = m.arrays
    nbody = m.nbody
    xpos = np.zeros((nbody, 3))
    xquat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
    xanchor = np.zeros((m.njnt, 3))
    xaxis = np.zeros((m.njnt, 3))
    for b in range(1, nbody):
        p = A["body_parentid"][b]
        pos = xpos[p] + rotate(A["body_pos"][b], xquat[p])
        quat = quat_mul(xquat[p], A["body_quat"][b])
        for k in range(A["body_jntnum"][b]):
            j = A["body_jntadr"][b] + k
            jt, qa = A["jnt_type"][j], A["jnt_qposadr"][j]
            if jt == JNT_FREE:
                xanchor[j] = qpos[qa:qa + 3]
                xaxis[j] = np.array([0, 0, 1.0])
                pos = qpos[qa:qa + 3].copy()
                quat = qpos[qa + 3:qa + 7] / np.linalg.norm(qpos[qa + 3:qa + 7])
            else:
                anchor = rotate(A["jnt_pos"][j], quat) + pos
                axis = rotate(A["jnt_axis"][j], quat)
                xanchor[j], xaxis[j] = anchor, axis
                if jt == JNT_HINGE:
                    qloc = axisangle_to_quat(A["jnt_axis"][j], qpos[qa] - A["qpos0"][qa])
                    quat = quat_mul(quat, qloc)
                    pos = anchor - rotate(A["jnt_pos"][j], quat)
                elif jt == JNT_SLIDE:
                    pos = pos + axis * (qpos[qa] - A["qpos0"][qa])
                else:
                    raise NotImplementedError("ball joints")
        xpos[b], xquat[b] = pos, quat / np.linalg.norm(quat)
    xmat = np.stack([quat_to_mat(q) for q in xquat])
    xipos = np.stack([xpos[b] + xmat[b] @ A["body_ipos"][b] for b in range(nbody)])
    ximat = np.stack([quat_to_mat(quat_mul(xquat[b], A["body_iquat"][b])) for b in range(nbody)])
    return dict(xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, ximat=ximat, xanchor=xanchor, xaxis=xaxis)


def body_jacobian(m: CompiledModel, kin: dict, point: np.ndarray, body: int):
    """World-frame translational (3,nv) and rotational (3,nv) Jacobian of `point` attached to `body`."""
    A = m.arrays
    nv = m.nv
    jacp, jacr = np.zeros((3, nv)), np.zeros((3, nv))
    b = body
    while b > 0:
        for k in range(A["body_jntnum"][b]):
            j = A["body_jntadr"][b] + k
            jt, da = A["jnt_type"][j], A["jnt_dofadr"][j]
            if jt == JNT_FREE:
                jacp[:, da:da + 3] = np.eye(3)
                R = kin["xmat"][b]
                for a in range(3):
                    ax = R[:, a]
                    jacr[:, da + 3 + a] = ax
                    jacp[:, da + 3 + a] = np.cross(ax, point - kin["xpos"][b])
            elif jt == JNT_HINGE:
                ax = kin["xaxis"][j]
                jacr[:, da] = ax
                jacp[:, da] = np.cross(ax, point - kin["xanchor"][j])
            elif jt == JNT_SLIDE:
                jacp[:, da] = kin["xaxis"][j]
        b = A["body_parentid"][b]
    return jacp, jacr


def mass_matrix(m: CompiledModel, qpos: np.ndarray, kin: Optional[dict] = None, body_mass=None) -> np.ndarray:
    """Joint-space inertia M(q) = sum_b Jp^T m Jp + Jr^T I Jr (+ armature), float64."""
    A = m.arrays
    kin = kin or forward_kinematics(m, qpos)
    nv = m.nv
    M = np.zeros((nv, nv))
    mass = A["body_mass"] if body_mass is None else body_mass
    for b in range(1, m.nbody):
        if mass[b] <= 0 and not np.any(A["body_inertia"][b] > 0):
            continue
        jp_, jr_ = body_jacobian(m, kin, kin["xipos"][b], b)
        I = kin["ximat"][b] @ np.diag(A["body_inertia"][b]) @ kin["ximat"][b].T
        M += mass[b] * jp_.T @ jp_ + jr_.T @ I @ jr_
    M += np.diag(A["dof_armature"])
    return M


def set_const(m: CompiledModel) -> None:
    """dof_invweight0, body_invweight0, stat_meaninertia at qpos0 (MuJoCo mj_setConst / set0)."""
    A = m.arrays
    nv = m.nv
    kin = forward_kinematics(m, A["qpos0"])
    M = mass_matrix(m, A["qpos0"], kin)
    Minv = np.linalg.inv(M) if nv else np.zeros((0, 0))
    A["stat_meaninertia"] = np.array([max(MJ_MINVAL, float(np.trace(M)) / max(1, nv))])
    d = np.diag(Minv).copy()
    for j in range(m.njnt):
        da = A["jnt_dofadr"][j]
        if A["jnt_type"][j] == JNT_FREE:
            d[da:da + 3] = d[da:da + 3].mean()
            d[da + 3:da + 6] = d[da + 3:da + 6].mean()
        elif A["jnt_type"][j] == JNT_BALL:
            d[da:da + 3] = d[da:da + 3].mean()
    A["dof_invweight0"] = d
    biw = np.zeros((m.nbody, 2))
    for b in range(1, m.nbody):
        if A["body_weldid"][b] == 0:
            continue
        jp_, jr_ = body_jacobian(m, kin, kin["xipos"][b], b)
        biw[b, 0] = np.trace(jp_ @ Minv @ jp_.T) / 3.0
        biw[b, 1] = np.trace(jr_ @ Minv @ jr_.T) / 3.0
    A["body_invweight0"] = biw


def compile_mjcf(path: str) -> CompiledModel:
    return MjcfCompiler(path).compile()
