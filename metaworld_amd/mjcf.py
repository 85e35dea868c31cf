"""MJCF subset compiler: Meta-World scene XML -> flat numeric tables.

This is host-side, offline data preparation shared by the CPU oracle and the
HIP runtime (both consume the same `Model`).  It restates the MuJoCo *model
compiler* semantics needed by the 36 Meta-World scenes
(reference: metaworld/assets/sawyer_xyz/*.xml, objects/assets/*.xml,
scene/basic_scene.xml; feature census in SURVEY.md Appendix B.1):

  include, nested default classes + childclass, compiler(angle=radian,
  inertiafromgeom=auto, inertiagrouprange), body/inertial/joint/freejoint/
  geom/site/mesh, position actuators, weld equality, option.

Nothing here is copied from MuJoCo; the compile rules are restated from its
public documentation ([EXT] in SURVEY.md) and are therefore "parity unpinned".
"""
from __future__ import annotations

import os
import struct
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

# geom types (MuJoCo numbering: collision dispatch orders pairs by type)
G_PLANE, G_HFIELD, G_SPHERE, G_CAPSULE, G_ELLIPSOID, G_CYLINDER, G_BOX, G_MESH = range(8)
GEOM_TYPES = {"plane": G_PLANE, "sphere": G_SPHERE, "capsule": G_CAPSULE,
              "ellipsoid": G_ELLIPSOID, "cylinder": G_CYLINDER, "box": G_BOX, "mesh": G_MESH}
J_FREE, J_BALL, J_SLIDE, J_HINGE = range(4)
JNT_TYPES = {"free": J_FREE, "ball": J_BALL, "slide": J_SLIDE, "hinge": J_HINGE}

MINVAL = 1e-15
EIG_EPS = 1e-12


# ----------------------------------------------------------------------------
# small math helpers (quaternions are (w, x, y, z))
# ----------------------------------------------------------------------------
def qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw])


def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def qnorm(q):
    q = np.asarray(q, dtype=np.float64)
    n = np.linalg.norm(q)
    if n < MINVAL:
        return np.array([1.0, 0, 0, 0])
    return q / n


def q2mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def qrot(q, v):
    return q2mat(q) @ np.asarray(v, dtype=np.float64)


def euler2quat(e):
    """intrinsic x-y-z (MuJoCo default eulerseq 'xyz')."""
    q = np.array([1.0, 0, 0, 0])
    for i in range(3):
        h = 0.5 * e[i]
        r = np.zeros(4)
        r[0] = np.cos(h)
        r[1 + i] = np.sin(h)
        q = qmul(q, r)
    return q


def eig3(mat):
    """Symmetric 3x3 eigen-decomposition by quaternion Jacobi sweeps.

    Returns (eigval descending, quat) with mat = R diag(eigval) R^T, R = q2mat(quat).
    The iteration/sorting convention is restated from the documented behaviour of
    MuJoCo's 3x3 eigensolver so that principal-axis frames (body_iquat, mesh
    frames) come out with the same handedness/ordering.
    """
    quat = np.array([1.0, 0, 0, 0])
    eigval = np.zeros(3)
    for _ in range(500):
        R = q2mat(quat)
        D = R.T @ mat @ R
        eigval = np.array([D[0, 0], D[1, 1], D[2, 2]])
        a01, a02, a12 = abs(D[0, 1]), abs(D[0, 2]), abs(D[1, 2])
        if a01 > a02 and a01 > a12:
            rk, ck, rotk = 0, 1, 2
        elif a02 > a12:
            rk, ck, rotk = 0, 2, 1
        else:
            rk, ck, rotk = 1, 2, 0
        if abs(D[rk, ck]) < EIG_EPS:
            break
        tau = (D[ck, ck] - D[rk, rk]) / (2 * D[rk, ck])
        if tau >= 0:
            t = 1.0 / (tau + np.sqrt(1 + tau * tau))
        else:
            t = -1.0 / (-tau + np.sqrt(1 + tau * tau))
        c = 1.0 / np.sqrt(1 + t * t)
        if c > 1.0 - EIG_EPS:
            break
        tmp = np.zeros(4)
        s = np.sqrt(0.5 - 0.5 * c)
        tmp[rotk + 1] = -s if tau >= 0 else s
        if rotk == 1:
            tmp[rotk + 1] = -tmp[rotk + 1]
        tmp[0] = np.sqrt(1.0 - tmp[rotk + 1] ** 2)
        quat = qnorm(qmul(quat, qnorm(tmp)))
    for j in range(3):
        j1 = j % 2
        if eigval[j1] + EIG_EPS < eigval[j1 + 1]:
            eigval[j1], eigval[j1 + 1] = eigval[j1 + 1], eigval[j1]
            tmp = np.zeros(4)
            tmp[0] = 0.707106781186548
            tmp[(j1 + 2) % 3 + 1] = tmp[0]
            quat = qnorm(qmul(quat, tmp))
    return eigval, quat


# ----------------------------------------------------------------------------
# meshes
# ----------------------------------------------------------------------------
def load_stl(path):
    with open(path, "rb") as f:
        data = f.read()
    ntri = struct.unpack_from("<I", data, 80)[0]
    if 84 + 50 * ntri != len(data):
        # ASCII STL fallback
        verts = []
        for line in data.decode("ascii", "ignore").splitlines():
            p = line.split()
            if len(p) == 4 and p[0] == "vertex":
                verts.append([float(p[1]), float(p[2]), float(p[3])])
        return np.asarray(verts, dtype=np.float64).reshape(-1, 3, 3)
    rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]),
                        count=ntri, offset=84)
    return rec["v"].astype(np.float64)


def mesh_volume_props(tris):
    """signed-tetrahedra volume, COM, inertia tensor about COM (density 1)."""
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    det = np.einsum("ij,ij->i", a, np.cross(b, c))
    vol = det.sum() / 6.0
    if abs(vol) < MINVAL:
        cen = tris.reshape(-1, 3).mean(0)
        return 0.0, cen, np.zeros((3, 3))
    com = (det[:, None] * (a + b + c)).sum(0) / (24.0 * vol)
    # second moments: integral of x x^T over tetra(0,a,b,c) = det/120 * (S S^T + sum v v^T), S=a+b+c
    S = a + b + c
    P = (np.einsum("i,ij,ik->jk", det, S, S)
         + np.einsum("i,ij,ik->jk", det, a, a)
         + np.einsum("i,ij,ik->jk", det, b, b)
         + np.einsum("i,ij,ik->jk", det, c, c)) / 120.0
    if vol < 0:
        vol, P = -vol, -P
    P = P - vol * np.outer(com, com)
    I = np.trace(P) * np.eye(3) - P
    return vol, com, I


@dataclass
class Mesh:
    name: str
    vert: np.ndarray  # hull vertices in the mesh's principal frame
    pos: np.ndarray   # mesh frame origin (COM) in file coordinates
    quat: np.ndarray  # mesh frame orientation in file coordinates
    volume: float
    inertia: np.ndarray  # diag, unit density, principal frame


def build_mesh(name, path, scale):
    from scipy.spatial import ConvexHull
    tris = load_stl(path) * np.asarray(scale, dtype=np.float64)[None, None, :]
    if np.prod(scale) < 0:
        tris = tris[:, ::-1, :]
    vol, com, I = mesh_volume_props(tris)
    if vol <= 0:
        # degenerate / open mesh: fall back to convex hull properties
        pts = tris.reshape(-1, 3)
        hull = ConvexHull(pts)
        ht = pts[hull.simplices]
        cen = pts[hull.vertices].mean(0)
        nrm = np.cross(ht[:, 1] - ht[:, 0], ht[:, 2] - ht[:, 0])
        flip = np.einsum("ij,ij->i", nrm, ht[:, 0] - cen) < 0
        ht[flip] = ht[flip][:, ::-1, :]
        vol, com, I = mesh_volume_props(ht)
    ev, quat = eig3(I)
    R = q2mat(quat)
    pts = np.unique(np.round(tris.reshape(-1, 3), 12), axis=0)
    local = (pts - com) @ R
    hull = ConvexHull(local)
    hv = local[hull.vertices]
    return Mesh(name, hv.copy(), com, quat, vol, ev)


# ----------------------------------------------------------------------------
# XML handling
# ----------------------------------------------------------------------------
def _expand_includes(elem, basedir):
    i = 0
    while i < len(elem):
        ch = elem[i]
        if ch.tag == "include":
            sub = ET.parse(os.path.join(basedir, ch.attrib["file"])).getroot()
            _expand_includes(sub, basedir)
            elem.remove(ch)
            for k, sc in enumerate(list(sub)):
                elem.insert(i + k, sc)
            i += len(sub)
        else:
            _expand_includes(ch, basedir)
            i += 1


def _floats(s):
    return np.array([float(x) for x in s.split()], dtype=np.float64)


class _Defaults:
    def __init__(self):
        self.classes = {"main": {}}
        self.parent = {"main": None}

    def load(self, elem, parent="main"):
        name = elem.attrib.get("class", "main")
        if parent is None and name != "main":
            parent = "main"
        if name not in self.classes:
            base = self.classes[parent] if parent else {}
            self.classes[name] = {k: dict(v) for k, v in base.items()}
            self.parent[name] = parent
        cur = self.classes[name]
        for ch in elem:
            if ch.tag == "default":
                continue
            cur.setdefault(ch.tag, {}).update(ch.attrib)
        for ch in elem:
            if ch.tag == "default":
                self.load(ch, name)

    def get(self, tag, cls):
        return self.classes.get(cls or "main", self.classes["main"]).get(tag, {})


@dataclass
class Model:
    """Flat model tables; field names follow MuJoCo's mjModel where a counterpart exists."""
    name: str = ""
    opt_timestep: float = 0.002
    opt_iterations: int = 100
    opt_tolerance: float = 1e-8
    gravity: np.ndarray = field(default_factory=lambda: np.array([0, 0, -9.81]))
    names: dict = field(default_factory=dict)  # kind -> {name: id}
    arrays: dict = field(default_factory=dict)
    meshes: list = field(default_factory=list)

    def __getattr__(self, k):
        arr = self.__dict__.get("arrays")
        if arr is not None and k in arr:
            return arr[k]
        raise AttributeError(k)

    def id(self, kind, name):
        return self.names[kind][name]


def _frame(attrib):
    pos = _floats(attrib["pos"]) if "pos" in attrib else np.zeros(3)
    if "quat" in attrib:
        quat = qnorm(_floats(attrib["quat"]))
    elif "euler" in attrib:
        quat = euler2quat(_floats(attrib["euler"]))
    else:
        quat = np.array([1.0, 0, 0, 0])
    return pos, quat


def _geom_mass_inertia(gtype, size, density, mesh):
    """volume-based mass & diagonal inertia in the geom frame for unit handling below."""
    if gtype == G_SPHERE:
        r = size[0]
        vol = 4.0 / 3.0 * np.pi * r ** 3
        I = np.full(3, 0.4 * r * r)
    elif gtype == G_CAPSULE:
        r, h = size[0], size[1]
        height = 2 * h
        vc = np.pi * r * r * height
        vs = 4.0 / 3.0 * np.pi * r ** 3
        vol = vc + vs
        # per unit total mass
        mc, ms = vc / vol, vs / vol
        ix = mc * (r * r / 4 + height * height / 12) + ms * (0.4 * r * r + 0.375 * r * height + height * height / 4)
        iz = mc * r * r / 2 + ms * 0.4 * r * r
        I = np.array([ix, ix, iz])
    elif gtype == G_CYLINDER:
        r, h = size[0], size[1]
        height = 2 * h
        vol = np.pi * r * r * height
        I = np.array([(3 * r * r + height * height) / 12, (3 * r * r + height * height) / 12, r * r / 2])
    elif gtype == G_BOX:
        vol = 8 * size[0] * size[1] * size[2]
        I = np.array([(size[1] ** 2 + size[2] ** 2) / 3, (size[0] ** 2 + size[2] ** 2) / 3,
                      (size[0] ** 2 + size[1] ** 2) / 3])
    elif gtype == G_ELLIPSOID:
        vol = 4.0 / 3.0 * np.pi * size[0] * size[1] * size[2]
        I = np.array([(size[1] ** 2 + size[2] ** 2) / 5, (size[0] ** 2 + size[2] ** 2) / 5,
                      (size[0] ** 2 + size[1] ** 2) / 5])
    elif gtype == G_MESH:
        vol = mesh.volume
        I = mesh.inertia / max(vol, MINVAL)
    else:
        vol, I = 0.0, np.zeros(3)
    return vol, I  # I is inertia per unit mass


def compile_mjcf(path) -> Model:
    path = os.path.abspath(path)
    basedir = os.path.dirname(path)
    root = ET.parse(path).getroot()
    _expand_includes(root, basedir)

    comp = {}
    for c in root.iter("compiler"):
        comp.update(c.attrib)
    assert comp.get("angle", "degree") == "radian", "only angle=radian scenes are supported"
    grp_lo, grp_hi = (int(x) for x in comp.get("inertiagrouprange", "0 5").split())
    inertiafromgeom = comp.get("inertiafromgeom", "auto")

    m = Model(name=os.path.splitext(os.path.basename(path))[0])
    for o in root.iter("option"):
        m.opt_timestep = float(o.attrib.get("timestep", m.opt_timestep))
        m.opt_iterations = int(o.attrib.get("iterations", m.opt_iterations))
        m.opt_tolerance = float(o.attrib.get("tolerance", m.opt_tolerance))
        assert o.attrib.get("solver", "Newton") == "Newton"
        assert o.attrib.get("cone", "pyramidal") == "elliptic"

    dfl = _Defaults()
    for d in root.findall("default"):
        dfl.load(d, None)

    # mesh assets (lazily built when a kept geom uses them)
    mesh_decl = {}
    for a in root.findall("asset"):
        for me in a.findall("mesh"):
            at = dict(dfl.get("mesh", me.attrib.get("class")))
            at.update(me.attrib)
            nm = at.get("name") or os.path.splitext(os.path.basename(at["file"]))[0]
            mesh_decl[nm] = (os.path.join(basedir, at["file"]),
                             _floats(at["scale"]) if "scale" in at else np.ones(3))
    mesh_cache = {}

    def get_mesh(nm):
        if nm not in mesh_cache:
            f, sc = mesh_decl[nm]
            mesh_cache[nm] = build_mesh(nm, f, sc)
        return mesh_cache[nm]

    bodies, joints, geoms, sites = [], [], [], []
    names = {"body": {}, "joint": {}, "geom": {}, "site": {}, "mesh": {}}

    def merged(tag, elem, childclass):
        at = dict(dfl.get(tag, elem.attrib.get("class", childclass)))
        at.update(elem.attrib)
        return at

    def add_body(elem, parent, childclass):
        bid = len(bodies)
        if elem is None or elem.tag == "worldbody":
            b = dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), mocap=False)
            src = root.findall("worldbody")
        else:
            pos, quat = _frame(elem.attrib)
            b = dict(name=elem.attrib.get("name", ""), parent=parent, pos=pos, quat=quat,
                     mocap=elem.attrib.get("mocap", "false") == "true")
            childclass = elem.attrib.get("childclass", childclass)
            src = [elem]
        b.update(inertial=None, jnts=[], geoms=[])
        bodies.append(b)
        if b["name"]:
            names["body"][b["name"]] = bid
        for s in src:
            for ch in s:
                if ch.tag == "inertial":
                    ipos, iquat = _frame(ch.attrib)
                    b["inertial"] = dict(pos=ipos, quat=iquat, mass=float(ch.attrib["mass"]),
                                         inertia=_floats(ch.attrib["diaginertia"]))
                elif ch.tag in ("joint", "freejoint"):
                    at = merged("joint", ch, childclass) if ch.tag == "joint" else dict(ch.attrib, type="free")
                    jt = JNT_TYPES[at.get("type", "hinge")]
                    assert jt != J_BALL
                    rng = _floats(at["range"]) if "range" in at else np.zeros(2)
                    lim = at.get("limited", "auto")
                    limited = (lim == "true") or (lim == "auto" and "range" in at)
                    if jt == J_FREE:
                        limited = False
                    j = dict(name=at.get("name", ""), type=jt, body=bid,
                             pos=_floats(at["pos"]) if "pos" in at else np.zeros(3),
                             axis=_floats(at["axis"]) if "axis" in at else np.array([0, 0, 1.0]),
                             limited=limited, range=rng,
                             damping=float(at.get("damping", 0)), armature=float(at.get("armature", 0)),
                             stiffness=float(at.get("stiffness", 0)), springref=float(at.get("springref", 0)),
                             solref=_floats(at.get("solreflimit", "0.02 1")),
                             solimp=_floats(at.get("solimplimit", "0.9 0.95 0.001 0.5 2")),
                             margin=float(at.get("margin", 0)))
                    assert float(at.get("ref", 0)) == 0 and float(at.get("frictionloss", 0)) == 0
                    if jt != J_FREE:
                        j["axis"] = j["axis"] / np.linalg.norm(j["axis"])
                    if j["name"]:
                        names["joint"][j["name"]] = len(joints)
                    b["jnts"].append(len(joints))
                    joints.append(j)
                elif ch.tag == "geom":
                    at = merged("geom", ch, childclass)
                    gt = GEOM_TYPES[at.get("type", "sphere")]
                    assert "fromto" not in at
                    size = np.zeros(3)
                    if "size" in at:
                        sz = _floats(at["size"])
                        size[:len(sz)] = sz
                    pos, quat = _frame(at)
                    g = dict(name=at.get("name", ""), type=gt, body=bid, size=size, pos=pos, quat=quat,
                             contype=int(at.get("contype", 1)), conaffinity=int(at.get("conaffinity", 1)),
                             condim=int(at.get("condim", 3)), group=int(at.get("group", 0)),
                             friction=_floats(at.get("friction", "1 0.005 0.0001")),
                             solref=_floats(at.get("solref", "0.02 1")),
                             solimp=_floats(at.get("solimp", "0.9 0.95 0.001 0.5 2")),
                             margin=float(at.get("margin", 0)), gap=float(at.get("gap", 0)),
                             solmix=float(at.get("solmix", 1)), priority=int(at.get("priority", 0)),
                             mass=float(at["mass"]) if "mass" in at else None,
                             density=float(at.get("density", 1000)), mesh=at.get("mesh"))
                    if len(g["friction"]) < 3:
                        g["friction"] = np.concatenate([g["friction"], [0.005, 0.0001][len(g["friction"]) - 1:]])
                    if len(g["solimp"]) < 5:
                        g["solimp"] = np.concatenate([g["solimp"], [0.9, 0.95, 0.001, 0.5, 2][len(g["solimp"]):]])
                    b["geoms"].append(len(geoms))
                    geoms.append(g)
                elif ch.tag == "site":
                    at = merged("site", ch, childclass)
                    pos, quat = _frame(at)
                    s_ = dict(name=at.get("name", ""), body=bid, pos=pos, quat=quat)
                    if s_["name"]:
                        names["site"][s_["name"]] = len(sites)
                    sites.append(s_)
                elif ch.tag == "body":
                    pass
            for ch in s:
                if ch.tag == "body":
                    add_body(ch, bid, childclass)

    add_body(None, 0, None)

    # ---- geoms: mesh alignment, mass, and which to keep ----------------------
    collides = lambda g: g["contype"] != 0 or g["conaffinity"] != 0
    for g in geoms:
        in_grp = grp_lo <= g["group"] <= grp_hi
        need_mesh = g["type"] == G_MESH and (g["name"] or collides(g) or in_grp)
        g["meshobj"] = None
        if need_mesh:
            me = get_mesh(g["mesh"])
            g["meshobj"] = me
            # geom frame moves to the mesh principal frame
            g["pos"] = g["pos"] + qrot(g["quat"], me.pos)
            g["quat"] = qnorm(qmul(g["quat"], me.quat))
        g["keep"] = bool(g["name"]) or collides(g)
        vol, Iu = _geom_mass_inertia(g["type"], g["size"], g["density"], g["meshobj"]) \
            if (g["type"] != G_MESH or g["meshobj"] is not None) else (0.0, np.zeros(3))
        g["gmass"] = g["mass"] if g["mass"] is not None else g["density"] * vol
        g["ginertia"] = Iu * g["gmass"]

    # ---- body inertial properties ------------------------------------------
    for bi, b in enumerate(bodies):
        if b["inertial"] is not None and inertiafromgeom != "true":
            I = b["inertial"]
            b.update(mass=I["mass"], ipos=I["pos"], iquat=I["quat"], inertia=I["inertia"])
            continue
        gl = [geoms[g] for g in b["geoms"] if grp_lo <= geoms[g]["group"] <= grp_hi]
        if bi == 0 or not gl or inertiafromgeom == "false":
            b.update(mass=0.0, ipos=np.zeros(3), iquat=np.array([1.0, 0, 0, 0]), inertia=np.zeros(3))
            continue
        M = sum(g["gmass"] for g in gl)
        if M < MINVAL:
            b.update(mass=0.0, ipos=np.zeros(3), iquat=np.array([1.0, 0, 0, 0]), inertia=np.zeros(3))
            continue
        com = sum(g["gmass"] * g["pos"] for g in gl) / M
        It = np.zeros((3, 3))
        for g in gl:
            R = q2mat(g["quat"])
            d = g["pos"] - com
            It += R @ np.diag(g["ginertia"]) @ R.T + g["gmass"] * (d @ d * np.eye(3) - np.outer(d, d))
        ev, iq = eig3(It)
        b.update(mass=M, ipos=com, iquat=iq, inertia=ev)

    # ---- assemble arrays -----------------------------------------------------
    A = m.arrays
    nbody = len(bodies)
    A["body_parentid"] = np.array([b["parent"] for b in bodies], dtype=np.int32)
    A["body_pos"] = np.array([b["pos"] for b in bodies])
    A["body_quat"] = np.array([b["quat"] for b in bodies])
    A["body_ipos"] = np.array([b["ipos"] for b in bodies])
    A["body_iquat"] = np.array([b["iquat"] for b in bodies])
    A["body_mass"] = np.array([b["mass"] for b in bodies])
    A["body_inertia"] = np.array([b["inertia"] for b in bodies])
    A["body_mocap"] = np.array([1 if b["mocap"] else 0 for b in bodies], dtype=np.int32)

    jnt_qposadr, jnt_dofadr = [], []
    dof_body, dof_jnt, dof_parent, dof_arm, dof_damp = [], [], [], [], []
    body_dofadr = np.full(nbody, -1, dtype=np.int32)
    body_dofnum = np.zeros(nbody, dtype=np.int32)
    body_jntadr = np.full(nbody, -1, dtype=np.int32)
    body_jntnum = np.zeros(nbody, dtype=np.int32)
    body_lastdof = np.full(nbody, -1, dtype=np.int32)  # last dof of the kinematic chain ending at this body
    nq = nv = 0
    qpos0 = []
    for bi, b in enumerate(bodies):
        last = body_lastdof[b["parent"]] if bi > 0 else -1
        for ji in b["jnts"]:
            j = joints[ji]
            if body_jntadr[bi] < 0:
                body_jntadr[bi] = ji
            body_jntnum[bi] += 1
            jnt_qposadr.append(nq)
            jnt_dofadr.append(nv)
            if body_dofadr[bi] < 0:
                body_dofadr[bi] = nv
            nd = 6 if j["type"] == J_FREE else 1
            for k in range(nd):
                dof_body.append(bi)
                dof_jnt.append(ji)
                dof_parent.append(last)
                dof_arm.append(j["armature"])
                dof_damp.append(j["damping"])
                last = nv
                nv += 1
            body_dofnum[bi] += nd
            if j["type"] == J_FREE:
                qpos0 += list(b["pos"]) + list(b["quat"])
                nq += 7
            else:
                qpos0.append(0.0)
                nq += 1
        body_lastdof[bi] = last
    A["body_dofadr"], A["body_dofnum"] = body_dofadr, body_dofnum
    A["body_jntadr"], A["body_jntnum"] = body_jntadr, body_jntnum
    A["body_lastdof"] = body_lastdof
    A["qpos0"] = np.array(qpos0)
    A["jnt_type"] = np.array([j["type"] for j in joints], dtype=np.int32)
    A["jnt_bodyid"] = np.array([j["body"] for j in joints], dtype=np.int32)
    A["jnt_qposadr"] = np.array(jnt_qposadr, dtype=np.int32)
    A["jnt_dofadr"] = np.array(jnt_dofadr, dtype=np.int32)
    A["jnt_pos"] = np.array([j["pos"] for j in joints])
    A["jnt_axis"] = np.array([j["axis"] for j in joints])
    A["jnt_limited"] = np.array([1 if j["limited"] else 0 for j in joints], dtype=np.int32)
    A["jnt_range"] = np.array([j["range"] for j in joints])
    A["jnt_stiffness"] = np.array([j["stiffness"] for j in joints])
    A["jnt_springref"] = np.array([j["springref"] for j in joints])
    A["jnt_solref"] = np.array([j["solref"] for j in joints])
    A["jnt_solimp"] = np.array([j["solimp"] for j in joints])
    A["jnt_margin"] = np.array([j["margin"] for j in joints])
    A["dof_bodyid"] = np.array(dof_body, dtype=np.int32)
    A["dof_jntid"] = np.array(dof_jnt, dtype=np.int32)
    A["dof_parentid"] = np.array(dof_parent, dtype=np.int32)
    A["dof_armature"] = np.array(dof_arm)
    A["dof_damping"] = np.array(dof_damp)

    # weld ids: first ancestor-or-self with joints (0 = static/world)
    weld = np.zeros(nbody, dtype=np.int32)
    for bi in range(1, nbody):
        weld[bi] = bi if body_jntnum[bi] > 0 else weld[bodies[bi]["parent"]]
    A["body_weldid"] = weld

    kept = [g for g in geoms if g["keep"]]
    gid = {id(g): i for i, g in enumerate(kept)}
    for i, g in enumerate(kept):
        if g["name"]:
            names["geom"][g["name"]] = i
    mesh_ids = {}
    mesh_vertadr, mesh_vertnum, mesh_vert = [], [], []
    for g in kept:
        if g["type"] == G_MESH and g["meshobj"].name not in mesh_ids:
            me = g["meshobj"]
            mesh_ids[me.name] = len(mesh_ids)
            mesh_vertadr.append(sum(mesh_vertnum))
            mesh_vertnum.append(len(me.vert))
            mesh_vert.append(me.vert)
            m.meshes.append(me)
    names["mesh"] = mesh_ids
    A["mesh_vertadr"] = np.array(mesh_vertadr, dtype=np.int32)
    A["mesh_vertnum"] = np.array(mesh_vertnum, dtype=np.int32)
    A["mesh_vert"] = np.concatenate(mesh_vert) if mesh_vert else np.zeros((0, 3))
    A["geom_type"] = np.array([g["type"] for g in kept], dtype=np.int32)
    A["geom_bodyid"] = np.array([g["body"] for g in kept], dtype=np.int32)
    A["geom_meshid"] = np.array([mesh_ids[g["meshobj"].name] if g["type"] == G_MESH else -1 for g in kept],
                                dtype=np.int32)
    A["geom_size"] = np.array([g["size"] for g in kept])
    A["geom_pos"] = np.array([g["pos"] for g in kept])
    A["geom_quat"] = np.array([g["quat"] for g in kept])
    A["geom_contype"] = np.array([g["contype"] for g in kept], dtype=np.int32)
    A["geom_conaffinity"] = np.array([g["conaffinity"] for g in kept], dtype=np.int32)
    A["geom_condim"] = np.array([g["condim"] for g in kept], dtype=np.int32)
    A["geom_priority"] = np.array([g["priority"] for g in kept], dtype=np.int32)
    A["geom_friction"] = np.array([g["friction"] for g in kept])
    A["geom_solref"] = np.array([g["solref"] for g in kept])
    A["geom_solimp"] = np.array([g["solimp"] for g in kept])
    A["geom_solmix"] = np.array([g["solmix"] for g in kept])
    A["geom_margin"] = np.array([g["margin"] for g in kept])
    A["geom_gap"] = np.array([g["gap"] for g in kept])
    rb = []
    for g in kept:
        t, s = g["type"], g["size"]
        if t == G_SPHERE:
            rb.append(s[0])
        elif t == G_CAPSULE:
            rb.append(s[0] + s[1])
        elif t == G_CYLINDER:
            rb.append(np.hypot(s[0], s[1]))
        elif t in (G_BOX, G_ELLIPSOID):
            rb.append(np.linalg.norm(s) if t == G_BOX else s.max())
        elif t == G_MESH:
            rb.append(np.linalg.norm(g["meshobj"].vert, axis=1).max())
        else:
            rb.append(0.0)
    A["geom_rbound"] = np.array(rb)

    A["site_bodyid"] = np.array([s["body"] for s in sites], dtype=np.int32)
    A["site_pos"] = np.array([s["pos"] for s in sites]).reshape(-1, 3)
    A["site_quat"] = np.array([s["quat"] for s in sites]).reshape(-1, 4)

    # ---- candidate collision pairs (static filters) ---------------
    pairs = []
    ng = len(kept)
    for i in range(ng):
        for k in range(i + 1, ng):
            g1, g2 = kept[i], kept[k]
            if not ((g1["contype"] & g2["conaffinity"]) or (g2["contype"] & g1["conaffinity"])):
                continue
            w1, w2 = weld[g1["body"]], weld[g2["body"]]
            if w1 == w2:
                continue
            wp1 = weld[bodies[w1]["parent"]] if w1 else 0
            wp2 = weld[bodies[w2]["parent"]] if w2 else 0
            if w1 != 0 and w2 != 0 and (w1 == wp2 or w2 == wp1):
                continue
            a, b_ = (i, k) if g1["type"] <= g2["type"] else (k, i)
            pairs.append((a, b_))
    A["pair_geom"] = np.array(pairs, dtype=np.int32).reshape(-1, 2)

    # ---- actuators / equality --------------------------------------------------
    act_dof, act_kp, act_range = [], [], []
    for a in root.findall("actuator"):
        for p in a.findall("position"):
            at = dict(dfl.get("position", p.attrib.get("class")))
            at.update(p.attrib)
            ji = names["joint"][at["joint"]]
            assert joints[ji]["type"] in (J_SLIDE, J_HINGE)
            act_dof.append(jnt_dofadr[ji])
            act_kp.append(float(at.get("kp", 1)))
            cr = _floats(at["ctrlrange"])
            assert at.get("ctrllimited", "auto") in ("true", "auto")
            act_range.append(cr)
    A["act_dofid"] = np.array(act_dof, dtype=np.int32)
    A["act_qposid"] = np.array([jnt_qposadr[dof_jnt[d]] for d in act_dof], dtype=np.int32)
    A["act_kp"] = np.array(act_kp)
    A["act_ctrlrange"] = np.array(act_range).reshape(-1, 2)
    eq_b1, eq_b2, eq_solref, eq_solimp = [], [], [], []
    for e in root.findall("equality"):
        for w in e.findall("weld"):
            eq_b1.append(names["body"][w.attrib["body1"]])
            eq_b2.append(names["body"][w.attrib["body2"]])
            eq_solref.append(_floats(w.attrib.get("solref", "0.02 1")))
            eq_solimp.append(_floats(w.attrib.get("solimp", "0.9 0.95 0.001 0.5 2")))
    A["eq_body1"] = np.array(eq_b1, dtype=np.int32)
    A["eq_body2"] = np.array(eq_b2, dtype=np.int32)
    A["eq_solref"] = np.array(eq_solref).reshape(-1, 2)
    A["eq_solimp"] = np.array(eq_solimp).reshape(-1, 5)
    m.names = names
    _set_const(m)
    add_mesh_cells(m.arrays)
    return m


# ----------------------------------------------------------------------------
# qpos0-dependent constants (invweight0, meaninertia): numpy FK + CRB
# ----------------------------------------------------------------------------
def fk_numpy(m: Model, qpos, mocap_pos=None, mocap_quat=None):
    A = m.arrays
    nb = len(A["body_parentid"])
    xpos = np.zeros((nb, 3))
    xquat = np.zeros((nb, 4))
    xquat[0, 0] = 1
    nv = len(A["dof_bodyid"])
    xanchor = np.zeros((len(A["jnt_type"]), 3))
    xaxis = np.zeros((len(A["jnt_type"]), 3))
    for b in range(1, nb):
        p = A["body_parentid"][b]
        if A["body_mocap"][b] and mocap_pos is not None:
            pos, quat = np.asarray(mocap_pos, float), qnorm(mocap_quat)
        else:
            pos = xpos[p] + qrot(xquat[p], A["body_pos"][b])
            quat = qmul(xquat[p], A["body_quat"][b])
        for k in range(A["body_jntnum"][b]):
            j = A["body_jntadr"][b] + k
            qa = A["jnt_qposadr"][j]
            t = A["jnt_type"][j]
            if t == J_FREE:
                pos = qpos[qa:qa + 3].copy()
                quat = qnorm(qpos[qa + 3:qa + 7])
                xanchor[j] = pos
                xaxis[j] = [0, 0, 1]
            else:
                xanchor[j] = pos + qrot(quat, A["jnt_pos"][j])
                xaxis[j] = qrot(quat, A["jnt_axis"][j])
                if t == J_SLIDE:
                    pos = pos + xaxis[j] * qpos[qa]
                else:
                    h = 0.5 * qpos[qa]
                    qr = np.concatenate([[np.cos(h)], np.sin(h) * A["jnt_axis"][j]])
                    quat = qmul(quat, qr)
                    pos = xanchor[j] - qrot(quat, A["jnt_pos"][j])
        xpos[b], xquat[b] = pos, qnorm(quat)
    xipos = np.array([xpos[b] + qrot(xquat[b], A["body_ipos"][b]) for b in range(nb)])
    return xpos, xquat, xipos, xanchor, xaxis


def jac_point(m: Model, xanchor, xaxis, point, body):
    """3xnv translational and rotational Jacobian of a world point attached to `body`."""
    A = m.arrays
    nv = len(A["dof_bodyid"])
    jp, jr = np.zeros((3, nv)), np.zeros((3, nv))
    d = A["body_lastdof"][body]
    while d >= 0:
        j = A["dof_jntid"][d]
        t = A["jnt_type"][j]
        k = d - A["jnt_dofadr"][j]
        if t == J_FREE:
            if k < 3:
                jp[k, d] = 1.0
            else:
                # rotational dofs of a free joint are expressed in the body frame
                pass
        elif t == J_SLIDE:
            jp[:, d] = xaxis[j]
        else:
            jr[:, d] = xaxis[j]
            jp[:, d] = np.cross(xaxis[j], point - xanchor[j])
        d = A["dof_parentid"][d]
    return jp, jr


def mass_matrix_numpy(m: Model, qpos):
    """dense M(q) via sum_b J_b^T I_b J_b (simple, O(nb nv^2); compile-time only)."""
    A = m.arrays
    xpos, xquat, xipos, xanchor, xaxis = fk_numpy(m, qpos)
    nb = len(A["body_parentid"])
    nv = len(A["dof_bodyid"])
    M = np.zeros((nv, nv))
    J = {}
    for b in range(1, nb):
        jp, jr = jac_point(m, xanchor, xaxis, xipos[b], b)
        # free-joint rotational dofs: body-frame axes
        d = A["body_lastdof"][b]
        while d >= 0:
            j = A["dof_jntid"][d]
            if A["jnt_type"][j] == J_FREE:
                k = d - A["jnt_dofadr"][j]
                if k >= 3:
                    fb = A["jnt_bodyid"][j]
                    ax = q2mat(xquat[fb])[:, k - 3]
                    jr[:, d] = ax
                    jp[:, d] = np.cross(ax, xipos[b] - xpos[fb])
            d = A["dof_parentid"][d]
        J[b] = (jp, jr)
        mass = A["body_mass"][b]
        if mass == 0 and not A["body_inertia"][b].any():
            continue
        R = q2mat(qmul(xquat[b], A["body_iquat"][b]))
        Iw = R @ np.diag(A["body_inertia"][b]) @ R.T
        M += mass * jp.T @ jp + jr.T @ Iw @ jr
    M += np.diag(A["dof_armature"])
    return M, J


def _set_const(m: Model):
    A = m.arrays
    nv = len(A["dof_bodyid"])
    nb = len(A["body_parentid"])
    M, J = mass_matrix_numpy(m, A["qpos0"])
    Minv = np.linalg.inv(M)
    inv_b = np.zeros((nb, 2))
    for b in range(1, nb):
        if A["body_weldid"][b] == 0:
            continue
        jp, jr = J[b]
        inv_b[b, 0] = np.trace(jp @ Minv @ jp.T) / 3
        inv_b[b, 1] = np.trace(jr @ Minv @ jr.T) / 3
    inv_d = np.zeros(nv)
    for j in range(len(A["jnt_type"])):
        da = A["jnt_dofadr"][j]
        if A["jnt_type"][j] == J_FREE:
            inv_d[da:da + 3] = np.mean(np.diag(Minv)[da:da + 3])
            inv_d[da + 3:da + 6] = np.mean(np.diag(Minv)[da + 3:da + 6])
        else:
            inv_d[da] = Minv[da, da]
    A["body_invweight0"] = inv_b
    A["dof_invweight0"] = inv_d
    A["stat_meaninertia"] = np.array([np.trace(M) / max(nv, 1)])


# ----------------------------------------------------------------------------
# (de)serialisation of compiled models: the GPU box has no MJCF/STL assets, only these tables
# ----------------------------------------------------------------------------
def save_model(m: Model, path):
    import json
    meta = dict(name=m.name, opt_timestep=m.opt_timestep, opt_iterations=m.opt_iterations,
                opt_tolerance=m.opt_tolerance, gravity=list(map(float, m.gravity)), names=m.names)
    np.savez_compressed(path, __meta__=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **m.arrays)


def add_mesh_cells(A):
    """Support cells of every mesh hull (metaworld_amd/hullcells.py): per cube-map cell of directions the ascending list of hull
    vertices that can be the support vertex -- within the tie tolerance -- for some direction of the cell; CSR over
    (mesh, cell): `mesh_celladr[mesh * NCELL + cell]` .. `[+ 1]` into `mesh_cellid` (local vertex ids).  The narrow phase's hull
    support function (csrc/mw_collide.hpp) scans its direction's list instead of all vertices: the same answer as the scan that
    DEFINES the support point (oracle/mjl_collide.c support()), by construction.  Derived from the stored hull vertices."""
    from .hullcells import NCELL, support_cells
    nmesh = len(A["mesh_vertnum"])
    if "mesh_celladr" in A and len(A["mesh_celladr"]) == nmesh * NCELL + 1:
        return
    adr, ids = [0], []
    for mi in range(nmesh):
        a, n = int(A["mesh_vertadr"][mi]), int(A["mesh_vertnum"][mi])
        cadr, cid = support_cells(A["mesh_vert"][a:a + n])
        adr += [len(ids) + int(x) for x in cadr[1:]]
        ids += [int(x) for x in cid]
    A["mesh_celladr"] = np.array(adr, dtype=np.int32)
    A["mesh_cellid"] = np.array(ids if ids else [0], dtype=np.int32)
    for k in ("mesh_nbradr", "mesh_nbr", "mesh_start", "mesh_hill"):        # tables of the hill-climbing support of rounds 1-3
        A.pop(k, None)


def load_model(path) -> Model:
    import json
    z = np.load(path)
    meta = json.loads(bytes(z["__meta__"]).decode())
    m = Model(name=meta["name"], opt_timestep=meta["opt_timestep"], opt_iterations=meta["opt_iterations"],
              opt_tolerance=meta["opt_tolerance"], gravity=np.array(meta["gravity"]))
    m.names = meta["names"]
    for k in z.files:
        if k != "__meta__":
            m.arrays[k] = z[k]
    add_mesh_cells(m.arrays)
    return m
