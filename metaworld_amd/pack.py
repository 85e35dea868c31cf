"""Lower a compiled `mjcf.Model` to the named tables the runtime's C ABI consumes
(include/mwgpu.h: mw_model_set_int / mw_model_set_real / mw_model_set_option)."""
from __future__ import annotations

import numpy as np

from .mjcf import Model

INT_FIELDS = ["body_parentid", "body_mocap", "body_jntadr", "body_jntnum", "body_lastdof", "jnt_type", "jnt_bodyid",
              "jnt_qposadr", "jnt_dofadr", "jnt_limited", "dof_bodyid", "dof_jntid", "dof_parentid", "geom_type",
              "geom_bodyid", "geom_meshid", "geom_condim", "mesh_vertadr", "mesh_vertnum", "mesh_celladr", "mesh_cellid",
              "pair_geom", "act_dofid",
              "act_qposid", "eq_body1", "eq_body2"]
REAL_FIELDS = ["body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia", "jnt_pos", "jnt_axis",
               "jnt_range", "jnt_stiffness", "jnt_springref", "jnt_solref", "jnt_solimp", "jnt_margin", "dof_armature",
               "dof_damping", "dof_invweight0", "qpos0", "geom_size", "geom_pos", "geom_quat", "geom_friction",
               "geom_solref", "geom_solimp", "geom_solmix", "geom_margin", "geom_gap", "geom_rbound", "mesh_vert",
               "act_kp", "act_ctrlrange", "eq_solref", "eq_solimp"]

# weld data written by the reference at construction (metaworld/sawyer_xyz_env.py:133-140)
REFERENCE_WELD_DATA = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -1.0, 0.0, 0.0, 0.0, 5.0]


def resolve_probe(m: Model, kind: str, name: str):
    A = m.arrays
    if kind == "body":
        return int(m.names["body"][name]), np.zeros(3), np.array([1.0, 0, 0, 0])
    if kind == "geom":
        g = m.names["geom"][name]
        return int(A["geom_bodyid"][g]), A["geom_pos"][g], A["geom_quat"][g]
    if kind == "site":
        s = m.names["site"][name]
        return int(A["site_bodyid"][s]), A["site_pos"][s], A["site_quat"][s]
    raise KeyError(kind)


def geom_aabb(m: Model):
    """[ngeom][6] = (centre, half extents) of each geom's bounding box in its own frame (mid-phase culling only)."""
    A = m.arrays
    out = np.zeros((len(A["geom_type"]), 6))
    for g, t in enumerate(A["geom_type"]):
        s = A["geom_size"][g]
        if t == 2:
            out[g, 3:] = s[0]
        elif t == 3:
            out[g, 3:] = [s[0], s[0], s[0] + s[1]]
        elif t == 5:
            out[g, 3:] = [s[0], s[0], s[1]]
        elif t in (4, 6):
            out[g, 3:] = s
        elif t == 7:
            mi = A["geom_meshid"][g]
            v = A["mesh_vert"][A["mesh_vertadr"][mi]:A["mesh_vertadr"][mi] + A["mesh_vertnum"][mi]]
            out[g, :3] = 0.5 * (v.max(0) + v.min(0))
            out[g, 3:] = 0.5 * (v.max(0) - v.min(0))
    return out


def _spatial_inertia(mass, ipos, iquat, inertia):
    """(mass, first moment m*c, 3x3 rotational inertia about the frame origin) of one body in its own frame."""
    from .mjcf import q2mat
    R = q2mat(iquat)
    Ic = R @ np.diag(inertia) @ R.T
    c = np.asarray(ipos, float)
    return mass, mass * c, Ic + mass * (c @ c * np.eye(3) - np.outer(c, c))


def _mat2quat(R):
    t = np.trace(R)
    if t > 0:
        s_ = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s_, (R[2, 1] - R[1, 2]) / s_, (R[0, 2] - R[2, 0]) / s_, (R[1, 0] - R[0, 1]) / s_])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s_ = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s_
        q[1 + i] = 0.25 * s_
        q[1 + j] = (R[j, i] + R[i, j]) / s_
        q[1 + k] = (R[k, i] + R[i, k]) / s_
    return q / np.linalg.norm(q)


def lower_model(m: Model, probes, reloc_bodies=()):
    """Compile-time fusion of jointless bodies into their nearest kept ancestor (world, bodies with joints, the mocap body,
    per-env relocatable bodies).  Kinematics/CRB/RNE loop over ~14 bodies instead of 34-40; the dynamics are identical
    (composite inertias are additive).  Geoms, probes and weld bodies are re-expressed in the kept body's frame;
    `invweight0` values stay those of the ORIGINAL bodies (they are per geom / per equality already).
    Returns (arrays dict, probe tables)."""
    from .mjcf import eig3, q2mat, qmul, qrot
    A = m.arrays
    nb = len(A["body_parentid"])
    keep = np.zeros(nb, dtype=bool)
    keep[0] = True
    keep |= A["body_jntnum"] > 0
    keep |= A["body_mocap"] > 0
    for name in reloc_bodies:
        keep[m.names["body"][name]] = True
    keep[A["eq_body1"]] = True          # weld bodies keep their own frame (the residual uses their orientation)
    keep[A["eq_body2"]] = True
    # pose of every body in the frame of its nearest kept ancestor-or-self
    anc = np.zeros(nb, dtype=int)
    rel_pos = np.zeros((nb, 3))
    rel_quat = np.tile([1.0, 0, 0, 0], (nb, 1))
    for b in range(1, nb):
        if keep[b]:
            anc[b] = b
        else:
            p = A["body_parentid"][b]
            anc[b] = anc[p]
            rel_pos[b] = rel_pos[p] + qrot(rel_quat[p], A["body_pos"][b])
            rel_quat[b] = qmul(rel_quat[p], A["body_quat"][b])
    kept = np.flatnonzero(keep)
    new_id = -np.ones(nb, dtype=int)
    new_id[kept] = np.arange(len(kept))
    out = {}
    # inertial properties: sum the spatial inertias of the merged bodies in the kept body's frame
    mass = np.zeros(len(kept)); ipos = np.zeros((len(kept), 3)); iquat = np.tile([1.0, 0, 0, 0], (len(kept), 1)); inertia = np.zeros((len(kept), 3))
    acc = {k: [0.0, np.zeros(3), np.zeros((3, 3))] for k in kept}
    for b in range(1, nb):
        if A["body_mass"][b] == 0 and not A["body_inertia"][b].any():
            continue
        R = q2mat(rel_quat[b])
        mb, hb, Jb = _spatial_inertia(A["body_mass"][b], A["body_ipos"][b], A["body_iquat"][b], A["body_inertia"][b])
        # move to the kept frame: rotate, then shift the origin by rel_pos
        c_local = hb / mb if mb > 0 else np.zeros(3)
        Jc = Jb - mb * (c_local @ c_local * np.eye(3) - np.outer(c_local, c_local))     # about the COM, body axes
        c_k = rel_pos[b] + R @ c_local
        Jk = R @ Jc @ R.T + mb * (c_k @ c_k * np.eye(3) - np.outer(c_k, c_k))
        a = acc[anc[b]]
        a[0] += mb; a[1] = a[1] + mb * c_k; a[2] = a[2] + Jk
    for k in kept:
        i = new_id[k]
        mk, hk, Jk = acc[k]
        if k == 0 or mk <= 0:
            continue
        c = hk / mk
        Jc = Jk - mk * (c @ c * np.eye(3) - np.outer(c, c))
        ev, V = np.linalg.eigh(Jc)               # full-precision principal frame (only the product R diag R^T matters)
        if np.linalg.det(V) < 0:
            V[:, 0] = -V[:, 0]
        q = _mat2quat(V)
        mass[i], ipos[i], iquat[i], inertia[i] = mk, c, q, ev
    out["body_mass"], out["body_ipos"], out["body_iquat"], out["body_inertia"] = mass, ipos, iquat, inertia
    # tree
    par = np.zeros(len(kept), dtype=np.int32)
    bpos = np.zeros((len(kept), 3)); bquat = np.tile([1.0, 0, 0, 0], (len(kept), 1))
    for k in kept[1:]:
        p = A["body_parentid"][k]
        par[new_id[k]] = new_id[anc[p]]
        bpos[new_id[k]] = rel_pos[p] + qrot(rel_quat[p], A["body_pos"][k])
        bquat[new_id[k]] = qmul(rel_quat[p], A["body_quat"][k])
    out["body_parentid"], out["body_pos"], out["body_quat"] = par, bpos, bquat
    for f in ("body_mocap", "body_jntadr", "body_jntnum", "body_lastdof"):
        out[f] = A[f][kept].astype(np.int32)
    out["jnt_bodyid"] = new_id[A["jnt_bodyid"]].astype(np.int32)
    out["dof_bodyid"] = new_id[A["dof_bodyid"]].astype(np.int32)
    # geoms
    gb = A["geom_bodyid"]
    out["geom_bodyid"] = new_id[anc[gb]].astype(np.int32)
    out["geom_pos"] = np.array([rel_pos[b] + qrot(rel_quat[b], p) for b, p in zip(gb, A["geom_pos"])]).reshape(-1, 3)
    out["geom_quat"] = np.array([qmul(rel_quat[b], q) for b, q in zip(gb, A["geom_quat"])]).reshape(-1, 4)
    out["eq_body1"] = new_id[anc[A["eq_body1"]]].astype(np.int32)
    out["eq_body2"] = new_id[anc[A["eq_body2"]]].astype(np.int32)
    eq_anchor = np.array([[rel_pos[b1], rel_pos[b2]] for b1, b2 in zip(A["eq_body1"], A["eq_body2"])]).reshape(-1, 2, 3)
    eq_relq = np.array([[rel_quat[b1], rel_quat[b2]] for b1, b2 in zip(A["eq_body1"], A["eq_body2"])]).reshape(-1, 2, 4)
    pb, pp, pq = [], [], []
    for kind, name in probes:
        b, p, q = resolve_probe(m, kind, name)
        pb.append(new_id[anc[b]]); pp.append(rel_pos[b] + qrot(rel_quat[b], p)); pq.append(qmul(rel_quat[b], q))
    relocid = np.full(len(kept), -1, dtype=np.int32)
    for i, name in enumerate(reloc_bodies):
        relocid[new_id[m.names["body"][name]]] = i
    out["body_relocid"] = relocid
    return out, (np.array(pb, dtype=np.int32), np.array(pp).reshape(-1, 3), np.array(pq).reshape(-1, 4)), (eq_anchor, eq_relq)


def pack_model(m: Model, probes, reloc_bodies=(), maxcon=64, maxefc=256, tolerance=None, iterations=None,
               ls_iterations=50, fuse_static=True):
    """probes: list of (kind, name); reloc_bodies: names of bodies whose `body_pos` is per-environment state."""
    if fuse_static:
        return _pack_lowered(m, probes, reloc_bodies, maxcon, maxefc, tolerance, iterations, ls_iterations)
    A = m.arrays
    ints = {k: np.asarray(A[k], dtype=np.int32).ravel() for k in INT_FIELDS}
    reals = {k: np.asarray(A[k], dtype=np.float64).ravel() for k in REAL_FIELDS}
    nb = len(A["body_parentid"])
    relocid = np.full(nb, -1, dtype=np.int32)
    for i, name in enumerate(reloc_bodies):
        relocid[m.names["body"][name]] = i
    ints["body_relocid"] = relocid
    reals["geom_invweight0"] = A["body_invweight0"][A["geom_bodyid"]].ravel()
    reals["geom_aabb"] = geom_aabb(m).ravel()
    eqw = A["body_invweight0"][A["eq_body1"]] + A["body_invweight0"][A["eq_body2"]]
    reals["eq_invweight0"] = eqw.ravel()
    reals["eq_data"] = np.tile(np.array(REFERENCE_WELD_DATA), len(A["eq_body1"]))
    pb, pp, pq = [], [], []
    for kind, name in probes:
        b, p, q = resolve_probe(m, kind, name)
        pb.append(b); pp.append(p); pq.append(q)
    ints["probe_body"] = np.array(pb, dtype=np.int32)
    reals["probe_pos"] = np.array(pp, dtype=np.float64).ravel()
    reals["probe_quat"] = np.array(pq, dtype=np.float64).ravel()
    options = dict(timestep=m.opt_timestep, tolerance=m.opt_tolerance if tolerance is None else tolerance,
                   reset_tolerance=m.opt_tolerance,
                   meaninertia=float(A["stat_meaninertia"][0]), gravity_z=float(m.gravity[2]),
                   iterations=m.opt_iterations if iterations is None else iterations, ls_iterations=ls_iterations,
                   maxcon=maxcon, maxefc=maxefc, nreloc=len(reloc_bodies))
    return dict(ints=ints, reals=reals, options=options)


def _pack_lowered(m, probes, reloc_bodies, maxcon, maxefc, tolerance, iterations, ls_iterations):
    A = m.arrays
    low, (pb, pp, pq), (eq_anchor, eq_relq) = lower_model(m, probes, reloc_bodies)
    ints = {k: np.asarray(low[k] if k in low else A[k], dtype=np.int32).ravel() for k in INT_FIELDS}
    reals = {k: np.asarray(low[k] if k in low else A[k], dtype=np.float64).ravel() for k in REAL_FIELDS}
    ints["body_relocid"] = low["body_relocid"]
    reals["geom_invweight0"] = A["body_invweight0"][A["geom_bodyid"]].ravel()        # ORIGINAL bodies
    reals["geom_aabb"] = geom_aabb(m).ravel()
    reals["eq_invweight0"] = (A["body_invweight0"][A["eq_body1"]] + A["body_invweight0"][A["eq_body2"]]).ravel()
    # weld data: the reference's (anchor 0, relpose identity up to sign, torquescale 5) expressed in the kept bodies:
    # point on body k = rel_pos ; orientation of the original body = q_kept * rel_quat
    data = []
    from .mjcf import qconj, qmul
    for e in range(len(A["eq_body1"])):
        a1, a2 = eq_anchor[e]
        r1, r2 = eq_relq[e]
        # residual quat conj(q2 r2) * (q1 r1) * rel = conj(r2) [conj(q2) q1] r1 * rel : valid only if r2 is identity
        assert np.allclose(r2, [1, 0, 0, 0]), "weld body2 must be a kept body or an unrotated child"
        rel = qmul(r1, np.array(REFERENCE_WELD_DATA[6:10]))
        data += list(a2) + list(a1) + list(rel) + [REFERENCE_WELD_DATA[10]]
    reals["eq_data"] = np.array(data)
    ints["probe_body"], reals["probe_pos"], reals["probe_quat"] = pb, pp.ravel(), pq.ravel()
    options = dict(timestep=m.opt_timestep, tolerance=m.opt_tolerance if tolerance is None else tolerance,
                   reset_tolerance=m.opt_tolerance,
                   meaninertia=float(A["stat_meaninertia"][0]), gravity_z=float(m.gravity[2]),
                   iterations=m.opt_iterations if iterations is None else iterations, ls_iterations=ls_iterations,
                   maxcon=maxcon, maxefc=maxefc, nreloc=len(reloc_bodies))
    return dict(ints=ints, reals=reals, options=options)
