"""Lower a compiled `mjcf.Model` to the named tables the runtime's C ABI consumes
(include/mwgpu.h: mw_model_set_int / mw_model_set_real / mw_model_set_option)."""
from __future__ import annotations

import numpy as np

from .mjcf import Model

INT_FIELDS = ["body_parentid", "body_mocap", "body_jntadr", "body_jntnum", "body_lastdof", "jnt_type", "jnt_bodyid",
              "jnt_qposadr", "jnt_dofadr", "jnt_limited", "dof_bodyid", "dof_jntid", "dof_parentid", "geom_type",
              "geom_bodyid", "geom_meshid", "geom_condim", "mesh_vertadr", "mesh_vertnum", "pair_geom", "act_dofid",
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


def pack_model(m: Model, probes, reloc_bodies=(), maxcon=64, maxefc=256, tolerance=None, iterations=None,
               ls_iterations=50):
    """probes: list of (kind, name); reloc_bodies: names of bodies whose `body_pos` is per-environment state."""
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
                   meaninertia=float(A["stat_meaninertia"][0]), gravity_z=float(m.gravity[2]),
                   iterations=m.opt_iterations if iterations is None else iterations, ls_iterations=ls_iterations,
                   maxcon=maxcon, maxefc=maxefc, nreloc=len(reloc_bodies))
    return dict(ints=ints, reals=reals, options=options)
