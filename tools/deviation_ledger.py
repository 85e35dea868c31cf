#!/usr/bin/env python
"""Which tasks each documented deviation from MuJoCo's narrow phase can touch (DESIGN.md 3 "deviation ledger"): for every v3 task,
the geom-type pairs of its compiled scene's static collision-pair list (the pairs the narrow phase can ever see), grouped by the
routine that handles them.  Static over-approximation: a pair in the list need not ever come into contact."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from metaworld_amd import tasks as T

PLANE, HFIELD, SPHERE, CAPSULE, ELLIPSOID, CYLINDER, BOX, MESH = range(8)
ROUTINES = {
    "capsule-box closed form (MuJoCo: mjraw_CapsuleBox, up to 2 points)": lambda a, b: (a, b) == (CAPSULE, BOX),
    "cylinder/capsule face upgrade: multi-point contact against a box face": lambda a, b: a in (CYLINDER, CAPSULE) and b == BOX,
    "box face axes before portal refinement (cylinder / hull vs box)": lambda a, b: (a in (CYLINDER, MESH) and b == BOX) or (a == BOX and b in (CYLINDER, MESH)),
    "box-box SAT + face clipping (MuJoCo: mjc_BoxBox point set)": lambda a, b: (a, b) == (BOX, BOX),
    "portal refinement, re-shot at most twice, lowest-index support ties (hull pairs)": lambda a, b: MESH in (a, b) and PLANE not in (a, b) and SPHERE not in (a, b),
    "portal refinement on primitives only (cylinder-cylinder, capsule-cylinder, ...)": lambda a, b: MESH not in (a, b) and a in (CAPSULE, CYLINDER, ELLIPSOID) and b in (CYLINDER, ELLIPSOID),
}
rows = {k: [] for k in ROUTINES}
for task in T.ALL_V3:
    m = T.compiled_model(T.TASK_CONST[task]["model"])
    A = m.arrays
    gt = np.asarray(A["geom_type"])
    pg = np.asarray(A["pair_geom"]).reshape(-1, 2)
    pairs = {(int(gt[a]), int(gt[b])) for a, b in pg}
    for k, f in ROUTINES.items():
        n = sum(1 for a, b in pg if f(int(gt[a]), int(gt[b])))
        if n:
            rows[k].append((task, n))
for k, v in rows.items():
    print(f"## {k}: {len(v)} tasks")
    print("   " + ", ".join(f"{t[:-3]} ({n})" for t, n in v))
