#!/usr/bin/env python
"""Compile the 36 Meta-World MJCF scenes (reference assets, /root/reference/metaworld/assets/sawyer_xyz/*.xml)
into the flat tables committed under metaworld_amd/models/*.npz.  The GPU box has no reference checkout,
so these generated tables (hull vertices instead of STL meshes) are what ships."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd.mjcf import compile_mjcf, save_model  # noqa: E402

REF = os.environ.get("MW_REFERENCE", "/root/reference")


def used_models():
    names = set()
    for f in glob.glob(os.path.join(REF, "metaworld", "envs", "sawyer_*_v3.py")):
        for m in re.findall(r'full_V3_path_for\("sawyer_xyz/([^"]+)\.xml"\)', open(f).read()):
            names.add(m)
    return sorted(names)


def main():
    out = os.path.join(ROOT, "metaworld_amd", "models")
    os.makedirs(out, exist_ok=True)
    total = 0
    for name in used_models():
        m = compile_mjcf(os.path.join(REF, "metaworld", "assets", "sawyer_xyz", name + ".xml"))
        p = os.path.join(out, name + ".npz")
        save_model(m, p)
        total += os.path.getsize(p)
        print(f"{name:36s} nq={len(m.arrays['qpos0']):2d} nv={len(m.arrays['dof_bodyid']):2d} nbody={len(m.arrays['body_parentid']):2d} "
              f"ngeom={len(m.arrays['geom_type']):2d} npair={len(m.arrays['pair_geom']):3d} meshverts={len(m.arrays['mesh_vert'])}")
    print("total KiB", total // 1024)


if __name__ == "__main__":
    main()
