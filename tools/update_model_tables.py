#!/usr/bin/env python
"""Re-derive the tables that depend only on the stored hull vertices (support cells, metaworld_amd/hullcells.py) in the
committed model files metaworld_amd/models/*.npz, without recompiling the MJCF (tools/build_models.py does that and needs the
reference checkout).  Everything else in the files is left byte-for-byte as it was."""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd.mjcf import add_mesh_cells  # noqa: E402


def main():
    for p in sorted(glob.glob(os.path.join(ROOT, "metaworld_amd", "models", "*.npz"))):
        z = np.load(p)
        A = {k: z[k] for k in z.files}
        A.pop("mesh_celladr", None)
        add_mesh_cells(A)
        np.savez_compressed(p, **A)
        L = np.diff(A["mesh_celladr"])
        meta = json.loads(bytes(A["__meta__"]).decode())
        print(f"{meta['name']:36s} meshes {len(A['mesh_vertnum'])} cell lists: mean {L.mean():.2f} max {L.max()} longer than 8: {(L > 8).mean():.3f}")


if __name__ == "__main__":
    main()
