"""The support cells of the hulls (metaworld_amd/hullcells.py, the acceleration table of the narrow phase's hull support function)
against THE DEFINITION of a hull's support point (oracle/mjl_collide.c support(): maximum over all vertices, lowest index within
TIE of it), evaluated by exhaustive scan -- for every distinct hull of the 36 scenes, on the directions where a search structure
would go wrong: face normals of the hull (whole faces tie), directions perpendicular to hull edges, directions on the borders
and corners of the direction cells, the coordinate axes, and random ones; in double and in single precision arithmetic.  Also
the fixed-place layout the runtime derives at upload (padding with the last entry, overflow batches) is restated here and
checked to select the same vertex."""
import glob
import os

import numpy as np
import pytest

from metaworld_amd import hullcells as H
from metaworld_amd.mjcf import load_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CELL_K = 8          # csrc/mw_common.hpp


def _distinct_hulls():
    seen, out = set(), []
    for p in sorted(glob.glob(os.path.join(ROOT, "metaworld_amd", "models", "*.npz"))):
        m = load_model(p)
        A = m.arrays
        for mi in range(len(A["mesh_vertnum"])):
            a, n = int(A["mesh_vertadr"][mi]), int(A["mesh_vertnum"][mi])
            V = np.ascontiguousarray(A["mesh_vert"][a:a + n])
            key = V.tobytes()
            if key in seen:
                continue
            seen.add(key)
            adr = A["mesh_celladr"][mi * H.NCELL:(mi + 1) * H.NCELL + 1]
            out.append((f"{m.name}:{mi}:{n}", V, adr - adr[0], A["mesh_cellid"][adr[0]:adr[-1]]))
    return out


HULLS = _distinct_hulls()


def _directions(V, rng):
    from scipy.spatial import ConvexHull
    hull = ConvexHull(V)
    fn = np.unique(np.round(hull.equations[:, :3], 12), axis=0)
    d = [rng.normal(size=(4000, 3)), fn, -fn, np.eye(3), -np.eye(3)]
    # perpendicular to hull edges: the edge direction crossed with random vectors
    e = V[hull.simplices[:, 0]] - V[hull.simplices[:, 1]]
    d.append(np.cross(e[:400], rng.normal(size=(min(400, len(e)), 3))))
    # borders and corners of the direction cells, on every cube face, and tiny steps off them
    g = np.linspace(-1, 1, H.GRID + 1)
    for ax in range(3):
        o = [c for c in range(3) if c != ax]
        for sg in (1.0, -1.0):
            for a in g:
                for b in rng.uniform(-1, 1, 6).tolist() + [g[3], g[-1], g[0]]:
                    for (u, v) in ((a, b), (b, a), (a + 3e-8, b), (a - 3e-8, b)):
                        x = np.zeros(3)
                        x[ax], x[o[0]], x[o[1]] = sg, u, v
                        d.append(x[None])
    d = np.concatenate(d)
    d = d[np.linalg.norm(d, axis=1) > 1e-9]
    return d / np.linalg.norm(d, axis=1)[:, None]


@pytest.mark.parametrize("name,V,adr,ids", HULLS, ids=[h[0] for h in HULLS])
def test_cell_lists_reproduce_the_scan(name, V, adr, ids):
    assert len(adr) == H.NCELL + 1 and (np.diff(adr) > 0).all()
    for c in range(H.NCELL):
        lst = ids[adr[c]:adr[c + 1]]
        assert (np.diff(lst) > 0).all() and lst[0] >= 0 and lst[-1] < len(V)        # ascending, in range
    rng = np.random.default_rng(len(V))
    dirs = _directions(V, rng)
    for dtype, tie in ((np.float64, 1e-9), (np.float32, 1e-6)):
        W = V.astype(dtype)
        bad = [(i, H.support_scan(W, d, dtype(tie)), H.support_lookup(W, adr, ids, d, dtype(tie)))
               for i, d in enumerate(dirs.astype(dtype)) if H.support_scan(W, d, dtype(tie)) != H.support_lookup(W, adr, ids, d, dtype(tie))]
        assert not bad, (dtype.__name__, len(bad), bad[:5])


def _fixed_place_lookup(V, adr, ids, d, tie):
    """csrc/mw_collide.hpp support(), G_MESH, on the layout of csrc/mw_runtime.hpp DeviceModel: CELL_K entries at the cell's fixed
    place (padded with the last entry), further batches of CELL_K in the overflow array (padded likewise)"""
    c = H.cell_of(d)
    lst = list(ids[adr[c]:adr[c + 1]])
    first = [lst[min(q, len(lst) - 1)] for q in range(CELL_K)]
    nb = (len(lst) - 1) // CELL_K if len(lst) > CELL_K else 0
    over = [lst[min(CELL_K + q, len(lst) - 1)] for q in range(nb * CELL_K)]
    dot = lambda i: V[i, 0] * d[0] + V[i, 1] * d[1] + V[i, 2] * d[2]  # noqa: E731
    m = max(dot(i) for i in first + over)
    best = None
    for i in reversed(first):
        if dot(i) >= m - tie:
            best = i
    if best is None:
        for i in over:
            if dot(i) >= m - tie:
                best = i
                break
    return best


def test_fixed_place_layout_selects_the_same_vertex():
    name, V, adr, ids = max(HULLS, key=lambda h: len(h[1]))          # the 884-vertex hull: it has the long lists
    assert (np.diff(adr) > CELL_K).any()
    rng = np.random.default_rng(1)
    dirs = _directions(V, rng)
    long_cells = set(np.flatnonzero(np.diff(adr) > CELL_K))
    hit_long = 0
    for d in dirs:
        hit_long += H.cell_of(d) in long_cells
        assert _fixed_place_lookup(V, adr, ids, d, 1e-9) == H.support_scan(V, d, 1e-9)
    assert hit_long > 50
