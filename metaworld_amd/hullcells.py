"""Support cells of a convex hull: the acceleration table behind the narrow phase's hull support function
(csrc/mw_collide.hpp support(), G_MESH).

DEFINITION of a hull's support point in direction d (oracle/mjl_collide.c support(), the same for every hull size):
m = max_i v_i . d over ALL hull vertices, answer = the LOWEST vertex index whose v_i . d >= m - TIE.  No search path, no
start vertex, no history.

The table makes that definition cheap without changing it.  Directions are binned into the cells of a cube map (6 faces x
GRID x GRID, `cell_of`); for every cell the table lists, in ascending index order, every vertex that can come within EPS of the
maximum for SOME direction of the cell.  With EPS >= sqrt(3) TIE (+ rounding) the list of a direction's cell contains every
vertex within TIE of that direction's maximum, so "maximum over the list, then the first list entry within TIE of it" IS the
definition above -- by construction, for every direction, ties included.  The lists are short (a direction cell of a
few degrees sees a handful of hull vertices), which is the point: one memory round trip and ~10 dot products per support call
instead of a walk over the hull graph or a scan of 884 vertices.

How a list is built (exact, no sampling): on cube face f a direction is d(a, b) = (major = +-1, a, b), |d| in [1, sqrt 3].  Vertex v
is within TIE of the maximum for the unit direction d/|d| only if (v - w) . d(a, b) >= -sqrt(3) TIE for every other vertex w:
n - 1 half planes in (a, b).  Their intersection with the face square is a convex polygon, computed by cutting planes (clip by
the most violated half plane until none is violated); the vertex enters the list of every cell whose (slightly enlarged) square
meets the polygon.  Nothing here depends on the hull's facets, so vertices lying ON a face or an edge of the hull (they tie with
its corners and win the tie if their index is lower) are handled like any other.
"""
from __future__ import annotations

import hashlib

import numpy as np

GRID = 16                   # cells per cube-face side (csrc/mw_collide.hpp CELL_GRID)
NCELL = 6 * GRID * GRID
EPS = 5e-6                  # >= sqrt(3) * the single-precision TIE (1e-6) + rounding of the dot products (~1e-8 at 0.1 m)
CELL_SLACK = 1e-5           # a cell's square is enlarged by this much in (a, b): the device bins a direction in floating point


def cell_of(d):
    """cube-map cell of a direction, the rule of csrc/mw_collide.hpp support_cell (single-precision arithmetic in both contexts;
    the device's reciprocal is approximate, which CELL_SLACK covers)"""
    f = np.float32
    x, y, z = (f(c) for c in d)
    ax, ay, az = abs(x), abs(y), abs(z)
    if ax >= ay and ax >= az:
        axis, m, c, u, v = 0, ax, x, y, z
    elif ay >= az:
        axis, m, c, u, v = 1, ay, y, x, z
    else:
        axis, m, c, u, v = 2, az, z, x, y
    if not m > 0:
        return 0
    s = f(0.5 * GRID) * (f(1) / m)
    iu, iv = int((u + m) * s), int((v + m) * s)
    iu = min(max(iu, 0), GRID - 1)
    iv = min(max(iv, 0), GRID - 1)
    return ((2 * axis + (1 if c < 0 else 0)) * GRID + iu) * GRID + iv


def _clip(poly, na, nb, c):
    """convex polygon (k x 2) cut by the half plane na * a + nb * b + c >= 0 (Sutherland-Hodgman)"""
    val = na * poly[:, 0] + nb * poly[:, 1] + c
    keep = val >= 0
    if keep.all():
        return poly
    if not keep.any():
        return poly[:0]
    out = []
    k = len(poly)
    for i in range(k):
        j = (i + 1) % k
        if keep[i]:
            out.append(poly[i])
        if keep[i] != keep[j]:
            t = val[i] / (val[i] - val[j])
            out.append(poly[i] + t * (poly[j] - poly[i]))
    return np.array(out)


def _direction_polygon(V, vi, face, eps):
    """the (a, b) polygon of cube face `face` on which vertex vi is within eps of every other vertex's support value"""
    ax, sg = face // 2, (-1.0 if face % 2 else 1.0)
    o = [c for c in range(3) if c != ax]
    D = V[vi] - V
    c0, ca, cb = D[:, ax] * sg + eps, D[:, o[0]], D[:, o[1]]
    lim = 1.0 + CELL_SLACK
    poly = np.array([[-lim, -lim], [lim, -lim], [lim, lim], [-lim, lim]])
    for _ in range(4 * len(V) + 16):
        val = c0[:, None] + ca[:, None] * poly[None, :, 0] + cb[:, None] * poly[None, :, 1]
        worst = val.min(axis=1)
        j = int(np.argmin(worst))
        if worst[j] >= -1e-15:
            return poly
        poly = _clip(poly, ca[j], cb[j], c0[j])
        if len(poly) == 0:
            return poly
    raise RuntimeError("cutting planes did not terminate")


def _cells_of_polygon(poly):
    """cells (iu, iv) whose enlarged square meets the convex polygon"""
    w = 2.0 / GRID
    lo, hi = poly.min(0), poly.max(0)
    r = []
    for iu in range(max(0, int(np.floor((lo[0] - CELL_SLACK + 1) / w))), min(GRID - 1, int(np.floor((hi[0] + CELL_SLACK + 1) / w))) + 1):
        for iv in range(max(0, int(np.floor((lo[1] - CELL_SLACK + 1) / w))), min(GRID - 1, int(np.floor((hi[1] + CELL_SLACK + 1) / w))) + 1):
            a0, b0 = -1 + iu * w - CELL_SLACK, -1 + iv * w - CELL_SLACK
            a1, b1 = -1 + (iu + 1) * w + CELL_SLACK, -1 + (iv + 1) * w + CELL_SLACK
            p = poly
            for (na, nb, c) in ((1, 0, -a0), (-1, 0, a1), (0, 1, -b0), (0, -1, b1)):
                p = _clip(p, na, nb, c)
                if len(p) == 0:
                    break
            if len(p):
                r.append((iu, iv))
    return r


_cache: dict = {}


def support_cells(V, eps=EPS):
    """-> (adr[NCELL + 1], ids): ids[adr[c]:adr[c + 1]] = ascending vertex indices of cell c's list (never empty)"""
    V = np.ascontiguousarray(V, dtype=np.float64)
    key = hashlib.sha1(V.tobytes()).hexdigest() + repr(eps)
    if key in _cache:
        return _cache[key]
    lists = [[] for _ in range(NCELL)]
    for face in range(6):
        for vi in range(len(V)):
            poly = _direction_polygon(V, vi, face, eps)
            if len(poly) == 0:
                continue
            for (iu, iv) in _cells_of_polygon(poly):
                lists[(face * GRID + iu) * GRID + iv].append(vi)
    adr = np.zeros(NCELL + 1, dtype=np.int32)
    ids = []
    for c, lst in enumerate(lists):
        assert lst, "a direction cell without any support vertex"
        ids += sorted(lst)
        adr[c + 1] = len(ids)
    _cache[key] = (adr, np.array(ids, dtype=np.int32))
    return _cache[key]


def support_scan(V, d, tie):
    """THE DEFINITION, by exhaustive scan (tests): lowest index within tie of the maximum; dot product summed in x, y, z order"""
    dd = V[:, 0] * d[0] + V[:, 1] * d[1] + V[:, 2] * d[2]
    return int(np.flatnonzero(dd >= dd.max() - tie)[0])


def support_lookup(V, adr, ids, d, tie):
    """the table's answer for direction d (what the device computes)"""
    c = cell_of(d)
    lst = ids[adr[c]:adr[c + 1]]
    W = V[lst]
    dd = W[:, 0] * d[0] + W[:, 1] * d[1] + W[:, 2] * d[2]
    return int(lst[np.flatnonzero(dd >= dd.max() - tie)[0]])
