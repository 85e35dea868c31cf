// mw_collide.hpp -- per-lane narrow phase for the Meta-World scenes.
//
// Pair list, type ordering and parameter mixing follow MuJoCo's conventions
// (SURVEY.md Appendix B.3 / C.1 step 3): contact normal points from geom1 to
// geom2, dist < 0 is penetration, pos is midway between the surfaces, condim and
// friction take the max, solref/solimp are solmix-weighted, margin = max.
// Analytic routines: plane-X, sphere-X, capsule-capsule, box-box (SAT + face
// clipping).  Everything else (cylinders, mesh hulls, capsule-box) is one
// Minkowski-portal-refinement routine on support functions with both shapes
// inflated by margin/2.
#pragma once
#include "mw_common.hpp"
#include "mw_phys.hpp"

namespace mw {

template <typename T>
struct Shape {
    int type;
    V3<T> pos;
    M3<T> mat;
    T size[3];
    CP<T> vert;               // hull vertices (only plane_x scans them)
    int nvert;
    T margin;
    // hull (G_MESH): the support cells of this mesh (Model::mesh_cellxyz / mesh_cellovf / mesh_ovfxyz, see mw_common.hpp)
    CP<T> cellxyz, ovfxyz;
    CP<int> cellovf;
    // everything except pos / mat is a model constant of the (wave-uniform) geom pair being tested
    MW_HD Shape uniform() const {
        Shape u = *this;
        u.type = mw_uniform(type); u.nvert = mw_uniform(nvert); u.margin = mw_uniform(margin);
        u.vert = (CP<T>)mw_uniform((unsigned long long)vert);
        u.cellxyz = (CP<T>)mw_uniform((unsigned long long)cellxyz); u.ovfxyz = (CP<T>)mw_uniform((unsigned long long)ovfxyz);
        u.cellovf = (CP<int>)mw_uniform((unsigned long long)cellovf);
        for (int k = 0; k < 3; k++) u.size[k] = mw_uniform(size[k]);
        return u;
    }
};
template <typename T> struct Hit { T dist; V3<T> pos, normal; };

template <typename T>
MW_HD int hit_sphere_sphere(V3<T> c1, T r1, V3<T> c2, T r2, T margin, Hit<T>* h) {
    V3<T> d = c2 - c1;
    T len;
    V3<T> n = normalized(d, &len);
    const T dist = len - r1 - r2;
    if (dist > margin) return 0;
    h->dist = dist; h->normal = n; h->pos = c1 + n * (r1 + T(0.5) * dist);
    return 1;
}
template <typename T>
MW_HD int hit_plane_point(V3<T> n, V3<T> p0, V3<T> pt, T margin, Hit<T>* h) {
    const T dist = dot(pt - p0, n);
    if (dist > margin) return 0;
    h->dist = dist; h->normal = n; h->pos = pt - n * (T(0.5) * dist);
    return 1;
}

template <typename T>
MW_HD int plane_x(const Shape<T>& p, const Shape<T>& s, T margin, Hit<T>* h) {
    const V3<T> n = col(p.mat, 2);
    int cnt = 0;
    if (s.type == G_SPHERE) {
        const T dist = dot(s.pos - p.pos, n) - s.size[0];
        if (dist > margin) return 0;
        h->dist = dist; h->normal = n; h->pos = s.pos - n * (s.size[0] + T(0.5) * dist);
        return 1;
    }
    if (s.type == G_CAPSULE) {
        const V3<T> ax = col(s.mat, 2);
        for (int sg = -1; sg <= 1; sg += 2) {
            const V3<T> c = s.pos + ax * (T(sg) * s.size[1]);
            const T dist = dot(c - p.pos, n) - s.size[0];
            if (dist > margin) continue;
            h[cnt].dist = dist; h[cnt].normal = n; h[cnt].pos = c - n * (s.size[0] + T(0.5) * dist);
            cnt++;
        }
        return cnt;
    }
    if (s.type == G_CYLINDER) {
        V3<T> ax = col(s.mat, 2);
        T prj = dot(n, ax);
        if (prj > 0) { ax = -ax; prj = -prj; }
        V3<T> vec = ax * prj - n;
        const T len = norm(vec), r = s.size[0], hh = s.size[1];
        if (len < T(1e-12)) vec = col(s.mat, 0) * r; else vec = vec * (r / len);
        cnt += hit_plane_point(n, p.pos, s.pos + ax * hh + vec, margin, h + cnt);
        if (!cnt) return 0;
        cnt += hit_plane_point(n, p.pos, s.pos - ax * hh + vec, margin, h + cnt);
        V3<T> v1 = normalized(cross(vec, ax)) * (r * T(0.86602540378443865));
        for (int sg = -1; sg <= 1; sg += 2)
            cnt += hit_plane_point(n, p.pos, s.pos + ax * hh - vec * T(0.5) + v1 * T(sg), margin, h + cnt);
        return cnt;
    }
    if (s.type == G_BOX) {
        for (int i = 0; i < 8 && cnt < 4; i++) {
            V3<T> loc{(i & 1 ? 1 : -1) * s.size[0], (i & 2 ? 1 : -1) * s.size[1], (i & 4 ? 1 : -1) * s.size[2]};
            cnt += hit_plane_point(n, p.pos, s.pos + s.mat * loc, margin, h + cnt);
        }
        return cnt;
    }
    if (s.type == G_MESH) {
        const V3<T> nl = mulT(s.mat, n);
        const T base = dot(s.pos - p.pos, n);
        int idx[4];
        T dd[4];
        for (int i = 0; i < s.nvert; i++) {
            const T dist = base + dot(mv3(s.vert + 3 * i), nl);
            if (dist > margin) continue;
            int k = cnt < 4 ? cnt++ : -1;
            if (k < 0) {
                int worst = 0;
                for (int j = 1; j < 4; j++) if (dd[j] > dd[worst]) worst = j;
                if (dist < dd[worst]) k = worst; else continue;
            }
            idx[k] = i; dd[k] = dist;
        }
        for (int k = 0; k < cnt; k++) {
            const V3<T> pt = s.pos + s.mat * mv3(s.vert + 3 * idx[k]);
            h[k].dist = dd[k]; h[k].normal = n; h[k].pos = pt - n * (T(0.5) * dd[k]);
        }
        return cnt;
    }
    return 0;
}

template <typename T>
MW_HD int sphere_x(const Shape<T>& s, const Shape<T>& o, T margin, Hit<T>* h) {
    const T rs = s.size[0];
    if (o.type == G_SPHERE) return hit_sphere_sphere(s.pos, rs, o.pos, o.size[0], margin, h);
    if (o.type == G_CAPSULE) {
        const V3<T> ax = col(o.mat, 2);
        const T x = mw_clamp(dot(s.pos - o.pos, ax), -o.size[1], o.size[1]);
        return hit_sphere_sphere(s.pos, rs, o.pos + ax * x, o.size[0], margin, h);
    }
    if (o.type == G_CYLINDER) {
        const V3<T> ax = col(o.mat, 2), t = s.pos - o.pos;
        const T x = dot(t, ax), R = o.size[0], hh = o.size[1];
        const V3<T> perp = t - ax * x;
        const T r = norm(perp);
        if (mw_abs(x) <= hh && r >= T(1e-15) && (R - r) < (hh - mw_abs(x)))
            return hit_sphere_sphere(s.pos, rs, o.pos + ax * x, R, margin, h);
        const T sg = x >= 0 ? T(1) : T(-1);
        if (r <= R) {
            const T dist = mw_abs(x) - hh - rs;
            if (dist > margin) return 0;
            h->dist = dist; h->normal = ax * (-sg); h->pos = s.pos + h->normal * (rs + T(0.5) * dist);
            return 1;
        }
        return hit_sphere_sphere(s.pos, rs, o.pos + ax * (sg * hh) + perp * (R / r), T(0), margin, h);
    }
    if (o.type == G_BOX) {
        const V3<T> loc = mulT(o.mat, s.pos - o.pos);
        T l[3] = {loc.x, loc.y, loc.z}, cl[3];
        bool inside = true;
        for (int k = 0; k < 3; k++) { cl[k] = mw_clamp(l[k], -o.size[k], o.size[k]); inside &= cl[k] == l[k]; }
        if (!inside) return hit_sphere_sphere(s.pos, rs, o.pos + o.mat * v3(cl[0], cl[1], cl[2]), T(0), margin, h);
        int best = 0;
        T bd = T(1e30);
        for (int k = 0; k < 3; k++) { const T dd = o.size[k] - mw_abs(l[k]); if (dd < bd) { bd = dd; best = k; } }
        T nl[3] = {0, 0, 0};
        nl[best] = l[best] >= 0 ? T(-1) : T(1);
        h->normal = o.mat * v3(nl[0], nl[1], nl[2]);
        h->dist = -bd - rs;
        h->pos = s.pos + h->normal * (rs + T(0.5) * h->dist);
        return 1;
    }
    return -1;  // not analytic: caller falls back to MPR
}

// capsule - box in closed form: the capsule is its axis segment inflated by the radius, so the contact is the closest pair
// (segment point, box point) pushed out by r.  In the box frame the squared distance of p(t) = a + t d to the box,
// f(t) = sum_i max(|p_i| - s_i, 0)^2, is convex and piecewise quadratic with breakpoints where a coordinate crosses a face plane;
// f'/2 = sum_i e_i(t) d_i is piecewise linear and non-decreasing: evaluated at the sorted breakpoints, interpolated in the bracketing
// interval.  -1 when the axis segment itself touches the box (depth >= radius): the caller falls back to portal refinement.
template <typename T>
MW_HD T cb_half_slope(const T* a, const T* d, const T* s, T t) {
    T g = 0;
    for (int i = 0; i < 3; i++) {
        const T p = a[i] + t * d[i], e = p > s[i] ? p - s[i] : (p < -s[i] ? p + s[i] : T(0));
        g += e * d[i];
    }
    return g;
}
template <typename T>
MW_HD int capsule_box(const Shape<T>& c, const Shape<T>& b, T margin, Hit<T>* h) {
    const V3<T> clv = mulT(b.mat, c.pos - b.pos), ulv = mulT(b.mat, col(c.mat, 2));
    const T cl[3] = {clv.x, clv.y, clv.z}, ul[3] = {ulv.x, ulv.y, ulv.z}, hh = c.size[1];
    const T s[3] = {b.size[0], b.size[1], b.size[2]};
    T a[3], d[3], ts[8];
    for (int i = 0; i < 3; i++) { a[i] = cl[i] - hh * ul[i]; d[i] = 2 * hh * ul[i]; }
    int n = 0;
    ts[n++] = 0;
    for (int i = 0; i < 3; i++) {
        if (mw_abs(d[i]) <= T(1e-14)) continue;
        for (int sg = -1; sg <= 1; sg += 2) {
            const T t = (sg * s[i] - a[i]) / d[i];
            if (t > 0 && t < 1) ts[n++] = t;
        }
    }
    ts[n++] = 1;
    for (int i = 1; i < n; i++) {            // insertion sort
        const T v = ts[i];
        int j = i - 1;
        while (j >= 0 && ts[j] > v) { ts[j + 1] = ts[j]; j--; }
        ts[j + 1] = v;
    }
    T tstar, glo = cb_half_slope(a, d, s, T(0));
    if (glo >= 0) tstar = 0;
    else {
        tstar = 1;
        for (int k = 1; k < n; k++) {
            const T ghi = cb_half_slope(a, d, s, ts[k]);
            if (ghi >= 0) { tstar = ts[k - 1] - glo * (ts[k] - ts[k - 1]) / (ghi - glo); break; }
            glo = ghi;
        }
    }
    T pl[3], ql[3], dist2 = 0;
    for (int i = 0; i < 3; i++) {
        pl[i] = a[i] + tstar * d[i];
        ql[i] = mw_clamp(pl[i], -s[i], s[i]);
        dist2 += (pl[i] - ql[i]) * (pl[i] - ql[i]);
    }
    if (dist2 < (sizeof(T) == 8 ? T(1e-20) : T(1e-12))) return -1;
    return hit_sphere_sphere(b.pos + b.mat * v3(pl[0], pl[1], pl[2]), c.size[0], b.pos + b.mat * v3(ql[0], ql[1], ql[2]), T(0), margin, h);
}

template <typename T>
MW_HD int capsule_capsule(const Shape<T>& a, const Shape<T>& b, T margin, Hit<T>* h) {
    const V3<T> ua = col(a.mat, 2), ub = col(b.mat, 2), w = a.pos - b.pos;
    const T la = a.size[1], lb = b.size[1];
    const T ab = dot(ua, ub), aw = dot(ua, w), bw = dot(ub, w), det = 1 - ab * ab;
    if (mw_abs(det) < T(1e-10)) {
        T e0 = -aw - ab * lb, e1 = -aw + ab * lb;
        if (e0 > e1) { const T t = e0; e0 = e1; e1 = t; }
        T s0 = mw_max(-la, e0), s1 = mw_min(la, e1);
        if (s0 > s1) s0 = s1 = mw_clamp(T(0.5) * (e0 + e1), -la, la);
        const T ss[2] = {s0, s1};
        int cnt = 0;
        for (int k = 0; k < (s1 - s0 > T(1e-9) ? 2 : 1); k++) {
            const V3<T> pa = a.pos + ua * ss[k];
            const T tb = mw_clamp(dot(pa - b.pos, ub), -lb, lb);
            cnt += hit_sphere_sphere(pa, a.size[0], b.pos + ub * tb, b.size[0], margin, h + cnt);
        }
        return cnt;
    }
    T s = mw_clamp((ab * bw - aw) / det, -la, la);
    T t = bw + ab * s;
    if (t < -lb) { t = -lb; s = mw_clamp(-aw + ab * t, -la, la); }
    else if (t > lb) { t = lb; s = mw_clamp(-aw + ab * t, -la, la); }
    return hit_sphere_sphere(a.pos + ua * s, a.size[0], b.pos + ub * t, b.size[0], margin, h);
}

// Two polygons (source / destination of a clipping pass) of up to POLY_CAP vertices.  The vertex indices are run-time values, so
// plain arrays live in scratch memory, and a clipping pass is a chain of dependent scratch round trips (~40 per box-box call:
// the stage's per-branch clocks put a box-box call at 50-60 k cycles, and most waves have one in every dynamics evaluation).
// With a thread-private scratchpad slice (Env::tls) the polygons live in LDS instead; same values in the same order.
constexpr int POLY_CAP = 10;
template <typename T>
struct PolyStore {
    MW_LDS T* p;          // null: the local arrays
    int stride;
    V3<T> loc[2][16];
    MW_HD V3<T> get(int w, int i) const {
        if (p) { const int o = (w * POLY_CAP + i) * 3; return {p[o * stride], p[(o + 1) * stride], p[(o + 2) * stride]}; }
        return loc[w][i];
    }
    MW_HD void set(int w, int i, V3<T> v) {
        if (p) { const int o = (w * POLY_CAP + i) * 3; p[o * stride] = v.x; p[(o + 1) * stride] = v.y; p[(o + 2) * stride] = v.z; }
        else loc[w][i] = v;
    }
};
// keep the part of polygon `src` with pn.x <= pd; the result goes to polygon 1 - src
template <typename T>
MW_HD int clip_poly(PolyStore<T>& ps, int src, int n, V3<T> pn, T pd) {
    const int cap = ps.p ? POLY_CAP : 16;
    int mcount = 0;
    for (int i = 0; i < n; i++) {
        const V3<T> a = ps.get(src, i), b = ps.get(src, (i + 1) % n);
        const T da = dot(pn, a) - pd, db = dot(pn, b) - pd;
        if (da <= 0) ps.set(1 - src, mcount++, a);
        if ((da < 0 && db > 0) || (da > 0 && db < 0)) { ps.set(1 - src, mcount++, a + (b - a) * (da / (da - db))); }
        if (mcount >= cap - 1) break;          // (a quadrilateral clipped by four half planes has at most eight vertices)
    }
    return mcount;
}

// element k of a 3-array held in registers (k is a run-time value: a dynamically indexed ARRAY would be placed in scratch memory)
template <typename V> MW_HD V sel3(const V& a0, const V& a1, const V& a2, int k) { return k == 0 ? a0 : (k == 1 ? a1 : a2); }

template <typename T>
MW_HD int box_box(const Shape<T>& A, const Shape<T>& B, T margin, Hit<T>* h, int maxh, MW_LDS T* tls, int tls_stride) {
    // (every small array below is only ever indexed by compile-time constants after unrolling, or read through sel3: the axes,
    //  half sizes and the reference / incident box then stay in registers)
    const V3<T> axA[3] = {col(A.mat, 0), col(A.mat, 1), col(A.mat, 2)}, axB[3] = {col(B.mat, 0), col(B.mat, 1), col(B.mat, 2)};
    const T szA[3] = {A.size[0], A.size[1], A.size[2]}, szB[3] = {B.size[0], B.size[1], B.size[2]};
    const V3<T> t = B.pos - A.pos;
    T bestsep = T(-1e30);
    V3<T> bestn{1, 0, 0};
    int bestcode = -1;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const V3<T> Lx = i < 3 ? axA[i < 3 ? i : 0] : axB[i < 3 ? 0 : i - 3];
        T ra = 0, rb = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) { ra += szA[k] * mw_abs(dot(axA[k], Lx)); rb += szB[k] * mw_abs(dot(axB[k], Lx)); }
        const T tl = dot(t, Lx), sep = mw_abs(tl) - ra - rb;
        if (sep > margin) return 0;
        if (sep > bestsep) { bestsep = sep; bestcode = i; bestn = Lx * (tl >= 0 ? T(1) : T(-1)); }
    }
    T edgesep = T(-1e30);
    V3<T> edgen{1, 0, 0};
    int ei = -1, ej = -1;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            V3<T> Lx = cross(axA[i], axB[j]);
            const T len = norm(Lx);
            if (len < T(1e-6)) continue;
            Lx = Lx * (1 / len);
            T ra = 0, rb = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) { ra += szA[k] * mw_abs(dot(axA[k], Lx)); rb += szB[k] * mw_abs(dot(axB[k], Lx)); }
            const T tl = dot(t, Lx), sep = mw_abs(tl) - ra - rb;
            if (sep > margin) return 0;
            if (sep > edgesep) { edgesep = sep; ei = i; ej = j; edgen = Lx * (tl >= 0 ? T(1) : T(-1)); }
        }
    if (ei >= 0 && edgesep > bestsep + T(1e-6) + T(0.05) * mw_abs(bestsep)) {
        V3<T> pa = A.pos, pb = B.pos;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k != ei) pa = pa + axA[k] * ((dot(axA[k], edgen) > 0 ? T(1) : T(-1)) * szA[k]);
            if (k != ej) pb = pb + axB[k] * ((dot(axB[k], edgen) > 0 ? T(-1) : T(1)) * szB[k]);
        }
        const V3<T> ua = sel3(axA[0], axA[1], axA[2], ei), ub = sel3(axB[0], axB[1], axB[2], ej), w = pa - pb;
        const T sa = sel3(szA[0], szA[1], szA[2], ei), sb = sel3(szB[0], szB[1], szB[2], ej);
        const T ab = dot(ua, ub), aw = dot(ua, w), bw = dot(ub, w), det = 1 - ab * ab;
        const T s = mw_clamp((ab * bw - aw) / det, -sa, sa);
        const T u = mw_clamp((bw - ab * aw) / det, -sb, sb);
        h->dist = edgesep; h->normal = edgen; h->pos = ((pa + ua * s) + (pb + ub * u)) * T(0.5);
        return 1;
    }
    const bool refA = bestcode < 3;
    // reference (R) and incident (I) box by value
    V3<T> axR[3], axI[3];
    T szR[3], szI[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        axR[k] = refA ? axA[k] : axB[k]; axI[k] = refA ? axB[k] : axA[k];
        szR[k] = refA ? szA[k] : szB[k]; szI[k] = refA ? szB[k] : szA[k];
    }
    const V3<T> posR = refA ? A.pos : B.pos, posI = refA ? B.pos : A.pos;
    const int ra_ = bestcode % 3;
    const V3<T> n = refA ? bestn : -bestn;   // reference-face normal toward the incident box
    int ia = 0;
    T bd = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) { const T dd = mw_abs(dot(axI[k], n)); if (dd > bd) { bd = dd; ia = k; } }
    const V3<T> axIa = sel3(axI[0], axI[1], axI[2], ia);
    const T sgn = dot(axIa, n) > 0 ? T(-1) : T(1);
    const V3<T> fc = posI + axIa * (sgn * sel3(szI[0], szI[1], szI[2], ia));
    const int i1 = (ia + 1) % 3, i2 = (ia + 2) % 3;
    const V3<T> axI1 = sel3(axI[0], axI[1], axI[2], i1), axI2 = sel3(axI[0], axI[1], axI[2], i2);
    const T szI1 = sel3(szI[0], szI[1], szI[2], i1), szI2 = sel3(szI[0], szI[1], szI[2], i2);
    PolyStore<T> ps;
    ps.p = tls; ps.stride = tls_stride;
    int src = 0;
    for (int c = 0; c < 4; c++) {
        const T s1 = (c == 0 || c == 3) ? T(-1) : T(1), s2 = c < 2 ? T(-1) : T(1);
        ps.set(src, c, fc + axI1 * (s1 * szI1) + axI2 * (s2 * szI2));
    }
    int np = 4;
    const int r1 = (ra_ + 1) % 3, r2 = (ra_ + 2) % 3;
    for (int side = 0; side < 4 && np > 0; side++) {
        const int ax = side < 2 ? r1 : r2;
        const V3<T> pn = sel3(axR[0], axR[1], axR[2], ax) * ((side & 1) ? T(-1) : T(1));
        np = clip_poly(ps, src, np, pn, dot(pn, posR) + sel3(szR[0], szR[1], szR[2], ax));
        src = 1 - src;
    }
    const V3<T> rc = posR + n * sel3(szR[0], szR[1], szR[2], ra_);
    int cnt = 0;
    for (int i = 0; i < np && cnt < maxh; i++) {
        const V3<T> pt = ps.get(src, i);
        const T dist = dot(pt - rc, n);
        if (dist > margin) continue;
        h[cnt].dist = dist; h[cnt].normal = refA ? n : -n; h[cnt].pos = pt - n * (T(0.5) * dist);
        cnt++;
    }
    return cnt;
}

// cube-map cell of a direction: face = 2 * (major axis) + (negative side), (u, v) = the two other components over |major|,
// each cut into CELL_GRID intervals (metaworld_amd/hullcells.py cell_of).  Computed in SINGLE precision in both contexts (one
// v_rcp_f32 instead of a double-precision division): the lists of neighbouring cells overlap by far more than this rounding
// (hullcells.py CELL_SLACK), so a direction on a cell border gets the same support point from either cell.
template <typename T>
MW_HD int support_cell(T x_, T y_, T z_) {
    const float x = (float)x_, y = (float)y_, z = (float)z_;
    const float ax = mw_abs(x), ay = mw_abs(y), az = mw_abs(z);
    int axis;
    float m, u, v, c;
    if (ax >= ay && ax >= az) { axis = 0; m = ax; c = x; u = y; v = z; }
    else if (ay >= az) { axis = 1; m = ay; c = y; u = x; v = z; }
    else { axis = 2; m = az; c = z; u = x; v = y; }
    if (!(m > 0)) return 0;
    const float s = (0.5f * CELL_GRID) * mw_rcp_f32(m);
    int iu = (int)((u + m) * s), iv = (int)((v + m) * s);
    iu = iu < 0 ? 0 : (iu > CELL_GRID - 1 ? CELL_GRID - 1 : iu);
    iv = iv < 0 ? 0 : (iv > CELL_GRID - 1 ? CELL_GRID - 1 : iv);
    return ((2 * axis + (c < 0 ? 1 : 0)) * CELL_GRID + iu) * CELL_GRID + iv;
}

// ------------------------------------------------------------ MPR on support functions
// support point; exact ties (direction perpendicular to a flat feature) are broken canonically (towards +, lowest
// vertex index) within `tie`, so that independent implementations walk the same portal
template <typename T>
MW_HD V3<T> support(const Shape<T>& s, V3<T> dir) {
    const T tie = sizeof(T) == 8 ? T(1e-9) : T(1e-6);
    const V3<T> dl = mulT(s.mat, dir);
    V3<T> pl{0, 0, 0};
    switch (s.type) {
    case G_SPHERE: pl = dl * s.size[0]; break;
    case G_CAPSULE: pl = dl * s.size[0]; pl.z += dl.z >= -tie ? s.size[1] : -s.size[1]; break;
    case G_CYLINDER: {
        const T r = mw_sqrt(dl.x * dl.x + dl.y * dl.y);
        if (r > tie) { pl.x = dl.x / r * s.size[0]; pl.y = dl.y / r * s.size[0]; }
        pl.z = dl.z >= -tie ? s.size[1] : -s.size[1];
        break;
    }
    case G_BOX:
        pl = v3(dl.x >= -tie ? s.size[0] : -s.size[0], dl.y >= -tie ? s.size[1] : -s.size[1], dl.z >= -tie ? s.size[2] : -s.size[2]);
        break;
    case G_MESH: {
        // THE DEFINITION (oracle/mjl_collide.c support()): m = max over ALL hull vertices of v . dl, answer = the lowest vertex
        // index within `tie` of m.  The list of the direction's cell holds, in ascending index order, every vertex that can be
        // within tie of the maximum for any direction of the cell (metaworld_amd/hullcells.py), so the maximum over the list
        // and the first list entry within tie of it are that answer.  The first CELL_K entries sit at a place computed from the
        // direction alone (padded with copies of the last one, which change neither the maximum nor the first entry within
        // tie): 3 CELL_K + 1 loads issued together, one round trip.  (Rounds 1-3: a steepest-ascent walk over the hull graph,
        // 2-3 dependent round trips and ~700 instructions per call, with a path-dependent answer among tied vertices.)
        MW_COUNT(3)
        const int cell = support_cell(dl.x, dl.y, dl.z);
        const int ovf = s.cellovf[cell];
        CP<T> p = s.cellxyz + cell * (3 * CELL_K);
        T xx[CELL_K], yy[CELL_K], zz[CELL_K], dd[CELL_K];
#pragma unroll
        for (int q = 0; q < CELL_K; q++) { xx[q] = p[3 * q]; yy[q] = p[3 * q + 1]; zz[q] = p[3 * q + 2]; }
#pragma unroll
        for (int q = 0; q < CELL_K; q++) dd[q] = xx[q] * dl.x + yy[q] * dl.y + zz[q] * dl.z;
        T m = dd[0];
#pragma unroll
        for (int q = 1; q < CELL_K; q++) m = mw_max(m, dd[q]);
        if (ovf >= 0) {                                   // a long list (a few % of the cells): maximum over its further batches first
            MW_COUNT(4)
            CP<T> o = s.ovfxyz + (ovf >> 4) * (3 * CELL_K);
            for (int b = 0; b < (ovf & 15); b++)
#pragma unroll
                for (int q = 0; q < CELL_K; q++) {
                    const int k = 3 * (b * CELL_K + q);
                    m = mw_max(m, o[k] * dl.x + o[k + 1] * dl.y + o[k + 2] * dl.z);
                }
        }
        const T thr = m - tie;
        bool found = false;
        T bx = xx[CELL_K - 1], by = yy[CELL_K - 1], bz = zz[CELL_K - 1];
#pragma unroll
        for (int q = CELL_K - 1; q >= 0; q--)             // downwards, so that the FIRST entry within tie of the maximum is kept
            if (dd[q] >= thr) { bx = xx[q]; by = yy[q]; bz = zz[q]; found = true; }
        if (ovf >= 0 && !found) {
            CP<T> o = s.ovfxyz + (ovf >> 4) * (3 * CELL_K);
            for (int k = 0; k < 3 * CELL_K * (ovf & 15); k += 3) {
                const T ox = o[k], oy = o[k + 1], oz = o[k + 2];
                if (ox * dl.x + oy * dl.y + oz * dl.z >= thr) { bx = ox; by = oy; bz = oz; break; }
            }
        }
        pl = v3(bx, by, bz);
        break;
    }
    default: break;
    }
    return s.pos + s.mat * pl + dir * s.margin;
}
template <typename T> struct SV { V3<T> v, a, b; };
template <typename T>
MW_HD SV<T> msupport(const Shape<T>& A, const Shape<T>& B, V3<T> dir) {
    SV<T> o;
    o.b = support(B, dir);
    o.a = support(A, -dir);
    o.v = o.b - o.a;
    return o;
}
template <typename T>
MW_HD void tri_closest_origin(V3<T> a, V3<T> b, V3<T> c, T* w) {
    const V3<T> ab = b - a, ac = c - a, ap = -a;
    const T d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0 && d2 <= 0) { w[0] = 1; w[1] = w[2] = 0; return; }
    const V3<T> bp = -b;
    const T d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) { w[1] = 1; w[0] = w[2] = 0; return; }
    const T vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) { const T v = d1 / (d1 - d3); w[0] = 1 - v; w[1] = v; w[2] = 0; return; }
    const V3<T> cp = -c;
    const T d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) { w[2] = 1; w[0] = w[1] = 0; return; }
    const T vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) { const T v = d2 / (d2 - d6); w[0] = 1 - v; w[1] = 0; w[2] = v; return; }
    const T va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { const T v = (d4 - d3) / ((d4 - d3) + (d5 - d6)); w[0] = 0; w[1] = 1 - v; w[2] = v; return; }
    const T den = 1 / (va + vb + vc);
    w[1] = vb * den; w[2] = vc * den; w[0] = 1 - w[1] - w[2];
}
template <typename T>
MW_HD int mpr(const Shape<T>& A, const Shape<T>& B, T margin, Hit<T>* h, const V3<T>* v0_override = nullptr) {
    const T tol = sizeof(T) == 8 ? T(1e-6) : T(2e-6);          // MuJoCo's ccd_tolerance default; single precision cannot resolve less than ~2e-6
    MW_COUNT(5)
    SV<T> v0, v1, v2, v3_, v4;
    v0.a = A.pos; v0.b = B.pos; v0.v = v0.b - v0.a;
    if (v0_override) v0.v = *v0_override;
    if (norm(v0.v) < T(1e-10)) v0.v.x = T(1e-5);
    V3<T> dir = normalized(-v0.v);
    v1 = msupport(A, B, dir);
    if (dot(v1.v, dir) <= 0) return 0;
    dir = cross(v1.v, v0.v);
    if (norm(dir) < T(1e-12)) {
        T depth;
        V3<T> d_ = normalized(v1.v, &depth);
        if (depth == 0) d_ = normalized(-v0.v);   // the surfaces just touch on the centre line (v1 at the origin): the ray is the contact direction, not normalized()'s (1, 0, 0)
        h->dist = -depth + margin; h->normal = -d_; h->pos = (v1.a + v1.b) * T(0.5);
        return 1;
    }
    dir = normalized(dir);
    v2 = msupport(A, B, dir);
    if (dot(v2.v, dir) <= 0) return 0;
    dir = normalized(cross(v1.v - v0.v, v2.v - v0.v));
    if (dot(dir, v0.v) > 0) { const SV<T> tmp = v1; v1 = v2; v2 = tmp; dir = -dir; }
    for (int it = 0;; it++) {
        if (it > 100) return 0;
        v3_ = msupport(A, B, dir);
        if (dot(v3_.v, dir) <= 0) return 0;
        if (dot(cross(v1.v, v3_.v), v0.v) < 0) { v2 = v3_; dir = normalized(cross(v1.v - v0.v, v3_.v - v0.v)); continue; }
        if (dot(cross(v3_.v, v2.v), v0.v) < 0) { v1 = v3_; dir = normalized(cross(v3_.v - v0.v, v2.v - v0.v)); continue; }
        break;
    }
    bool hit = false;
    const int maxit = 200;
    for (int it = 0; it < maxit; it++) {
        T len;
        dir = normalized(cross(v2.v - v1.v, v3_.v - v1.v), &len);
        if (len == 0) break;
        if (dot(dir, v1.v) >= 0) hit = true;
        MW_COUNT(6)
        MW_PAIR_ADD(1, 1)
        v4 = msupport(A, B, dir);
        const T dv4 = dot(v4.v, dir);
        if (dv4 < 0 && !hit) return 0;
        if (dv4 - dot(v1.v, dir) <= tol || it == maxit - 1) { MW_HIST(2, it) break; }
        const V3<T> t = cross(v4.v, v0.v);
        if (dot(v1.v, t) > 0) { if (dot(v2.v, t) > 0) v1 = v4; else v3_ = v4; }
        else { if (dot(v3_.v, t) > 0) v2 = v4; else v1 = v4; }
    }
    if (!hit) return 0;
    // contact normal = portal plane normal, depth = distance of the origin to that plane; the contact point is where
    // the origin ray pierces the portal (barycentric), falling back to the closest point if degenerate
    T w[3];
    const V3<T> rd = normalized(-v0.v);
    const T denom = dot(rd, dir), depth = dot(v1.v, dir);
    bool okw = false;
    if (denom > T(1e-12)) {
        const V3<T> x = rd * (depth / denom), e1 = v2.v - v1.v, e2 = v3_.v - v1.v, ex = x - v1.v;
        const T d11 = dot(e1, e1), d12 = dot(e1, e2), d22 = dot(e2, e2), dx1 = dot(ex, e1), dx2 = dot(ex, e2);
        const T den = d11 * d22 - d12 * d12;
        if (mw_abs(den) > T(1e-30)) {
            w[1] = (d22 * dx1 - d12 * dx2) / den; w[2] = (d11 * dx2 - d12 * dx1) / den; w[0] = 1 - w[1] - w[2];
            okw = w[0] > T(-1e-6) && w[1] > T(-1e-6) && w[2] > T(-1e-6);
        }
    }
    if (!okw) tri_closest_origin(v1.v, v2.v, v3_.v, w);
    h->normal = -dir;
    h->dist = -depth + margin;
    h->pos = ((v1.a + v1.b) * w[0] + (v2.a + v2.b) * w[1] + (v3_.a + v3_.b) * w[2]) * T(0.5);
    return 1;
}

// MPR's penetration direction depends on the interior ray (centre to centre): re-shoot the ray along the normal just found
// while the depth still decreases, at most MPR_RESHOOT_MAX times (the direction error shrinks by ~depth/radius per round; for
// polytope pairs the first re-shot run confirms the face).  The cap is 2 because a wave waits for the slowest of its 64 lanes:
// with 10 rounds the narrow phase of a late-episode MT50 step cost 3-4x what it costs without any (5.3-7.8 M vs 1.3-2.3 M
// cycles per step, fp64), while the reference's scripted-policy gate gives identical per-task success counts for 2, 3 and 10.
constexpr int MPR_RESHOOT_MAX = 2;
template <typename T>
MW_HD int mpr_refined(const Shape<T>& A, const Shape<T>& B, T margin, Hit<T>* h) {
    if (!mpr(A, B, margin, h)) return 0;
    const T rel = sizeof(T) == 8 ? T(1e-10) : T(1e-6);
    for (int it = 0; it < MPR_RESHOOT_MAX; it++) {
        const T depth = margin - h->dist;
        if (depth <= T(1e-9)) break;
        const V3<T> v0 = h->normal * (T(0.02) * depth);
        Hit<T> h2;
        MW_PAIR_ADD(3, 1)
        if (!mpr(A, B, margin, &h2, &v0)) break;
        const T d2 = margin - h2.dist;
        if (d2 > depth) break;
        *h = h2;
        if (depth - d2 <= rel * depth) { MW_HIST(3, it) break; }
        if (it == MPR_RESHOOT_MAX - 1) MW_HIST(3, MPR_RESHOOT_MAX)
    }
    return 1;
}

// A cylinder / capsule touching the interior of a box face: replace the single MPR point by the multi-point
// plane-cylinder / plane-capsule contact against that face (well-conditioned resting and grasp contacts).
template <typename T>
MW_HD int face_upgrade(const Shape<T>& c, const Shape<T>& box, Hit<T>* h, T margin) {
    const V3<T> n = h[0].normal;
    int k = -1;
    T best = 0;
    for (int i = 0; i < 3; i++) {
        const T cth = dot(col(box.mat, i), n);
        if (mw_abs(cth) > mw_abs(best)) { best = cth; k = i; }
    }
    if (mw_abs(best) < 1 - T(1e-4)) return 0;
    const V3<T> nf = col(box.mat, k) * (best > 0 ? T(-1) : T(1));
    V3<T> fy = (nf.y < T(0.5) && nf.y > T(-0.5)) ? v3<T>(0, 1, 0) : v3<T>(0, 0, 1);
    fy = normalized(fy - nf * dot(nf, fy));
    const V3<T> fz = cross(nf, fy);
    Shape<T> pl;
    pl.type = G_PLANE; pl.pos = box.pos + nf * box.size[k];
    pl.mat.m[0] = fy.x; pl.mat.m[1] = fz.x; pl.mat.m[2] = nf.x;
    pl.mat.m[3] = fy.y; pl.mat.m[4] = fz.y; pl.mat.m[5] = nf.y;
    pl.mat.m[6] = fy.z; pl.mat.m[7] = fz.z; pl.mat.m[8] = nf.z;
    pl.size[0] = pl.size[1] = pl.size[2] = 0; pl.vert = nullptr; pl.nvert = 0; pl.margin = 0;
    pl.cellxyz = nullptr; pl.ovfxyz = nullptr; pl.cellovf = nullptr;
    Hit<T> t[8];
    int cnt = 0;
    V3<T> cax = col(c.mat, 2);
    const T prj = dot(nf, cax);
    if (c.type == G_CYLINDER && mw_abs(prj) > T(0.7)) {
        cnt = plane_x(pl, c, margin, t);                 // cap on the face: rim points
    } else {
        // side / capsule: the contact line between the two ends, clipped to the face rectangle
        if (prj > 0) cax = -cax;
        const T r = c.size[0], hh = c.size[1];
        V3<T> vec;
        if (c.type == G_CYLINDER) {
            vec = cax * dot(nf, cax) - nf;
            const T len = norm(vec);
            if (len < T(1e-12)) return 0;
            vec = vec * (r / len);
        } else vec = nf * (-r);
        const V3<T> s1 = c.pos + cax * hh + vec, s2 = c.pos - cax * hh + vec, dl = s2 - s1;
        T u0 = 0, u1 = 1;
        for (int j = 0; j < 3; j++) {
            if (j == k) continue;
            const V3<T> ax = col(box.mat, j);
            const T a0 = dot(s1 - box.pos, ax), da = dot(dl, ax), lim = box.size[j];
            if (mw_abs(da) < T(1e-14)) { if (mw_abs(a0) > lim) return 0; continue; }
            T ua = (-lim - a0) / da, ub = (lim - a0) / da;
            if (ua > ub) { const T tt = ua; ua = ub; ub = tt; }
            if (ua > u0) u0 = ua;
            if (ub < u1) u1 = ub;
        }
        if (u0 > u1) return 0;
        const T us[2] = {u0, u1};
        const int np = (u1 - u0) * norm(dl) > T(1e-6) ? 2 : 1;
        for (int q = 0; q < np; q++) cnt += hit_plane_point(nf, pl.pos, s1 + dl * us[q], margin, t + cnt);
    }
    int mcount = 0;
    T deepest = T(1e30);
    for (int i = 0; i < cnt; i++) {
        const V3<T> d_ = t[i].pos - box.pos;
        bool inside = true;
        for (int j = 0; j < 3; j++)
            if (j != k && mw_abs(dot(d_, col(box.mat, j))) > box.size[j] + (sizeof(T) == 8 ? T(1e-9) : T(1e-6))) inside = false;
        if (!inside) continue;
        t[mcount] = t[i];
        t[mcount].normal = -nf;
        if (t[mcount].dist < deepest) deepest = t[mcount].dist;
        mcount++;
    }
    // order-independent acceptance: some point lies on the face and none of the depth found by MPR is lost
    // (single precision: MPR resolves the depth to ~2e-6 + rounding of the poses, so the comparison gets that much slack)
    if (mcount == 0 || deepest > h[0].dist + (sizeof(T) == 8 ? T(1e-6) : T(2e-5))) return 0;
    for (int i = 0; i < mcount; i++) h[i] = t[i];
    return mcount;
}

template <typename T>
MW_HD Shape<T> make_shape(const Env<T> e, int g) {
    CModel<T>& m = e.model();
    Shape<T> s;
    s.type = m.geom_type[g];
    s.pos = ld3(e, e.lay().geom_xpos + 3 * g);
    s.mat = ld9(e, e.lay().geom_xmat + 9 * g);
    for (int k = 0; k < 3; k++) s.size[k] = m.geom_size[3 * g + k];
    s.margin = 0; s.vert = nullptr; s.nvert = 0;
    s.cellxyz = nullptr; s.ovfxyz = nullptr; s.cellovf = nullptr;
    if (s.type == G_MESH) {
        const int mi = m.geom_meshid[g];
        s.vert = m.mesh_vert + 3 * m.mesh_vertadr[mi];
        s.nvert = m.mesh_vertnum[mi];
        s.cellxyz = m.mesh_cellxyz + mi * (CELL_N * 3 * CELL_K);
        s.cellovf = m.mesh_cellovf + mi * CELL_N;
        s.ovfxyz = m.mesh_ovfxyz;
    }
    return s;
}

// A convex shape against a box, decided on the six face axes of the box when that is enough.  For the outward face normal v,
// delta_v = min_A x.v - max_box x.v is the separation along v (one support evaluation of A).  (i) max_v delta_v > margin: a
// separating axis, no contact.  (ii) otherwise, if the witness p of the best face (A's support point towards it) projects inside
// that face, at least the depth away from its edges, the face normal is the exact contact direction (delta >= 0: the box lies in
// the half space below the face and p's projection is a box point; delta < 0: the face (face - p) of the Minkowski difference
// contains the foot of the origin with an in-plane clearance >= depth, so no other supporting plane is closer than the depth).
// (iii) anything else (edges, corners, partial overlaps, an overlap deeper than the box's half thickness along that axis): -1, the
// caller runs the portal refinement.
template <typename T>
MW_HD int box_face_sat(const Shape<T>& A, const Shape<T>& box, bool box_first, T margin, Hit<T>* h) {
    T best = T(-1e30);
    V3<T> bp{0, 0, 0}, bv{0, 0, 0};
    int bk = 0;
#pragma unroll 1                             // one instance of the (inlined) support function, not six
    for (int f = 0; f < 6; f++) {            // faces in the order -x +x -y +y -z +z
        const int k = f >> 1;
        const V3<T> v = col(box.mat, k) * T((f & 1) ? 1 : -1);
        const V3<T> p = support(A, -v);
        const T delta = dot(p - box.pos, v) - box.size[k];
        if (delta > margin) return 0;          // a separating face axis: the answer is "no contact" whatever the other faces say (same result as completing the loop, up to five support evaluations fewer)
        if (delta > best) { best = delta; bk = k; bp = p; bv = v; }
    }
    const V3<T> locv = mulT(box.mat, bp - box.pos);
    const T loc[3] = {locv.x, locv.y, locv.z}, inset = best < 0 ? -best : T(0);
    // deeper than the box is thick along that axis (a thin wall or plate): the witness lies beyond the box's mid plane and the face
    // normal need not be the direction of least penetration -> the caller's portal refinement decides (same guard in the oracle)
    if (inset > box.size[bk]) return -1;
    for (int j = 0; j < 3; j++)
        if (j != bk && mw_abs(loc[j]) > box.size[j] - inset - (sizeof(T) == 8 ? T(1e-9) : T(1e-6))) return -1;
    h->dist = best;
    h->normal = box_first ? bv : -bv;
    h->pos = bp - bv * (T(0.5) * best);
    return 1;
}

// UNIFORM: every active lane of the wave tests the same geom pair (no sub-lanes), so the shapes' model constants can
// live in scalar registers; with sub-lanes each sub-lane group has its own pair and nothing is wave-uniform.
// `active`: the call itself is made by EVERY live lane of the wave (see collision()); lanes without a pair to test pass false
// and leave through a branch inside this function.
template <typename T, bool UNIFORM>
MW_STAGE_FN int collide_pair(const Shape<T>& a_, const Shape<T>& b_, T margin, Hit<T>* h, bool active, MW_LDS T* tls, int tls_stride MW_CP_EXTRA) {
    if (!active) return 0;
    const Shape<T> ua = UNIFORM ? a_.uniform() : a_, ub = UNIFORM ? b_.uniform() : b_;
    if (UNIFORM) margin = mw_uniform(margin);
    const int t1 = ua.type, t2 = ub.type;
    MW_COUNT(7)
    MW_PAIR_BEGIN(t1, t2)
    int n = -1;
    MW_CTICK(tp0)
    if (t1 == G_PLANE) n = plane_x(ua, ub, margin, h);
    else if (t1 == G_SPHERE) n = sphere_x(ua, ub, margin, h);
    else if (t1 == G_CAPSULE && t2 == G_CAPSULE) n = capsule_capsule(ua, ub, margin, h);
    MW_CTICK(tp1)
    if (t1 == G_BOX && t2 == G_BOX) { n = box_box(ua, ub, margin, h, 8, tls, tls_stride); MW_CTICK(tp2) MW_CSTAT(0, tp1, tp2) }
    MW_CTICK(tp3)
    const bool on_box = (t1 == G_CYLINDER || t1 == G_CAPSULE) && t2 == G_BOX;
    if (t1 == G_CAPSULE && t2 == G_BOX) n = capsule_box(ua, ub, margin, h);
    else if (t2 == G_BOX && (t1 == G_CYLINDER || t1 == G_MESH)) n = box_face_sat(ua, ub, false, margin, h);
    else if (t1 == G_BOX && (t2 == G_CYLINDER || t2 == G_MESH)) n = box_face_sat(ub, ua, true, margin, h);
    MW_CTICK(tp4)
    if (!(t1 == G_BOX && t2 == G_BOX)) { MW_CSTAT(2, tp0, tp1) MW_CSTAT(2, tp3, tp4) }
    if (n < 0) {
        Shape<T> a = ua, b = ub;
        a.margin = b.margin = T(0.5) * margin;
        n = mpr_refined(a, b, margin, h);
        MW_CTICK(tp5)
        MW_CSTAT(1, tp4, tp5)
    }
    if (n && on_box) {
        MW_CTICK(tp6)
        const int k = face_upgrade(a_, b_, h, margin);
        if (k) n = k;
        MW_CTICK(tp7)
        MW_CSTAT(3, tp6, tp7)
    }
    return n;
}

// conservative oriented-box test on the 6 face axes of the two geoms' local bounding boxes (never culls a touching pair)
template <typename T>
MW_HD bool obb_overlap(const Env<T> e, int g1, int g2, T margin) {
    CModel<T>& m = e.model();
    const M3<T> R1 = ld9(e, e.lay().geom_xmat + 9 * g1), R2 = ld9(e, e.lay().geom_xmat + 9 * g2);
    const T *a = m.geom_aabb + 6 * g1, *b = m.geom_aabb + 6 * g2;
    const V3<T> c1 = ld3(e, e.lay().geom_xpos + 3 * g1) + R1 * mv3(a), c2 = ld3(e, e.lay().geom_xpos + 3 * g2) + R2 * mv3(b);
    const V3<T> d = c2 - c1;
    T Rm[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            Rm[3 * i + j] = R1.m[i] * R2.m[j] + R1.m[3 + i] * R2.m[3 + j] + R1.m[6 + i] * R2.m[6 + j];   // R1^T R2
    const V3<T> t1 = mulT(R1, d);
    const T tt[3] = {t1.x, t1.y, t1.z};
    const T slack = margin + T(1e-6);
    for (int i = 0; i < 3; i++) {
        const T rb = b[3] * mw_abs(Rm[3 * i]) + b[4] * mw_abs(Rm[3 * i + 1]) + b[5] * mw_abs(Rm[3 * i + 2]);
        if (mw_abs(tt[i]) > a[3 + i] + rb + slack) return false;
    }
    for (int j = 0; j < 3; j++) {
        const T ra = a[3] * mw_abs(Rm[j]) + a[4] * mw_abs(Rm[3 + j]) + a[5] * mw_abs(Rm[6 + j]);
        const T tj = tt[0] * Rm[j] + tt[1] * Rm[3 + j] + tt[2] * Rm[6 + j];
        if (mw_abs(tj) > ra + b[3 + j] + slack) return false;
    }
    return true;
}

// broad phase (bounding spheres / plane distance) + mid phase (oriented boxes, conservative) of pair p
template <typename T>
MW_HD bool pair_near(const Env<T> e, int p) {
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const int g1 = m.pair_geom[2 * p], g2 = m.pair_geom[2 * p + 1];
    const T margin = mw_max(m.geom_margin[g1], m.geom_margin[g2]);
    const V3<T> p1 = ld3(e, L.geom_xpos + 3 * g1), p2 = ld3(e, L.geom_xpos + 3 * g2);
    if (m.geom_type[g1] != G_PLANE) {
        const T bound = m.geom_rbound[g1] + m.geom_rbound[g2] + margin;
        const V3<T> t = p1 - p2;
        return dot(t, t) <= bound * bound && obb_overlap(e, g1, g2, margin);
    }
    const V3<T> nn{e.R(L.geom_xmat + 9 * g1 + 2), e.R(L.geom_xmat + 9 * g1 + 5), e.R(L.geom_xmat + 9 * g1 + 8)};
    return dot(p2 - p1, nn) <= m.geom_rbound[g2] + margin;
}

// contact records c0 .. c0+cnt-1 (dropped beyond maxcon) for the hits of pair p
template <typename T>
MW_HD void append_contacts(const Env<T> e, int p, int cnt, const Hit<T>* hh, int c0, int maxcon) {
    CModel<T>& m = e.model();
    const int g1 = m.pair_geom[2 * p], g2 = m.pair_geom[2 * p + 1];
    const T margin = mw_max(m.geom_margin[g1], m.geom_margin[g2]);
    // mixed contact parameters (equal priorities)
    const T gap = mw_max(m.geom_gap[g1], m.geom_gap[g2]);
    const int dim = m.geom_condim[g1] > m.geom_condim[g2] ? m.geom_condim[g1] : m.geom_condim[g2];
    const T s1 = m.geom_solmix[g1], s2 = m.geom_solmix[g2];
    T mix;
    if (s1 >= T(1e-15) && s2 >= T(1e-15)) mix = s1 / (s1 + s2);
    else if (s1 < T(1e-15) && s2 < T(1e-15)) mix = T(0.5);
    else mix = s1 < T(1e-15) ? T(0) : T(1);
    T solref[2], solimp[5], fr[3];
    const T r10 = m.geom_solref[2 * g1], r20 = m.geom_solref[2 * g2];
    for (int k = 0; k < 2; k++) {
        const T a = m.geom_solref[2 * g1 + k], b = m.geom_solref[2 * g2 + k];
        solref[k] = (r10 > 0 && r20 > 0) ? mix * a + (1 - mix) * b : mw_min(a, b);
    }
    for (int k = 0; k < 5; k++) solimp[k] = mix * m.geom_solimp[5 * g1 + k] + (1 - mix) * m.geom_solimp[5 * g2 + k];
    for (int k = 0; k < 3; k++) fr[k] = mw_max(m.geom_friction[3 * g1 + k], m.geom_friction[3 * g2 + k]);
    for (int i = 0; i < cnt; i++) {
        const int c = c0 + i;
        if (c >= maxcon) break;            // contact buffer full: the rest of the list is dropped (flagged by the caller)
        // frame: normal, then the MuJoCo tangent construction
        V3<T> nx = normalized(hh[i].normal);
        V3<T> ny = (nx.y < T(0.5) && nx.y > T(-0.5)) ? v3<T>(0, 1, 0) : v3<T>(0, 0, 1);
        ny = normalized(ny - nx * dot(nx, ny));
        const V3<T> nz = cross(nx, ny);
        CON(e, c, 0) = hh[i].dist;
        CON(e, c, 1) = hh[i].pos.x; CON(e, c, 2) = hh[i].pos.y; CON(e, c, 3) = hh[i].pos.z;
        CON(e, c, 4) = nx.x; CON(e, c, 5) = nx.y; CON(e, c, 6) = nx.z;
        CON(e, c, 7) = ny.x; CON(e, c, 8) = ny.y; CON(e, c, 9) = ny.z;
        CON(e, c, 10) = nz.x; CON(e, c, 11) = nz.y; CON(e, c, 12) = nz.z;
        CON(e, c, 13) = margin - gap;
        CON(e, c, 14) = fr[0]; CON(e, c, 15) = fr[1]; CON(e, c, 16) = fr[2];
        CON(e, c, 17) = solref[0]; CON(e, c, 18) = solref[1];
        for (int k = 0; k < 5; k++) CON(e, c, 19 + k) = solimp[k];
        CON(e, c, 24) = fr[0];
        ICON(e, c, 0) = g1; ICON(e, c, 1) = g2; ICON(e, c, 2) = dim; ICON(e, c, 3) = -1;
    }
}

// Collision over the model's static pair list.
//  * no sub-lanes (nsub = 1): one sweep; every lane of the wave is at the SAME pair, so the shapes' model constants are
//    wave-uniform (collide_pair<T, true>: scalar registers, scalar branches on the geom types);
//  * sub-lanes: (1) broad + mid phase for every pair, split over the sub-lanes -> compacted, ordered candidate list;
//    (2) narrow phase over the candidates in rounds of nsub (a round is nsub real narrow-phase calls, not nsub pairs of
//    which most were culled); each sub-lane has its own pair, nothing is wave-uniform (collide_pair<T, false>).
// Hits are appended in pair order (exclusive prefix of the hit counts over the sub-lanes): the contact list is identical
// to a serial sweep whatever nsub is.
// CONVERGENT CALLS.  collide_pair is a 170 KB non-inlined function that uses every register; it is entered by ALL live lanes
// of the wave or by none: the loop / skip conditions are wave-uniform (mw_any), lanes without work pass active = false and
// branch inside the callee.  Rounds 1-2 made the call from inside the per-lane `if (near)` / `if (c0 + sub < ncand)`: values
// kept in registers across that partially-masked call came back as garbage in lanes that had sat it out.
template <typename T>
MW_STAGE_FN void collision(const Env<T> e_) {
    const Env<T> e = e_.uniform();
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const int npair = m.sz.npair, maxcon = m.sz.maxcon;
    int ncon = 0, flags = 0, want = 0;
#if defined(MW_COLL_TIMING) && defined(MW_SOLVER_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    int tstat[4] = {0, 0, 0, 0};
#endif
    if (e.nsub == 1) {
        for (int p = 0; p < npair; p++) {
            const bool near = pair_near(e, p);
            if (!mw_any(near)) continue;                       // wave-uniform: no lane of the wave is near this pair
            const int g1 = m.pair_geom[2 * p], g2 = m.pair_geom[2 * p + 1];
            const T margin = mw_max(m.geom_margin[g1], m.geom_margin[g2]);
            Hit<T> h[16];
            const int cnt = collide_pair<T, true>(make_shape(e, g1), make_shape(e, g2), margin, h, near, e.tls, e.tls_stride MW_CP_PASS(tstat));
            if (cnt <= 0) continue;
            append_contacts(e, p, cnt, h, ncon, maxcon);
            ncon += cnt;
            want += cnt;
            if (ncon > maxcon) { ncon = maxcon; flags |= ST_CON_OVERFLOW; }
        }
    } else {
        int ncand = 0;
        MW_CTICK(tc0)
        // broad + mid phase, MID_BATCH pairs per sub-lane and trip: the loads of a batch (pair -> geoms -> positions, bounds) are
        // issued together, so a sub-lane pays the two dependent round trips once per batch instead of once per pair; the
        // candidate list is written in pair order (batch slot q covers pairs p0 + q * nsub + sub)
        constexpr int MID_BATCH = 4;
        for (int p0 = 0; p0 < npair; p0 += MID_BATCH * e.nsub) {
            int f[MID_BATCH][MW_NSLOT], foff[MW_NSLOT];
            MW_SUBS(e, sub) {
                int g1[MID_BATCH], g2[MID_BATCH];
                bool ok[MID_BATCH], plane[MID_BATCH];
                T margin[MID_BATCH];
#pragma unroll
                for (int q = 0; q < MID_BATCH; q++) {
                    const int p = p0 + q * e.nsub + sub;
                    ok[q] = p < npair;
                    g1[q] = m.pair_geom[2 * (ok[q] ? p : 0)]; g2[q] = m.pair_geom[2 * (ok[q] ? p : 0) + 1];
                }
#pragma unroll
                for (int q = 0; q < MID_BATCH; q++) {
                    margin[q] = mw_max(m.geom_margin[g1[q]], m.geom_margin[g2[q]]);
                    plane[q] = m.geom_type[g1[q]] == G_PLANE;
                    const V3<T> p1 = ld3(e, L.geom_xpos + 3 * g1[q]), p2 = ld3(e, L.geom_xpos + 3 * g2[q]);
                    if (!plane[q]) {
                        const T bound = m.geom_rbound[g1[q]] + m.geom_rbound[g2[q]] + margin[q];
                        const V3<T> t = p1 - p2;
                        ok[q] = ok[q] && dot(t, t) <= bound * bound;
                    } else {
                        const V3<T> nn{e.R(L.geom_xmat + 9 * g1[q] + 2), e.R(L.geom_xmat + 9 * g1[q] + 5), e.R(L.geom_xmat + 9 * g1[q] + 8)};
                        ok[q] = ok[q] && dot(p2 - p1, nn) <= m.geom_rbound[g2[q]] + margin[q];
                    }
                }
#pragma unroll
                for (int q = 0; q < MID_BATCH; q++) {
                    if (ok[q] && !plane[q]) ok[q] = obb_overlap(e, g1[q], g2[q], margin[q]);
                    f[q][MW_SLOT(sub)] = ok[q] ? 1 : 0;
                }
            }
            for (int q = 0; q < MID_BATCH; q++) {
                const int tot = sub_scan(e, f[q], foff);
                MW_SUBS(e, sub) {
                    if (f[q][MW_SLOT(sub)]) e.I(L.ipair + ncand + foff[MW_SLOT(sub)]) = p0 + q * e.nsub + sub;
                }
                ncand += tot;
            }
        }
        MW_SYNC();
        MW_CTICK(tc1)
        MW_CTOCK(e, L, 0, tc0, tc1)
        MW_CADD(e, L, 2, ncand)
        for (int c0 = 0; mw_any(c0 < ncand); c0 += e.nsub) {   // wave-uniform trip count: the longest candidate list of the wave
            MW_CADD(e, L, 3, 1)
            Hit<T> h[MW_NSLOT][16];
            int n[MW_NSLOT], off[MW_NSLOT], pp[MW_NSLOT];
            MW_SUBS(e, sub) {
                const bool act = c0 + sub < ncand;
                const int p = act ? e.I(L.ipair + c0 + sub) : 0;           // (an idle sub-lane builds the shapes of pair 0 and discards them)
                const int g1 = m.pair_geom[2 * p], g2 = m.pair_geom[2 * p + 1];
                const T margin = mw_max(m.geom_margin[g1], m.geom_margin[g2]);
                int cnt = collide_pair<T, false>(make_shape(e, g1), make_shape(e, g2), margin, h[MW_SLOT(sub)], act, e.tls, e.tls_stride MW_CP_PASS(tstat));
                if (cnt < 0) cnt = 0;
                n[MW_SLOT(sub)] = cnt; pp[MW_SLOT(sub)] = p;
            }
            const int total = sub_scan(e, n, off);
            MW_SUBS(e, sub) {
                if (n[MW_SLOT(sub)] > 0) append_contacts(e, pp[MW_SLOT(sub)], n[MW_SLOT(sub)], h[MW_SLOT(sub)], ncon + off[MW_SLOT(sub)], maxcon);
            }
            ncon += total;
            want += total;
            if (ncon > maxcon) { ncon = maxcon; flags |= ST_CON_OVERFLOW; }
        }
        MW_CTICK(tc2)
        MW_CTOCK(e, L, 1, tc1, tc2)
    }
#if defined(MW_COLL_TIMING) && defined(MW_SOLVER_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    for (int k = 0; k < 4; k++) {          // the slowest sub-lane's share of every branch class (max over the environment's sub-lanes)
        int v = tstat[k];
        for (int off = e.lds_stride; off < 64; off <<= 1) { const int o = __shfl_xor(v, off); v = o > v ? o : v; }
        e.I(L.icount + 8 + k) += v;
    }
#endif
    // canary: ncon / want / flags are computed redundantly by every sub-lane of the environment and must agree
    if (sub_disagree(e, ncon) || sub_disagree(e, want) || sub_disagree(e, flags)) flags |= ST_DIVERGED;
    e.I(L.icount) = ncon;
    if (want > e.I(L.icount + IC_WANT_CON)) e.I(L.icount + IC_WANT_CON) = want;   // demand statistic (capacity planning)
    if (flags) e.I(L.icount + 3) |= flags;
    MW_SYNC();
}

}  // namespace mw
