// mw_phys.hpp -- per-lane rigid-body dynamics for the batched Sawyer scenes.
//
// Replaces, for one environment per lane, what the reference obtains from
// MuJoCo through `do_simulation(ctrl, 5)` -> mj_step x5 and `mj_forward`
// (reference: metaworld/sawyer_xyz_env.py:595, :620; pipeline restated in
// SURVEY.md Appendix C.1):  kinematics -> composite-rigid-body mass matrix ->
// Cholesky -> collision -> constraint rows (weld, joint limits, elliptic
// contacts) -> RNE bias / passive / position actuators -> Newton solver with
// exact line search -> semi-implicit Euler with implicit joint damping.
//
// All arrays are addressed through Env<T>::R(i) (column store, one lane = one env); the sweeps over constraint rows,
// dofs and bodies' independent components are split over the environment's sub-lanes (mw_common.hpp, MW_SUBS).
#pragma once
#include <type_traits>
#include "mw_common.hpp"

namespace mw {

template <typename T> MW_STAGE_FN void collision(const Env<T> e);  // mw_collide.hpp

// ------------------------------------------------------------------ views: an array in the scratchpad or in the column store
// (Env::chain_lds, mw_common.hpp).  LM is a compile-time switch: every stage below exists in both forms and picks one per call.
template <typename T, bool LM>
struct View {
    static constexpr bool is_lds = LM;
    Env<T> e;
    int base;          // LM: first scratchpad slot; else: first element in the column store
    MW_HD T get(int i) const {
        if (LM) return e.lds[(base + i) * e.lds_stride];
        return e.R(base + i);
    }
    MW_HD void set(int i, T v) const {
        if (LM) e.lds[(base + i) * e.lds_stride] = v;
        else e.R(base + i) = v;
    }
    MW_HD V3<T> get3(int i) const { return {get(i), get(i + 1), get(i + 2)}; }
    MW_HD Q4<T> get4(int i) const { return {get(i), get(i + 1), get(i + 2), get(i + 3)}; }
    MW_HD M3<T> get9(int i) const { M3<T> r; for (int k = 0; k < 9; k++) r.m[k] = get(i + k); return r; }
};
// dst[0 .. n) (a view) <- column-store elements src .. src + n - 1, in batches of 8 loads issued together: an element-wise loop
// waits for every load before it can store it
template <typename T, bool LM>
MW_HD void stage_in(const Env<T> e, const View<T, LM>& dst, int src, int n, int dst_stride = 1, int src_stride = 1) {
    for (int i0 = 0; i0 < n; i0 += 8) {
        T x[8];
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = e.R(src + (i0 + q < n ? i0 + q : n - 1) * src_stride);
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (i0 + q < n) dst.set((i0 + q) * dst_stride, x[q]);
    }
}
// cdof / qvel / qpos as the dynamics stages read them: the copies in front of the rows (scratchpad slots 0 .. 6 nv - 1,
// 6 nv .. 7 nv - 1, 7 nv .. 7 nv + nq - 1) or the columns
template <typename T, bool LM> MW_HD View<T, LM> cdof_view(const Env<T> e) { return View<T, LM>{e, LM ? 0 : e.lay().cdof}; }
template <typename T, bool LM> MW_HD View<T, LM> qvel_view(const Env<T> e) { return View<T, LM>{e, LM ? 6 * e.nv : e.lay().qvel}; }
template <typename T, bool LM> MW_HD View<T, LM> qpos_view(const Env<T> e) { return View<T, LM>{e, LM ? 7 * e.nv : e.lay().qpos}; }

// ------------------------------------------------------------------ kinematics
// LM: the body frames the tree walk reads back (xpos, xquat, xmat of the parent) live in the scratchpad while the walk runs;
// everything is ALSO stored to the columns, where the later stages and the task layer read it.  cdof, qvel and qpos go to their
// slots in front of the rows.
// (LM = the slots in front of the rows are in use, Env::chain_lds >= 1; FR = the body frames too, chain_lds == 2)
template <typename T, bool LM, bool FR>
MW_HD void kinematics_impl(const Env<T> e) {
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const int nb = m.sz.nbody, T0 = e.lds_perm;
    const View<T, FR> xpos{e, FR ? T0 : L.xpos}, xquat{e, FR ? T0 + 3 * nb : L.xquat}, xmat{e, FR ? T0 + 7 * nb : L.xmat};
    const View<T, LM> qpos = qpos_view<T, LM>(e), cdof = cdof_view<T, LM>(e);
    if (LM) {
        stage_in(e, qpos, L.qpos, m.sz.nq);
        stage_in(e, qvel_view<T, LM>(e), L.qvel, e.nv);
    }
    auto put = [&](const auto& v, int col, int i, T x) { v.set(i, x); if (v.is_lds) e.R(col + i) = x; };   // scratchpad copy + column
    auto put3 = [&](const auto& v, int col, int i, V3<T> x) { put(v, col, i, x.x); put(v, col, i + 1, x.y); put(v, col, i + 2, x.z); };
    put3(xpos, L.xpos, 0, v3<T>(0, 0, 0));
    { const T q0[4] = {1, 0, 0, 0}; for (int k = 0; k < 4; k++) put(xquat, L.xquat, k, q0[k]); }
    { const M3<T> R0 = q2mat(Q4<T>{1, 0, 0, 0}); for (int k = 0; k < 9; k++) put(xmat, L.xmat, k, R0.m[k]); }
    st3(e, L.xipos, v3<T>(0, 0, 0));
    for (int k = 0; k < 10; k++) e.R(L.cinert + k) = 0;
    for (int b = 1; b < nb; b++) {
        const int p = m.body_parentid[b];
        V3<T> pos;
        Q4<T> quat;
        if (m.body_mocap[b]) {
            pos = ld3(e, L.mocap);
            quat = Q4<T>{T(0.70710678118654752), 0, T(0.70710678118654752), 0};  // reference always commands [1,0,1,0]
        } else {
            const int rl = m.body_relocid[b];
            V3<T> bp = rl >= 0 ? ld3(e, L.reloc + 3 * rl) : mv3(m.body_pos + 3 * b);
            pos = xpos.get3(3 * p) + xmat.get9(9 * p) * bp;
            quat = qmul(xquat.get4(4 * p), mq4(m.body_quat + 4 * b));
        }
        const int j0 = m.body_jntadr[b], jn = m.body_jntnum[b];
        for (int k = 0; k < jn; k++) {
            const int j = j0 + k, qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j], jt = m.jnt_type[j];
            if (jt == J_FREE) {
                Q4<T> q = qnormalized(qpos.get4(qa + 3));
                put(qpos, L.qpos, qa + 3, q.w); put(qpos, L.qpos, qa + 4, q.x); put(qpos, L.qpos, qa + 5, q.y); put(qpos, L.qpos, qa + 6, q.z);
                pos = qpos.get3(qa);
                quat = q;
                M3<T> R = q2mat(quat);
                for (int c = 0; c < 3; c++) {
                    const int o = 6 * (da + c);
                    for (int r = 0; r < 6; r++) put(cdof, L.cdof, o + r, (r == 3 + c) ? T(1) : T(0));
                    V3<T> ax = col(R, c);
                    put3(cdof, L.cdof, 6 * (da + 3 + c), ax);
                    put3(cdof, L.cdof, 6 * (da + 3 + c) + 3, cross(pos, ax));
                }
            } else {
                M3<T> R = q2mat(quat);
                V3<T> anchor = pos + R * mv3(m.jnt_pos + 3 * j);
                V3<T> axis = R * mv3(m.jnt_axis + 3 * j);
                const T q = qpos.get(qa);
                if (jt == J_SLIDE) {
                    pos = pos + axis * q;
                    put3(cdof, L.cdof, 6 * da, v3<T>(0, 0, 0));
                    put3(cdof, L.cdof, 6 * da + 3, axis);
                } else {
                    const T h = T(0.5) * q, s = sin(h);
                    const T* a = m.jnt_axis + 3 * j;
                    quat = qmul(quat, Q4<T>{cos(h), s * a[0], s * a[1], s * a[2]});
                    pos = anchor - q2mat(quat) * mv3(m.jnt_pos + 3 * j);
                    put3(cdof, L.cdof, 6 * da, axis);
                    put3(cdof, L.cdof, 6 * da + 3, cross(anchor, axis));
                }
            }
        }
        quat = qnormalized(quat);
        M3<T> R = q2mat(quat);
        put3(xpos, L.xpos, 3 * b, pos);
        put(xquat, L.xquat, 4 * b, quat.w); put(xquat, L.xquat, 4 * b + 1, quat.x); put(xquat, L.xquat, 4 * b + 2, quat.y); put(xquat, L.xquat, 4 * b + 3, quat.z);
        for (int k = 0; k < 9; k++) put(xmat, L.xmat, 9 * b + k, R.m[k]);
    }
    // The tree walk above is a serial chain (replicated on the sub-lanes: each reads back what it wrote itself); the
    // per-body inertial quantities and the geom frames only depend on the finished body frames -> split over sub-lanes.
    MW_SUBS(e, sub) {
    for (int b = 1 + sub; b < nb; b += e.nsub) {
        const V3<T> pos = xpos.get3(3 * b);
        const Q4<T> quat = xquat.get4(4 * b);
        const M3<T> R = xmat.get9(9 * b);
        // inertial frame and spatial inertia about the world origin: {m, m*c, J(xx,yy,zz,xy,xz,yz)}
        V3<T> c = pos + R * mv3(m.body_ipos + 3 * b);
        st3(e, L.xipos + 3 * b, c);
        const T mass = m.body_mass[b];
        M3<T> Ri = q2mat(qmul(quat, mq4(m.body_iquat + 4 * b)));
        const T* di = m.body_inertia + 3 * b;
        T Ic[6];  // xx yy zz xy xz yz
        const int ia[6] = {0, 1, 2, 0, 0, 1}, ib[6] = {0, 1, 2, 1, 2, 2};
        for (int k = 0; k < 6; k++)
            Ic[k] = Ri.m[3 * ia[k]] * di[0] * Ri.m[3 * ib[k]] + Ri.m[3 * ia[k] + 1] * di[1] * Ri.m[3 * ib[k] + 1] +
                    Ri.m[3 * ia[k] + 2] * di[2] * Ri.m[3 * ib[k] + 2];
        const T cc = dot(c, c);
        const int o = L.cinert + 10 * b;
        e.R(o) = mass;
        e.R(o + 1) = mass * c.x; e.R(o + 2) = mass * c.y; e.R(o + 3) = mass * c.z;
        e.R(o + 4) = Ic[0] + mass * (cc - c.x * c.x);
        e.R(o + 5) = Ic[1] + mass * (cc - c.y * c.y);
        e.R(o + 6) = Ic[2] + mass * (cc - c.z * c.z);
        e.R(o + 7) = Ic[3] - mass * c.x * c.y;
        e.R(o + 8) = Ic[4] - mass * c.x * c.z;
        e.R(o + 9) = Ic[5] - mass * c.y * c.z;
    }
    for (int g = sub; g < m.sz.ngeom; g += e.nsub) {
        const int b = m.geom_bodyid[g];
        st3(e, L.geom_xpos + 3 * g, xpos.get3(3 * b) + xmat.get9(9 * b) * mv3(m.geom_pos + 3 * g));
        st9(e, L.geom_xmat + 9 * g, q2mat(qmul(xquat.get4(4 * b), mq4(m.geom_quat + 4 * g))));
    }
    }
    MW_SYNC();
}
template <typename T>
MW_STAGE_FN void kinematics(const Env<T> e_) {
    const Env<T> e = e_.uniform();
    if (e.chain_lds == 2) kinematics_impl<T, true, true>(e);
    else if (e.chain_lds == 1) kinematics_impl<T, true, false>(e);
    else kinematics_impl<T, false, false>(e);
}

// world pose of probe `p` (named body / geom / site frames the task layer reads)
template <typename T>
MW_HD V3<T> probe_pos(const Env<T> e, int p) {
    CModel<T>& m = e.model();
    const int b = m.probe_body[p];
    return ld3(e, e.lay().xpos + 3 * b) + ld9(e, e.lay().xmat + 9 * b) * mv3(m.probe_pos + 3 * p);
}
template <typename T>
MW_HD Q4<T> probe_quat(const Env<T> e, int p) {
    CModel<T>& m = e.model();
    return qmul(ld4(e, e.lay().xquat + 4 * m.probe_body[p]), mq4(m.probe_quat + 4 * p));
}

// f[6] = I10 * s[6]  (spatial inertia about origin times motion vector [w; v])
template <typename T>
MW_HD void inertia_mul(T* f, const T* I, const T* s) {
    V3<T> w{s[0], s[1], s[2]}, v{s[3], s[4], s[5]}, h{I[1], I[2], I[3]};
    V3<T> p = v * I[0] + cross(w, h);
    V3<T> t = cross(h, v);
    f[0] = I[4] * w.x + I[7] * w.y + I[8] * w.z + t.x;
    f[1] = I[7] * w.x + I[5] * w.y + I[9] * w.z + t.y;
    f[2] = I[8] * w.x + I[9] * w.y + I[6] * w.z + t.z;
    f[3] = p.x; f[4] = p.y; f[5] = p.z;
}

// ---- register-resident dense helpers -------------------------------------------------------------------------
// On the GPU a read-modify-write loop over a lane's global column serialises on memory latency (every iteration is
// a dependent round trip).  The solver's dense pieces (H assembly, Cholesky, triangular solves, J'f, M x) therefore
// run on compile-time-sized register arrays with every loop fully unrolled; the routines are instantiated for the
// exact nv of the scenes (MW_NV_DISPATCH; a size bound NV > nv also works: rows/columns >= nv are identity/zero padding).
constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

template <typename T, int NV>
MW_HD void vec_load(const Env<T> e, int off, int n, T* x) {
#pragma unroll
    for (int k = 0; k < NV; k++) { const T v = e.R(off + (k < n ? k : 0)); x[k] = k < n ? v : T(0); }   // branch-free
}
template <typename T, int NV>
MW_HD void vec_store(const Env<T> e, int off, int n, const T* x) {
#pragma unroll
    for (int k = 0; k < NV; k++)
        if (k < n) e.R(off + k) = x[k];
}
// lower triangle of the row-major n x n matrix at A -> packed registers, identity padded
template <typename T, int NV>
MW_HD void tri_load(const Env<T> e, int A, int n, T* h) {
#pragma unroll
    for (int i = 0; i < NV; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            const T v = e.R(A + (i < n ? i * n + j : 0));
            h[tri(i, j)] = i < n ? v : (i == j ? T(1) : T(0));
        }
}
template <typename T, int NV>
MW_HD void tri_store(const Env<T> e, int A, int n, const T* h) {
#pragma unroll
    for (int i = 0; i < NV; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
            if (i < n) e.R(A + i * n + j) = h[tri(i, j)];
}
// (the reciprocals of the diagonal are kept so that the 136 + 34 divisions of a factor + solve become multiplications:
//  a division is ~10 instructions in fp32 and more in fp64, the Cholesky runs 4-5 times per dynamics evaluation)
template <typename T, int NV>
MW_HD void chol_reg(T* h, T* inv) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) {
            T s = h[tri(i, j)];
#pragma unroll
            for (int k = 0; k < j; k++) s -= h[tri(i, k)] * h[tri(j, k)];
            if (i == j) { h[tri(i, i)] = mw_sqrt(s < T(1e-15) ? T(1e-15) : s); inv[i] = T(1) / h[tri(i, i)]; }
            else h[tri(i, j)] = s * inv[j];
        }
    }
}
template <typename T, int NV>
MW_HD void chol_solve_reg(const T* h, const T* inv, T* x) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
        T s = x[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= h[tri(i, k)] * x[k];
        x[i] = s * inv[i];
    }
#pragma unroll
    for (int i = NV - 1; i >= 0; i--) {
        T s = x[i];
#pragma unroll
        for (int k = i + 1; k < NV; k++) s -= h[tri(k, i)] * x[k];
        x[i] = s * inv[i];
    }
}
// column dst <- M x (M the full row-major n x n matrix at A, x zero beyond n), the rows split over the environment's sub-lanes
// (each row's sum in index order: identical values for any split), then visible to all of them.  The fully unrolled register form (mat_vec: nv^2 loads at nv^2 hoisted 64-bit addresses,
// which the compiler kept in AGPRs / scratch and fetched one load per wait) cost 880 of the 970 kcycles of this phase per step.
template <typename T, int NV>
MW_HD void mat_vec_rows(const Env<T> e, int A, int n, const T* x, int dst) {
    MW_SUBS(e, sub) {
        for (int k = sub; k < n; k += e.nsub) {
            T s = 0;
#pragma unroll
            for (int j = 0; j < NV; j++) s += e.R(A + (j < n ? k * n + j : 0)) * x[j];
            e.R(dst + k) = s;
        }
    }
    MW_SYNC();
}
// factor the n x n matrix at A in place / solve with the factor at A, through registers
template <typename T, int NV>
MW_HD void chol_factor_via_reg(const Env<T> e, int A, int Lout, int n) {
    T h[NV * (NV + 1) / 2], inv[NV];
    tri_load<T, NV>(e, A, n, h);
    chol_reg<T, NV>(h, inv);
    tri_store<T, NV>(e, Lout, n, h);
}
template <typename T, int NV>
MW_HD void chol_solve_via_reg(const Env<T> e, int A, int x, int n) {
    T h[NV * (NV + 1) / 2], v[NV];
    tri_load<T, NV>(e, A, n, h);
    vec_load<T, NV>(e, x, n, v);
    T inv[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) inv[i] = T(1) / h[tri(i, i)];
    chol_solve_reg<T, NV>(h, inv, v);
    vec_store<T, NV>(e, x, n, v);
}
template <typename T, int NV>
MW_HD void chol_factor_solve_via_reg(const Env<T> e, int A, int x, int n) {   // the factor is not written back
    T h[NV * (NV + 1) / 2], v[NV];
    tri_load<T, NV>(e, A, n, h);
    vec_load<T, NV>(e, x, n, v);
    T inv[NV];
    chol_reg<T, NV>(h, inv);
    chol_solve_reg<T, NV>(h, inv, v);
    vec_store<T, NV>(e, x, n, v);
}
// nv of the 36 scenes is 10, 11, 15, 16 or 17: the register-resident routines are instantiated for exactly those sizes
// (H for nv = 15 is 120 registers instead of 153: together with four Jacobian rows it still fits the 256 architectural
// VGPRs, which a 17-padded H does not -- accumulating in AccVGPRs costs two extra moves per FMA)
#define MW_NV_DISPATCH(nv, CALL)                                                              \
    switch (nv) {                                                                             \
    case 10: { constexpr int NVC = 10; CALL; break; }                                         \
    case 11: { constexpr int NVC = 11; CALL; break; }                                         \
    case 15: { constexpr int NVC = 15; CALL; break; }                                         \
    case 16: { constexpr int NVC = 16; CALL; break; }                                         \
    default: { constexpr int NVC = 17; CALL; break; }                                         \
    }

// ------------------------------------------------------------------ mass matrix
// LM: the composite inertias are accumulated in the scratchpad (the leaf-to-root sum is a chain of read-modify-writes per
// component) and the rows of M read them and cdof from there; only M itself goes to the column store.
template <typename T, bool LM>
MW_HD void crb_impl(const Env<T> e) {
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const int nb = m.sz.nbody, nv = m.sz.nv;
    const View<T, LM> crbv{e, LM ? e.lds_perm : L.crb}, cdof = cdof_view<T, LM>(e);
    MW_SUBS(e, sub) {
        // composite inertias: the leaf-to-root accumulation is a serial chain per component, the 10 components are
        // independent -> one (or two) per sub-lane; same order of additions as a serial sweep
        for (int k = sub; k < 10; k += e.nsub) {
            stage_in(e, View<T, LM>{e, crbv.base + k}, L.cinert + k, nb, 10, 10);          // crb[b][k] <- cinert[b][k] for all b
            for (int b = nb - 1; b > 0; b--) {
                const int p = m.body_parentid[b];
                if (p > 0) crbv.set(10 * p + k, crbv.get(10 * p + k) + crbv.get(10 * b + k));
            }
        }
        for (int i = sub; i < nv * nv; i += e.nsub) e.R(L.qM + i) = 0;
    }
    MW_SYNC();
    MW_SUBS(e, sub) {
        for (int i = sub; i < nv; i += e.nsub) {          // row / column i of M: one dof per sub-lane
            T I[10], s[6], f[6];
            const int b = m.dof_bodyid[i];
            for (int k = 0; k < 10; k++) I[k] = crbv.get(10 * b + k);
            for (int k = 0; k < 6; k++) s[k] = cdof.get(6 * i + k);
            inertia_mul(f, I, s);
            for (int j = i; j >= 0; j = m.dof_parentid[j]) {
                T v = 0;
                for (int k = 0; k < 6; k++) v += cdof.get(6 * j + k) * f[k];
                if (j == i) v += m.dof_armature[i];
                e.R(L.qM + i * nv + j) = v;
                e.R(L.qM + j * nv + i) = v;
            }
        }
    }
    MW_SYNC();
    // Cholesky factor of M (lower triangle) into qL, through registers (replicated on the sub-lanes)
    MW_NV_DISPATCH(nv, (chol_factor_via_reg<T, NVC>(e, L.qM, L.qL, nv)))
}
template <typename T>
MW_STAGE_FN void crb(const Env<T> e_) {
    const Env<T> e = e_.uniform();
    if (e.chain_lds) crb_impl<T, true>(e);
    else crb_impl<T, false>(e);
}

// ------------------------------------------------------------------ bias forces (RNE), passive, actuation
template <typename T>
MW_HD void cross_motion(T* r, const T* v, const T* s) {
    V3<T> vw{v[0], v[1], v[2]}, vl{v[3], v[4], v[5]}, sw{s[0], s[1], s[2]}, sl{s[3], s[4], s[5]};
    V3<T> a = cross(vw, sw), b = cross(vw, sl) + cross(vl, sw);
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = b.x; r[4] = b.y; r[5] = b.z;
}
// end of smooth_forces: qfrc_smooth += position actuators; qacc_smooth = M^-1 qfrc_smooth through the factor of M.  All loads
// first, then arithmetic and stores (see integrate_impl: an element-wise read-modify-write or copy loop over the column store is
// one memory round trip per element).  The actuated dof is a wave-uniform model constant: the addition is selected per
// compile-time register index, not by indexing the register array.
template <typename T, int NV>
MW_HD void smooth_tail(const Env<T> e) {
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const int nv = e.nv;
    T sm[NV], h[NV * (NV + 1) / 2], inv[NV], add[2] = {0, 0};
    int dof[2] = {-1, -1};
    vec_load<T, NV>(e, L.smooth, nv, sm);
    tri_load<T, NV>(e, L.qL, nv, h);
    const int nu = m.sz.nu < 2 ? m.sz.nu : 2;          // (ModelData::finalize insists on the two finger actuators)
    for (int u = 0; u < nu; u++) {
        const T c = mw_clamp(e.R(L.ctrl + u), m.act_ctrlrange[2 * u], m.act_ctrlrange[2 * u + 1]);
        add[u] = m.act_kp[u] * (c - e.R(L.qpos + m.act_qposid[u]));
        dof[u] = m.act_dofid[u];
    }
#pragma unroll
    for (int k = 0; k < NV; k++) {
        if (k == dof[0]) sm[k] += add[0];
        if (k == dof[1]) sm[k] += add[1];
    }
    vec_store<T, NV>(e, L.smooth, nv, sm);
#pragma unroll
    for (int i = 0; i < NV; i++) inv[i] = T(1) / h[tri(i, i)];
    chol_solve_reg<T, NV>(h, inv, sm);
    vec_store<T, NV>(e, L.qacc_smooth, nv, sm);
}

// LM: body velocities / accelerations / forces (every body reads its parent's, then the forces are summed leaf to root) stay in
// the scratchpad and are never written to the column store; cdof and qvel come from their slots in front of the rows.
template <typename T, bool LM, bool FR>
MW_HD void smooth_forces_impl(const Env<T> e) {
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const int nb = m.sz.nbody, nv = m.sz.nv, T0 = e.lds_perm;
    const View<T, FR> cvel{e, FR ? T0 : L.cvel}, cacc{e, FR ? T0 + 6 * nb : L.cacc}, cfrc{e, FR ? T0 + 12 * nb : L.cfrc};
    const View<T, LM> cdof = cdof_view<T, LM>(e), qvel = qvel_view<T, LM>(e), qpos = qpos_view<T, LM>(e);
    for (int k = 0; k < 6; k++) { cvel.set(k, T(0)); cfrc.set(k, T(0)); }
    cacc.set(0, T(0)); cacc.set(1, T(0)); cacc.set(2, T(0));
    cacc.set(3, -m.gravity[0]); cacc.set(4, -m.gravity[1]); cacc.set(5, -m.gravity[2]);
    for (int b = 1; b < nb; b++) {
        const int p = m.body_parentid[b];
        T v[6], a[6];
        for (int k = 0; k < 6; k++) { v[k] = cvel.get(6 * p + k); a[k] = cacc.get(6 * p + k); }
        const int j0 = m.body_jntadr[b], jn = m.body_jntnum[b];
        for (int jj = 0; jj < jn; jj++) {
            const int j = j0 + jj, da = m.jnt_dofadr[j];
            if (m.jnt_type[j] == J_FREE) {
                for (int i = 0; i < 3; i++) {
                    const T qd = qvel.get(da + i);
                    for (int c = 0; c < 6; c++) v[c] += cdof.get(6 * (da + i) + c) * qd;
                }
                T vs[6];
                for (int c = 0; c < 6; c++) vs[c] = v[c];
                for (int i = 3; i < 6; i++) {
                    T s[6], sd[6];
                    for (int c = 0; c < 6; c++) s[c] = cdof.get(6 * (da + i) + c);
                    cross_motion(sd, vs, s);
                    const T qd = qvel.get(da + i);
                    for (int c = 0; c < 6; c++) { v[c] += s[c] * qd; a[c] += sd[c] * qd; }
                }
            } else {
                T s[6], sd[6];
                for (int c = 0; c < 6; c++) s[c] = cdof.get(6 * da + c);
                cross_motion(sd, v, s);
                const T qd = qvel.get(da);
                for (int c = 0; c < 6; c++) { v[c] += s[c] * qd; a[c] += sd[c] * qd; }
            }
        }
        T I[10], Ia[6], Iv[6];
        for (int k = 0; k < 10; k++) I[k] = e.R(L.cinert + 10 * b + k);
        inertia_mul(Ia, I, a);
        inertia_mul(Iv, I, v);
        // v x* (I v)
        V3<T> vw{v[0], v[1], v[2]}, vl{v[3], v[4], v[5]}, fn{Iv[0], Iv[1], Iv[2]}, ff{Iv[3], Iv[4], Iv[5]};
        V3<T> tn = cross(vw, fn) + cross(vl, ff), tf = cross(vw, ff);
        for (int k = 0; k < 6; k++) { cvel.set(6 * b + k, v[k]); cacc.set(6 * b + k, a[k]); }
        cfrc.set(6 * b, Ia[0] + tn.x); cfrc.set(6 * b + 1, Ia[1] + tn.y); cfrc.set(6 * b + 2, Ia[2] + tn.z);
        cfrc.set(6 * b + 3, Ia[3] + tf.x); cfrc.set(6 * b + 4, Ia[4] + tf.y); cfrc.set(6 * b + 5, Ia[5] + tf.z);
    }
    MW_SYNC();
    MW_SUBS(e, sub) {
        for (int k = sub; k < 6; k += e.nsub)          // leaf-to-root force accumulation: one component per sub-lane
            for (int b = nb - 1; b > 0; b--) {
                const int p = m.body_parentid[b];
                if (p > 0) cfrc.set(6 * p + k, cfrc.get(6 * p + k) + cfrc.get(6 * b + k));
            }
    }
    MW_SYNC();
    // qfrc_smooth = passive - bias + actuator ; qacc_smooth = M^-1 qfrc_smooth
    MW_SUBS(e, sub)
    for (int i = sub; i < nv; i += e.nsub) {
        T bias = 0;
        const int b = m.dof_bodyid[i];
        for (int k = 0; k < 6; k++) bias += cdof.get(6 * i + k) * cfrc.get(6 * b + k);
        e.R(L.bias + i) = bias;
        T f = -m.dof_damping[i] * qvel.get(i) - bias;
        const int j = m.dof_jntid[i];
        if (m.jnt_type[j] != J_FREE && m.jnt_stiffness[j] != 0)
            f -= m.jnt_stiffness[j] * (qpos.get(m.jnt_qposadr[j]) - m.jnt_springref[j]);
        e.R(L.smooth + i) = f;
    }
    MW_SYNC();
    MW_NV_DISPATCH(nv, (smooth_tail<T, NVC>(e)))
}
template <typename T>
MW_STAGE_FN void smooth_forces(const Env<T> e_) {
    const Env<T> e = e_.uniform();
    if (e.chain_lds == 2) smooth_forces_impl<T, true, true>(e);
    else if (e.chain_lds == 1) smooth_forces_impl<T, true, false>(e);
    else smooth_forces_impl<T, false, false>(e);
}

// ------------------------------------------------------------------ constraint rows
// A constraint row = its Jacobian (nv entries) + the scalars the solver sweeps need.  Rows live in the WORKGROUP SCRATCHPAD
// (LDS on the device) while there is room -- row r of the environment in lane l: slots (r * (SR_N + nv) + field) * lpb + l,
// fields 0 .. SR_N-1 the scalars (enum SR_*), SR_N + i the Jacobian entry of dof i -- and in the column store beyond that
// (efcJ[row][nv], efcX[row][EFC_EXTRA]).  The solver sweeps every row ~10-30 times per dynamics evaluation; from LDS a row is
// one ~100-cycle round trip, from the column store (L2 / HBM) 300-900, and the rows no longer travel to HBM and back once per
// evaluation (they were the largest part of the kernel's write traffic).
// efcX per row: 3 D, 4 aref, 5 force, 6 jar, 7 Jv, 8 friction scale, 9 row descriptor, 10 state (0-2 unused)
template <typename T> MW_HD GRef<T> EX(const Env<T> e, int row, int k) { return e.R(e.o_efcX + EFC_EXTRA * row + k); }
template <typename T> MW_HD GRef<T> EJ(const Env<T> e, int row, int i) { return e.R(e.o_efcJ + row * e.nv + i); }
template <typename T> MW_HD GRef<T> CON(const Env<T> e, int c, int k) { return e.R(e.o_con + CON_STRIDE * c + k); }
// contact record: 0 dist, 1-3 pos, 4-12 frame, 13 includemargin, 14-16 friction(slide,torsion,roll), 17-18 solref, 19-23 solimp, 24 mu
template <typename T> MW_HD GRef<int> ICON(const Env<T> e, int c, int k) { return e.I(e.o_icon + CON_ISTRIDE * c + k); }  // g1,g2,dim,efc_address
template <typename T> MW_HD GRef<int> IEFC(const Env<T> e, int r, int k) { return e.I(e.o_iefc + EFC_ISTRIDE * r + k); }  // 0 type, 1 id, 2 first row of block r (rows beyond the scratchpad)

// Solver row scalars: `info` = type + 16 * dim + 256 * k (k-th row of a dim-row cone block); the rows of a block are visited
// from its first row with a wave-uniform counter.
enum { SR_D = 0, SR_JAR, SR_JV, SR_FRI, SR_INFO, SR_STATE, SR_FORCE, SR_BLK, SR_AREF };   // (AREF is read by the warm start only, JV is rewritten after every Newton direction: between the two, solve_wave parks the cone blocks' scalars in them)
MW_HD constexpr int sr_slot(int f) { return f == SR_D ? 3 : f == SR_JAR ? 6 : f == SR_JV ? 7 : f == SR_FRI ? 8 : f == SR_INFO ? 9 : f == SR_STATE ? 10 : f == SR_AREF ? 4 : 5; }
template <typename T>
MW_HD T sr_get(const Env<T> e, int i, int f) {
    if (i < e.lds_rows) return e.lds[e.S(i, f) * e.lds_stride];
    return EX(e, i, sr_slot(f));
}
template <typename T>
MW_HD void sr_set(const Env<T> e, int i, int f, T v) {
    if (i < e.lds_rows) e.lds[e.S(i, f) * e.lds_stride] = v;
    else EX(e, i, sr_slot(f)) = v;
}
template <typename T>
MW_HD void ej_set(const Env<T> e, int row, int i, T v) {
    if (row < e.lds_rows) e.lds[e.S(row, SR_N + i) * e.lds_stride] = v;
    else EJ(e, row, i) = v;
}
template <typename T>
MW_HD T ej_get(const Env<T> e, int row, int i) {
    if (row < e.lds_rows) return e.lds[e.S(row, SR_N + i) * e.lds_stride];
    return EJ(e, row, i);
}
// first row of constraint block k (list built by make_constraints)
template <typename T>
MW_HD int block_row(const Env<T> e, int k) {
    if (k < e.lds_rows) return (int)e.lds[e.S(k, SR_BLK) * e.lds_stride];
    return IEFC(e, k, 2);
}
template <typename T>
MW_HD void set_block_row(const Env<T> e, int k, int row) {
    if (k < e.lds_rows) e.lds[e.S(k, SR_BLK) * e.lds_stride] = T(row);
    else IEFC(e, k, 2) = row;
}

// dense load of constraint row `row` of J into registers (rows are zero outside their dof range; entries >= nv re-read
// entry 0 and are never used -- no per-element branches, so the loads issue back to back)
template <typename T, typename HT, int NV>
MW_HD void jrow_load_as(const Env<T> e, int row, int nv, HT* j) {
    if (row < e.lds_rows) {
        const int base = e.S(row, SR_N);
#pragma unroll
        for (int k = 0; k < NV; k++) j[k] = (HT)e.lds[(base + (k < nv ? k : 0)) * e.lds_stride];
    } else {
#pragma unroll
        for (int k = 0; k < NV; k++) j[k] = (HT)EJ(e, row, k < nv ? k : 0);
    }
}
template <typename T, int NV>
MW_HD void jrow_load(const Env<T> e, int row, int nv, T* j) { jrow_load_as<T, T, NV>(e, row, nv, j); }

template <typename T, typename P>
MW_HD T impedance(P solimp, T x) {
    T d0 = mw_clamp(solimp[0], T(0.0001), T(0.9999)), dw = mw_clamp(solimp[1], T(0.0001), T(0.9999));
    T width = solimp[2], mid = mw_clamp(solimp[3], T(0.0001), T(0.9999)), power = mw_max(solimp[4], T(1));
    if (width < T(1e-15) || d0 == dw) return T(0.5) * (d0 + dw);
    x = mw_abs(x) / width;
    if (x >= 1) return dw;
    if (x <= 0) return d0;
    T y;
    if (power == 1) y = x;
    else if (power == 2) y = x <= mid ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);
    else y = x <= mid ? T(pow(double(x / mid), double(power))) * mid : 1 - T(pow(double((1 - x) / (1 - mid)), double(power))) * (1 - mid);
    return d0 + y * (dw - d0);
}

// The scalars of a row whose Jacobian is written: regulariser R -> D = 1 / R and the reference acceleration, from the row's
// residual r = pos - margin and velocity vel = J qvel (accumulated by the caller while it fills the row, dof by dof in ascending
// order: the sum the row-by-row form took from the stored row); Jv starts at 0 (it is read, times alpha = 0, before the first
// search direction exists).
template <typename T, typename P1, typename P2>
MW_HD void finish_row(const Env<T> e, int row, P1 solref, P2 solimp, T diagApprox, T r, T vel, T fri, T info, T* Rout, T* Bout) {
    CModel<T>& m = e.model();
    T tc = solref[0], dr = solref[1];
    if (tc > 0) tc = mw_max(tc, 2 * m.timestep);
    const T dmax = mw_clamp(solimp[1], T(0.0001), T(0.9999));
    const T K = 1 / mw_max(T(1e-15), dmax * dmax * tc * tc * dr * dr), B = 2 / mw_max(T(1e-15), dmax * tc);
    const T imp = impedance(solimp, r);
    const T R = mw_max(T(1e-15), (1 - imp) / imp * diagApprox);
    sr_set(e, row, SR_D, 1 / R); sr_set(e, row, SR_AREF, -B * vel - K * imp * r);
    sr_set(e, row, SR_FRI, fri); sr_set(e, row, SR_INFO, info); sr_set(e, row, SR_JV, T(0));
    if (Rout) { *Rout = R; *Bout = B; }
}

// Jacobian rows are filled DOF BY DOF: body_dofmask[b] (derived at upload) says which dofs lie on body b's chain, so the motion
// axes cdof[i] are loaded once per dof at an address that does not depend on a previous load, and every entry
// J[row][i] = sign_a * (axis . jac_a) + sign_b * (axis . jac_b) is stored once (zero off the chains).  The two terms are added in
// the order the reference-style accumulation used (first body of the call order first), so the values are the same.
// (Round 6 measured the six rows of the weld as six work items of one row each -- the shared prologue and the dof loop repeated per
//  item, one finish_row instead of six: 1.4 % SLOWER at MT50 @ 4096, the wave runs the weld / limit / contact bodies one after the other
//  whatever the split, and the longer work list costs light scenes a second round.  One item per constraint stays.)
template <typename T, bool LM>
MW_HD void weld_rows(const Env<T> e, int q, int r0) {
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const int b1 = m.eq_body1[q], b2 = m.eq_body2[q];
    const T* data = m.eq_data + 11 * q;
    V3<T> p1 = ld3(e, L.xpos + 3 * b1) + ld9(e, L.xmat + 9 * b1) * mv3(data + 3);
    V3<T> p2 = ld3(e, L.xpos + 3 * b2) + ld9(e, L.xmat + 9 * b2) * mv3(data);
    V3<T> cp = p1 - p2;
    const T ts = data[10];
    Q4<T> qa = qmul(ld4(e, L.xquat + 4 * b1), mq4(data + 6));
    Q4<T> q2n = qconj(ld4(e, L.xquat + 4 * b2));
    Q4<T> qr = qmul(q2n, qa);
    // translational rows k: + axis_k . (v + w x p1) of body 1, - axis_k . (v + w x p2) of body 2
    // rotational rows: 0.5 * imag( conj(q2) * (w1 - w2) * q1 * rel ) * torquescale
    const int m1 = m.body_dofmask[b1], m2 = m.body_dofmask[b2], both = m1 | m2;
    T vel[6] = {0, 0, 0, 0, 0, 0};
    const View<T, LM> cdof = cdof_view<T, LM>(e), qvel = qvel_view<T, LM>(e);
    for (int i = 0; i < e.nv; i++) {
        if (!((both >> i) & 1)) {
            for (int k = 0; k < 6; k++) ej_set(e, r0 + k, i, T(0));
            continue;
        }
        const V3<T> w = cdof.get3(6 * i), v = cdof.get3(6 * i + 3);
        const T qd = qvel.get(i);
        const bool in1 = (m1 >> i) & 1, in2 = (m2 >> i) & 1;
        const V3<T> l1 = v + cross(w, p1), l2 = v + cross(w, p2);
        const Q4<T> q4 = qmul(qmul(q2n, Q4<T>{0, w.x, w.y, w.z}), qa);
        const T rot[3] = {q4.x, q4.y, q4.z};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            V3<T> ax{T(k == 0), T(k == 1), T(k == 2)};
            T acc = 0, acr = 0;
            if (in1) { acc += T(1) * dot(ax, l1); acr += T(1) * T(0.5) * rot[k] * ts; }
            if (in2) { acc += T(-1) * dot(ax, l2); acr += T(-1) * T(0.5) * rot[k] * ts; }
            ej_set(e, r0 + k, i, acc);
            ej_set(e, r0 + 3 + k, i, acr);
            vel[k] += acc * qd; vel[3 + k] += acr * qd;
        }
    }
    const T res[6] = {cp.x, cp.y, cp.z, ts * qr.x, ts * qr.y, ts * qr.z};
#pragma unroll
    for (int k = 0; k < 6; k++) {          // (unrolled: res[k] / vel[k] stay in registers)
        IEFC(e, r0 + k, 0) = C_EQUALITY; IEFC(e, r0 + k, 1) = q;
        finish_row(e, r0 + k, m.eq_solref + 2 * q, m.eq_solimp + 5 * q, m.eq_invweight0[2 * q + (k >= 3)], res[k], vel[k], T(0), T(C_EQUALITY + 16),
                   (T*)nullptr, (T*)nullptr);
    }
}

template <typename T, bool LM>
MW_HD void limit_row(const Env<T> e, int id, int r) {
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const int j = id >> 1, side = (id & 1) ? 1 : -1, dof = m.jnt_dofadr[j];
    const T q = qpos_view<T, LM>(e).get(m.jnt_qposadr[j]), margin = m.jnt_margin[j];
    const T dist = side * (m.jnt_range[2 * j + (side + 1) / 2] - q);
    for (int i = 0; i < e.nv; i++) ej_set(e, r, i, i == dof ? T(-side) : T(0));
    IEFC(e, r, 0) = C_LIMIT; IEFC(e, r, 1) = id;
    T vel = 0;
    vel += T(-side) * qvel_view<T, LM>(e).get(dof);
    finish_row(e, r, m.jnt_solref + 2 * j, m.jnt_solimp + 5 * j, m.dof_invweight0[dof], dist - margin, vel, T(0), T(C_LIMIT + 16), (T*)nullptr, (T*)nullptr);
}

template <typename T, bool LM>
MW_HD void contact_rows(const Env<T> e, int c, int r0) {
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const T dist = CON(e, c, 0), inc = CON(e, c, 13);
    const int g1 = ICON(e, c, 0), g2 = ICON(e, c, 1), dim = ICON(e, c, 2);
    const int b1 = m.geom_bodyid[g1], b2 = m.geom_bodyid[g2];
    V3<T> pos{CON(e, c, 1), CON(e, c, 2), CON(e, c, 3)};
    const int m1 = m.body_dofmask[b1], m2 = m.body_dofmask[b2], both = m1 | m2;
    // (The loops over the rows of the block are unrolled over the fixed bound 4 and masked by k < dim: with the run-time bound `dim` the
    //  arrays ax[] and vel[] were indexed dynamically and lived in SCRATCH memory -- vel[k] += ... was a load + store round trip per
    //  row and dof, ~60 per contact.  Same operations in the same order.  dim <= 4: elliptic cones with condim 3 / 4, as ConeEval assumes.)
    V3<T> ax[3];
#pragma unroll
    for (int a = 0; a < 3; a++) ax[a] = V3<T>{CON(e, c, 4 + 3 * a), CON(e, c, 5 + 3 * a), CON(e, c, 6 + 3 * a)};
    T vel[4] = {0, 0, 0, 0};
    const View<T, LM> cdof = cdof_view<T, LM>(e), qvel = qvel_view<T, LM>(e);
    for (int i = 0; i < e.nv; i++) {
        if (!((both >> i) & 1)) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k < dim) ej_set(e, r0 + k, i, T(0));
            continue;
        }
        const V3<T> w = cdof.get3(6 * i), lin = cdof.get3(6 * i + 3) + cross(w, pos);
        const T qd = qvel.get(i);
        const bool in1 = (m1 >> i) & 1, in2 = (m2 >> i) & 1;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k < dim) {
                const T val = k >= 3 ? dot(ax[k >= 3 ? k - 3 : 0], w) : dot(ax[k < 3 ? k : 0], lin);
                T acc = 0;
                if (in2) acc += T(1) * val;            // body 2 first, then body 1 (the order of the row-by-row accumulation)
                if (in1) acc += T(-1) * val;
                ej_set(e, r0 + k, i, acc);
                vel[k] += acc * qd;
            }
        }
    }
    for (int k = 0; k < dim; k++) { IEFC(e, r0 + k, 0) = C_CONTACT; IEFC(e, r0 + k, 1) = c; }
    T solref[2] = {CON(e, c, 17), CON(e, c, 18)}, solimp[5];
    for (int k = 0; k < 5; k++) solimp[k] = CON(e, c, 19 + k);
    const T wt = m.geom_invweight0[2 * g1] + m.geom_invweight0[2 * g2];
    const T f0 = CON(e, c, 14), f1 = CON(e, c, 15);
    T R0, B;
    // solver row descriptor: type + 16 * (rows in this cone block) + 256 * (index inside the block); friction scale of row 0 = mu
    finish_row(e, r0, solref, solimp, wt, dist - inc, vel[0], f0, T(C_CONTACT + 16 * dim), &R0, &B);
    // friction rows: R scaled by friction ratios (impratio 1), aref = -B * vel
#pragma unroll
    for (int k = 1; k < 4; k++) {
        if (k < dim) {
            const T fk = k < 3 ? f0 : f1;
            const T R = R0 * f0 * f0 / (fk * fk);
            sr_set(e, r0 + k, SR_D, 1 / R); sr_set(e, r0 + k, SR_AREF, -B * vel[k]);
            sr_set(e, r0 + k, SR_FRI, fk); sr_set(e, r0 + k, SR_INFO, T(C_CONTACT + 16 * dim + 256 * k)); sr_set(e, r0 + k, SR_JV, T(0));
        }
    }
    CON(e, c, 24) = f0;  // mu = friction[0] * sqrt(R[1]/R[0]) with impratio 1
}

// Constraint rows in two passes: (A) every sub-lane redundantly walks the welds / limited joints / contacts and assigns
// row ranges, the solver's block list and a work-item list (cheap: a few loads per constraint); (B) the work items
// -- building the Jacobian rows, impedance, reference accelerations -- are split over the environment's sub-lanes.
template <typename T>
MW_STAGE_FN void make_constraints(const Env<T> e_) {
    const Env<T> e = e_.uniform();
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    int nefc = 0, nblk = 0, nwork = 0, flags = 0, want = 0;
    const int maxefc = m.sz.maxefc;
    MW_TICK(t_mc0)
    auto work = [&](int kind, int id, int r0) {
        e.I(L.iwork + 3 * nwork) = kind; e.I(L.iwork + 3 * nwork + 1) = id; e.I(L.iwork + 3 * nwork + 2) = r0;
        nwork++;
    };
    // ---- weld(mocap, hand): 3 translational + 3 rotational rows ----
    for (int q = 0; q < m.sz.neq; q++) {
        want += 6;
        if (nefc + 6 > maxefc) { flags |= ST_ROW_OVERFLOW; continue; }
        work(C_EQUALITY, q, nefc);
        for (int k = 0; k < 6; k++) set_block_row(e, nblk + k, nefc + k);
        nblk += 6; nefc += 6;
    }
    // ---- joint limits ----
    for (int j = 0; j < m.sz.njnt; j++) {
        if (!m.jnt_limited[j] || m.jnt_type[j] == J_FREE) continue;
        const T q = e.chain_lds ? qpos_view<T, true>(e).get(m.jnt_qposadr[j]) : e.R(L.qpos + m.jnt_qposadr[j]), margin = m.jnt_margin[j];
        for (int side = -1; side <= 1; side += 2) {
            const T dist = side * (m.jnt_range[2 * j + (side + 1) / 2] - q);
            if (dist < margin) {
                want += 1;
                if (nefc + 1 > maxefc) { flags |= ST_ROW_OVERFLOW; continue; }
                work(C_LIMIT, 2 * j + (side > 0), nefc);
                set_block_row(e, nblk, nefc);
                nblk++; nefc++;
            }
        }
    }
    // ---- contacts (elliptic cones, condim 3 or 4) ----
    // (eight contacts per trip: their three loads each are issued together; the walk itself -- row ranges in contact order -- is serial)
    const int ncon = e.I(L.icount);
    for (int c0 = 0; c0 < ncon; c0 += 8) {
        T cdist[8], cinc[8];
        int cdim[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int c = c0 + q < ncon ? c0 + q : ncon - 1;
            cdist[q] = CON(e, c, 0); cinc[q] = CON(e, c, 13); cdim[q] = ICON(e, c, 2);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int c = c0 + q;
            if (c >= ncon) break;
            const T dist = cdist[q], inc = cinc[q];
            const int dim = cdim[q];
            int adr = -1;
            if (dist < inc) {
                want += dim;
                if (nefc + dim > maxefc) flags |= ST_ROW_OVERFLOW;
                else {
                    adr = nefc;
                    work(C_CONTACT, c, nefc);
                    set_block_row(e, nblk, nefc);
                    nblk++; nefc += dim;
                }
            }
            ICON(e, c, 3) = adr;
        }
    }
    if (sub_disagree(e, nefc) || sub_disagree(e, (nblk << 12) ^ nwork) || sub_disagree(e, want)) flags |= ST_DIVERGED;   // canary (mw_common.hpp)
    e.I(L.icount + 1) = nefc;
    e.I(L.icount + IC_NBLK) = nblk;
    if (want > e.I(L.icount + IC_WANT_EFC)) e.I(L.icount + IC_WANT_EFC) = want;
    if (flags) e.I(L.icount + 3) |= flags;
    MW_SYNC();
    MW_TICK(t_mc1)
    MW_SUBS(e, sub) {
        for (int w = sub; w < nwork; w += e.nsub) {
            const int kind = e.I(L.iwork + 3 * w), id = e.I(L.iwork + 3 * w + 1), r0 = e.I(L.iwork + 3 * w + 2);
            if (e.chain_lds) {          // (cdof / qvel from their scratchpad slots: wave-uniform choice)
                if (kind == C_EQUALITY) weld_rows<T, true>(e, id, r0);
                else if (kind == C_LIMIT) limit_row<T, true>(e, id, r0);
                else contact_rows<T, true>(e, id, r0);
            } else {
                if (kind == C_EQUALITY) weld_rows<T, false>(e, id, r0);
                else if (kind == C_LIMIT) limit_row<T, false>(e, id, r0);
                else contact_rows<T, false>(e, id, r0);
            }
        }
    }
    MW_SYNC();
#if defined(MW_STEP_FINE) && defined(MW_SOLVER_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    MW_TICK(t_mc2)          // step-level timers: slot 5 = the serial walk (row ranges, work list), slot 6 = the work items (rows)
    e.I(L.icount + 4 + 5) += (int)((t_mc1 - t_mc0) >> 4); e.I(L.icount + 4 + 6) += (int)((t_mc2 - t_mc1) >> 4);
#endif
}


// ------------------------------------------------------------------ Newton solver
// Row accessors for the sweep bodies: Rows<T, 1> (= Rows<T, true>) reads the scratchpad unconditionally (the caller has checked
// that row i .. i+3 are inside it), Rows<T, 2> the column store unconditionally (all of them beyond the scratchpad), Rows<T, 0>
// (= Rows<T, false>) takes the per-access generic path.  Instantiating each sweep body per mode keeps the reads of one row in
// a single basic block (issued back to back, one wait): behind a per-access branch every read of a row in the column store is
// its own L2 round trip, 16 of them for one cone block.
template <typename T, int MODE>
struct Rows {
    Env<T> e;
    MW_HD T get(int i, int f) const {
        if (MODE == 1) return e.lds[e.S(i, f) * e.lds_stride];
        if (MODE == 2) return EX(e, i, sr_slot(f));
        return sr_get(e, i, f);
    }
    MW_HD void set(int i, int f, T v) const {
        if (MODE == 1) e.lds[e.S(i, f) * e.lds_stride] = v;
        else if (MODE == 2) EX(e, i, sr_slot(f)) = v;
        else sr_set(e, i, f, v);
    }
    MW_HD T getj(int i, int k) const {          // Jacobian entry k of row i
        if (MODE == 1) return e.lds[e.S(i, SR_N + k) * e.lds_stride];
        if (MODE == 2) return EJ(e, i, k);
        return ej_get(e, i, k);
    }
};

// cone bookkeeping for one contact at jar + alpha * Jv; fixed 4-row form (rows >= dim are masked and re-read row r0):
// every index is compile-time, so U / fri stay in registers instead of a dynamically indexed private array
template <typename T>
struct ConeEval { T mu, fri[4], U[4], D[4], jv[4], x[4], N, Tn; int dim, zone; };   // zone: 0 top, 1 bottom(quadratic), 2 middle
template <typename T, typename R>
MW_HD ConeEval<T> cone_eval(const R& rows, int r0, int dim, T alpha) {
    ConeEval<T> z;
    z.dim = dim;
    T tt = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool on = k < dim;
        const int r = on ? r0 + k : r0;
        z.fri[k] = rows.get(r, SR_FRI);
        z.D[k] = rows.get(r, SR_D);
        z.jv[k] = rows.get(r, SR_JV);
        z.x[k] = rows.get(r, SR_JAR) + alpha * z.jv[k];
        z.U[k] = on ? z.x[k] * z.fri[k] : T(0);
        if (k > 0) tt += z.U[k] * z.U[k];
    }
    z.mu = z.fri[0];
    z.N = z.U[0];
    z.Tn = mw_sqrt(tt);
    if (z.N >= z.mu * z.Tn || (z.Tn <= 0 && z.N >= 0)) z.zone = 0;
    else if (z.mu * z.N + z.Tn <= 0 || (z.Tn <= 0 && z.N < 0)) z.zone = 1;
    else z.zone = 2;
    return z;
}

// one row (or cone block) of update_constraint: force, state, cost
template <typename T, typename R>
MW_HD void uc_row(const R& rows, const Env<T> e, int i, T* cost) {
    const int info = (int)rows.get(i, SR_INFO), type = info & 15;
    if (info >= 256) return;                    // inside a cone block: handled from the block's first row
    const T D = rows.get(i, SR_D), jar = rows.get(i, SR_JAR);
    if (type == C_EQUALITY || (type == C_LIMIT && jar < 0)) {
        const T f = -D * jar;
        *cost += T(0.5) * D * jar * jar;
        rows.set(i, SR_FORCE, f); rows.set(i, SR_STATE, T(S_QUADRATIC));
    } else if (type == C_LIMIT) {
        rows.set(i, SR_FORCE, T(0)); rows.set(i, SR_STATE, T(S_SATISFIED));
    } else {
        const int dim = (info >> 4) & 15;
        ConeEval<T> z = cone_eval<T>(rows, i, dim, T(0));
        int st;
        T f[4] = {0, 0, 0, 0};
        if (z.zone == 0) {
            st = S_SATISFIED;
        } else if (z.zone == 1) {
            st = S_QUADRATIC;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k < dim) { f[k] = -z.D[k] * z.x[k]; *cost += T(0.5) * z.D[k] * z.x[k] * z.x[k]; }
        } else {
            st = S_CONE;
            const T Dm = D / (z.mu * z.mu * (1 + z.mu * z.mu)), NmT = z.N - z.mu * z.Tn;
            *cost += T(0.5) * Dm * NmT * NmT;
            const T f0 = -Dm * NmT * z.mu;
            f[0] = f0;
#pragma unroll
            for (int k = 1; k < 4; k++) f[k] = -f0 / z.Tn * z.U[k] * z.fri[k];
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k < dim) { rows.set(i + k, SR_FORCE, f[k]); rows.set(i + k, SR_STATE, T(st)); }
    }
}

// cost, forces, states at the current jar; qfrc_constraint = J' force; returns total cost incl. Gauss term
template <typename T, int NV>
MW_STAGE_FN T update_constraint(const Env<T> e_) {
    const Env<T> e = e_.uniform();
    CLayout& L = e.lay();
    const int nv = e.nv, nefc = e.I(L.icount + 1), nblk = e.I(L.icount + IC_NBLK);
    const Rows<T, true> fast{e};
    const Rows<T, false> slow{e};
    T cp[MW_NSLOT];
    MW_SUBS(e, sub) {
        T c = 0;
        for (int k = sub; k < nblk; k += e.nsub) {
            const int i = block_row(e, k);
            if (i + 4 <= e.lds_rows) uc_row<T>(fast, e, i, &c);
            else uc_row<T>(slow, e, i, &c);
        }
        cp[MW_SLOT(sub)] = c;
    }
    MW_SYNC();
    const T cost = sub_sum(e, cp);
    T gauss = 0;
    {          // the four vectors in one batch of loads (a runtime loop waits for its four loads in every trip)
        T ma[NV], sm[NV], qa[NV], qs[NV];
        vec_load<T, NV>(e, L.Ma, nv, ma); vec_load<T, NV>(e, L.smooth, nv, sm);
        vec_load<T, NV>(e, L.qacc, nv, qa); vec_load<T, NV>(e, L.qacc_smooth, nv, qs);
#pragma unroll
        for (int k = 0; k < NV; k++)
            if (k < nv) gauss += (ma[k] - sm[k]) * (qa[k] - qs[k]);
    }
    T qf[MW_NSLOT][NV];
    MW_SUBS(e, sub) {
        T* q = qf[MW_SLOT(sub)];
#pragma unroll
        for (int k = 0; k < NV; k++) q[k] = 0;
        for (int i = sub; i < nefc; i += e.nsub) {      // J' f over the active rows, accumulated in registers
            const T f = sr_get(e, i, SR_FORCE);
            if (f == 0) continue;
            T j[NV];
            jrow_load<T, NV>(e, i, nv, j);
#pragma unroll
            for (int k = 0; k < NV; k++) q[k] += j[k] * f;
        }
    }
    sub_sum_n<NV>(e, qf);
    vec_store<T, NV>(e, L.qfrc_c, nv, qf[0]);
    return cost + T(0.5) * gauss;
}

// one row (or cone block) of the line-search cost: cost C, derivatives D1 / D2, cancellation magnitude A1
template <typename T, typename R>
MW_HD void le_row(const R& rows, int i, T alpha, T* C, T* D1, T* D2, T* A1) {
    const int info = (int)rows.get(i, SR_INFO), type = info & 15;
    if (info >= 256) return;
    if (type != C_CONTACT) {
        const T D = rows.get(i, SR_D), jv = rows.get(i, SR_JV), x0 = rows.get(i, SR_JAR), x = x0 + alpha * jv;
        if (type == C_EQUALITY || x < 0) {
            *C += T(0.5) * D * x * x; *D1 += D * x * jv; *D2 += D * jv * jv;
            *A1 += D * (mw_abs(x0) + mw_abs(alpha * jv)) * mw_abs(jv);
        }
        return;
    }
    const int dim = (info >> 4) & 15;
    ConeEval<T> z = cone_eval<T>(rows, i, dim, alpha);
    if (z.zone == 1) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const T Dk = z.D[k], jk = z.jv[k], xk = z.x[k];
            if (k < dim) { *C += T(0.5) * Dk * xk * xk; *D1 += Dk * xk * jk; *D2 += Dk * jk * jk; *A1 += Dk * mw_abs(xk * jk); }
        }
    } else if (z.zone == 2) {
        T UV = 0, VV = 0;
#pragma unroll
        for (int k = 1; k < 4; k++) {
            const T v = k < dim ? z.jv[k] * z.fri[k] : T(0);
            UV += z.U[k] * v; VV += v * v;
        }
        const T Dm = z.D[0] / (z.mu * z.mu * (1 + z.mu * z.mu));
        const T N1 = z.jv[0] * z.mu, T1 = UV / z.Tn, T2 = VV / z.Tn - UV * T1 / (z.Tn * z.Tn);
        const T NmT = z.N - z.mu * z.Tn, g = N1 - z.mu * T1;
        *C += T(0.5) * Dm * NmT * NmT; *D1 += Dm * NmT * g; *D2 += Dm * (g * g - NmT * z.mu * T2);
        *A1 += Dm * (mw_abs(z.N) + z.mu * z.Tn) * (mw_abs(N1) + z.mu * mw_abs(T1));
    }
}

// constraint part of the cost (no forces written) at jar + alpha*Jv, with 1st/2nd derivatives along the line
// *mag: sum of the magnitudes that cancel inside d1 (rounding-noise scale of the derivative, fp32 termination)
template <typename T>
MW_HD void line_eval(const Env<T> e, int nblk, T alpha, const T* quadGauss, T* cost, T* d1, T* d2, T* mag = nullptr) {
    MW_COUNT(0)
    const Rows<T, true> fast{e};
    const Rows<T, false> slow{e};
    T part[MW_NSLOT][4];
    MW_SUBS(e, sub) {
        T C = 0, D1 = 0, D2 = 0, A1 = 0;
        for (int k = sub; k < nblk; k += e.nsub) {
            const int i = block_row(e, k);
            if (i + 4 <= e.lds_rows) le_row<T>(fast, i, alpha, &C, &D1, &D2, &A1);
            else le_row<T>(slow, i, alpha, &C, &D1, &D2, &A1);
        }
        T* p = part[MW_SLOT(sub)];
        p[0] = C; p[1] = D1; p[2] = D2; p[3] = A1;
    }
    sub_sum_n<4>(e, part);
    *cost = part[0][0] + (alpha * alpha * quadGauss[2] + alpha * quadGauss[1] + quadGauss[0]);
    *d1 = part[0][1] + (2 * alpha * quadGauss[2] + quadGauss[1]);
    *d2 = part[0][2] + 2 * quadGauss[2];
    if (mag) *mag = part[0][3] + mw_abs(2 * alpha * quadGauss[2]) + mw_abs(quadGauss[1]);
}

// Precision of the Newton Hessian and its Cholesky factor: SINGLE precision also in an fp64 context (unless -DMW_HESSIAN_F64).
// H = M + J' D J is then a PRECONDITIONER of a double-precision solve: the Jacobian, the constraint forces, the gradient, the
// exact line search, the cost and every convergence test stay in T, so the iteration converges to the same minimiser under
// the same tolerance (the problem is strictly convex; an inexact Newton direction only changes the path).  What it buys: the
// lower triangle is 120 instead of 240 registers for nv = 15, which is what made the fp64 assembly + factorisation spill
// (-9 % time per MT50 step at 4096 envs; Newton iteration counts unchanged, ~1 extra line-search evaluation per iteration).
// What it costs: two implementations no longer follow bit-identical iterates, so device-vs-oracle agreement over hundreds of
// substeps is ~1e-9 instead of ~1e-14 (the reference tolerance is 1e-5).
template <typename T> struct HessType { typedef T type; };
#if !defined(MW_HESSIAN_F64)
template <> struct HessType<double> { typedef float type; };
#endif
template <typename T, typename HT, int NV>
MW_HD void tri_load_as(const Env<T> e, int A, int n, HT* h) {
#pragma unroll
    for (int i = 0; i < NV; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            const T v = e.R(A + (i < n ? i * n + j : 0));
            h[tri(i, j)] = i < n ? (HT)v : (i == j ? HT(1) : HT(0));
        }
}

// search direction s = -H^-1 g of one Newton iteration: reads -g from L.grad, writes s to L.search.
// H = M + J' D J over quadratic rows (+ dense cone blocks): lower triangle in registers (a partial sum per sub-lane over its
// blocks, M on sub-lane 0), butterfly total, then Cholesky + solve replicated on the sub-lanes.
template <typename T, typename HT, int NV>
MW_STAGE_FN void newton_direction(const Env<T> e_) {
    const Env<T> e = e_.uniform();
    CLayout& L = e.lay();
    const int nv = e.nv, nblk = e.I(L.icount + IC_NBLK);
    constexpr int NT = NV * (NV + 1) / 2;
    HT H[MW_NSLOT][NT];
    MW_SUBS(e, sub) {
        HT* h = H[MW_SLOT(sub)];
        if (sub == 0) tri_load_as<T, HT, NV>(e, L.qM, nv, h);
        else {
#pragma unroll
            for (int k = 0; k < NT; k++) h[k] = 0;
        }
        for (int kb = sub; kb < nblk; kb += e.nsub) {
            const int i = block_row(e, kb);
            const int st = (int)sr_get(e, i, SR_STATE);
            if (st == S_SATISFIED) continue;
            const int info = (int)sr_get(e, i, SR_INFO), dim = (info >> 4) & 15, type = info & 15;
            if (st == S_CONE) {
                ConeEval<T> z = cone_eval<T>(Rows<T, false>{e}, i, dim, T(0));
                const T Dm = z.D[0] / (z.mu * z.mu * (1 + z.mu * z.mu));
                HT Hc[16];
                const T scl = z.mu * z.N / (z.Tn * z.Tn * z.Tn), dg = z.mu * z.mu - z.mu * z.N / z.Tn;
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        T v;
                        if (r == 0 && c == 0) v = 1;
                        else if (r == 0) v = -z.mu * z.U[c] / z.Tn;
                        else if (c == 0) v = -z.mu * z.U[r] / z.Tn;
                        else v = scl * z.U[r] * z.U[c] + (r == c ? dg : T(0));
                        Hc[4 * r + c] = (r < dim && c < dim) ? HT(v * Dm * z.fri[r] * z.fri[c]) : HT(0);
                    }
                HT j[4][NV];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const bool on = r < dim;
                    jrow_load_as<T, HT, NV>(e, on ? i + r : i, nv, j[r]);
#pragma unroll
                    for (int a = 0; a < NV; a++) j[r][a] = (on && a < nv) ? j[r][a] : HT(0);
                }
#pragma unroll
                for (int a = 0; a < NV; a++) {
                    HT t[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        t[c] = 0;
#pragma unroll
                        for (int r = 0; r < 4; r++) t[c] += j[r][a] * Hc[4 * r + c];
                    }
#pragma unroll
                    for (int b = 0; b <= a; b++) {
                        HT acc = 0;
#pragma unroll
                        for (int c = 0; c < 4; c++) acc += t[c] * j[c][b];
                        h[tri(a, b)] += acc;
                    }
                }
            } else {          // quadratic: every row of the block is an independent rank-1 term
                const int nr = type == C_CONTACT ? dim : 1;
                for (int r = 0; r < nr; r++) {
                    const HT D = (HT)sr_get(e, i + r, SR_D);
                    HT j[NV];
                    jrow_load_as<T, HT, NV>(e, i + r, nv, j);
#pragma unroll
                    for (int a = 0; a < NV; a++) j[a] = a < nv ? j[a] : HT(0);
#pragma unroll
                    for (int a = 0; a < NV; a++) {
                        const HT Da = D * j[a];
#pragma unroll
                        for (int b = 0; b <= a; b++) h[tri(a, b)] += Da * j[b];
                    }
                }
            }
        }
    }
    MW_TICK(t_rows)
    sub_sum_n<NT>(e, H);
    MW_TICK(t_bfly)
    HT inv[NV], sh[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) sh[k] = k < nv ? (HT)e.R(L.grad + k) : HT(0);
    chol_reg<HT, NV>(H[0], inv);
    chol_solve_reg<HT, NV>(H[0], inv, sh);
#pragma unroll
    for (int k = 0; k < NV; k++)
        if (k < nv) e.R(L.search + k) = (T)sh[k];
    MW_TICK(t_end)
    MW_TOCK(e, L, 2, t_rows, t_bfly)      // timing builds: slot 2 ("chol") = butterfly of the partial Hessians + factorisation + solve
    MW_TOCK(e, L, 2, t_bfly, t_end)
}

// ------------------------------------------------------------------ lane-role helpers (device only)
// The wave-cooperative Newton direction of rounds 4-5 (newton_direction_wave: H = M + J' D J in the accumulators of
// v_mfma_f32_16x16x1_4b_f32, four environments per instruction; right-looking Cholesky in that layout; DPP triangular solves; the
// 17th dof of the stick scenes as a border) lives on inside solve_wave (mw_solve_wave.hpp), which runs the WHOLE solve in its
// layout -- lane 16 b + i = environment b of a group of four, dof i.  What is left here are the pieces both use.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float mw_f16v __attribute__((ext_vector_type(16)));
template <int CTRL> __device__ inline float mw_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a block, in every lane: four rotations inside the DPP row (row_ror:8 / 4 / 2 / 1), no LDS crossbar
__device__ inline float blk_sum(float v) { v += mw_dpp<0x128>(v); v += mw_dpp<0x124>(v); v += mw_dpp<0x122>(v); v += mw_dpp<0x121>(v); return v; }
// lane k of every 16-lane block to all lanes of that block: one DPP move (row_newbcast:k; k is a constant after unrolling)
__device__ inline float blk_bcast(float v, int k) {
    switch (k & 15) {
    case 0: return mw_dpp<0x150>(v); case 1: return mw_dpp<0x151>(v); case 2: return mw_dpp<0x152>(v); case 3: return mw_dpp<0x153>(v);
    case 4: return mw_dpp<0x154>(v); case 5: return mw_dpp<0x155>(v); case 6: return mw_dpp<0x156>(v); case 7: return mw_dpp<0x157>(v);
    case 8: return mw_dpp<0x158>(v); case 9: return mw_dpp<0x159>(v); case 10: return mw_dpp<0x15A>(v); case 11: return mw_dpp<0x15B>(v);
    case 12: return mw_dpp<0x15C>(v); case 13: return mw_dpp<0x15D>(v); case 14: return mw_dpp<0x15E>(v); default: return mw_dpp<0x15F>(v);
    }
}
template <typename T>
__device__ inline Env<T> env_view(const Env<T>& e, int slot) {          // the same workgroup's environment `slot`, seen from this thread
    Env<T> r = e;
    const int d = slot - e.slot;
    r.col = e.col + d; r.icol = e.icol + d; r.lds = e.lds + d; r.slot = slot;
    return r;
}
#endif
}  // namespace mw
#include "mw_solve_wave.hpp"          // the whole solve in the lane-role layout (device, >= 4 sub-lanes per environment)
namespace mw {

template <typename T, int NV>
MW_HD void solve_impl(const Env<T> e) {
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const int nv = e.nv, nefc = e.I(L.icount + 1), nblk = e.I(L.icount + IC_NBLK);
    constexpr int NT = NV * (NV + 1) / 2;
    MW_TICK(t_a)
    auto set_point = [&](int src) {   // qacc <- src ; Ma, jar
        T x[NV];
        vec_load<T, NV>(e, src, nv, x);
        vec_store<T, NV>(e, L.qacc, nv, x);
        mat_vec_rows<T, NV>(e, L.qM, nv, x, L.Ma);
        MW_SUBS(e, sub) {
            for (int i = sub; i < nefc; i += e.nsub) {
                T j[NV], s = -sr_get(e, i, SR_AREF);
                jrow_load<T, NV>(e, i, nv, j);
#pragma unroll
                for (int k = 0; k < NV; k++) s += j[k] * x[k];
                sr_set(e, i, SR_JAR, s);
            }
        }
        MW_SYNC();
    };
    // warm start: the better of qacc_warmstart and qacc_smooth (ties go to the warm start)
    set_point(L.qacc_smooth);
    const T cs = update_constraint<T, NV>(e);
    set_point(L.warm);
    T cost = update_constraint<T, NV>(e);
    // (nefc == 0 -- an environment without rows inside a wave that has some, see solve(): the unconstrained optimum IS qacc_smooth,
    //  whatever the rounding of the two costs says)
    if (cost > cs || nefc == 0) { set_point(L.qacc_smooth); cost = update_constraint<T, NV>(e); }
    const T scale = 1 / (m.meaninertia * T(nv > 1 ? nv : 1));
    MW_TICK(t_b)
    MW_TOCK(e, L, 0, t_a, t_b)
    MW_COUNT(2)
    // (Rounds 4-5 took the Newton direction from a wave-cooperative routine here; since round 6 every layout with sub-lanes runs
    //  solve_wave instead of this function, which remains the solver of the host build and of layouts with fewer than four sub-lanes.)
    bool active = true;
    for (int iter = 0; iter < m.sz.iterations; iter++) {
        T sr[NV];             // gradient, then the search direction
        if (active) {
            MW_COUNT(1)
            T gn = 0;
#pragma unroll
            for (int k = 0; k < NV; k++) {
                const int kk = k < nv ? k : 0;
                const T g = e.R(L.Ma + kk) - e.R(L.smooth + kk) - e.R(L.qfrc_c + kk);
                sr[k] = k < nv ? -g : T(0);
                if (k < nv) gn += g * g;
            }
            if (scale * mw_sqrt(gn) < m.tolerance) active = false;
            // Newton direction: -g goes through the column store (L.grad) and the direction comes back in L.search
            else vec_store<T, NV>(e, L.grad, nv, sr);
        }
        MW_TICK(t_c)
        if (!active) break;
        // (its own non-inlined function, so that the register allocation of the Hessian -- nv (nv + 1) / 2 accumulators + up to
        //  four Jacobian rows -- is not mixed with everything that is live in this loop)
        newton_direction<T, typename HessType<T>::type, NV>(e);
        if (!active) continue;
        active = [&]() -> bool {
        vec_load<T, NV>(e, L.search, nv, sr);
        MW_TICK(t_e)
        MW_TOCK(e, L, 1, t_c, t_e)
        MW_TICK(t_f)
        // ---- exact line search (safeguarded Newton on the 1-D convex cost) ----
        T snorm = 0, quadGauss[3] = {0, 0, 0};
        {
            T Mv[NV];
            vec_store<T, NV>(e, L.search, nv, sr);
            mat_vec_rows<T, NV>(e, L.qM, nv, sr, L.Mv);
            vec_load<T, NV>(e, L.Mv, nv, Mv);
#pragma unroll
            for (int k = 0; k < NV; k++) {
                const int kk = k < nv ? k : 0;
                const T sk = sr[k], r = e.R(L.Ma + kk) - e.R(L.smooth + kk), dq = e.R(L.qacc + kk) - e.R(L.qacc_smooth + kk);
                if (k < nv) {
                    snorm += sk * sk;
                    quadGauss[1] += sk * r;
                    quadGauss[2] += T(0.5) * sk * Mv[k];
                    quadGauss[0] += T(0.5) * r * dq;
                }
            }
        }
        snorm = mw_sqrt(snorm);
        if (snorm < T(1e-15)) return false;
        MW_SUBS(e, sub) {
            for (int i = sub; i < nefc; i += e.nsub) {
                T j[NV], s = 0;
                jrow_load<T, NV>(e, i, nv, j);
#pragma unroll
                for (int k = 0; k < NV; k++) s += j[k] * sr[k];
                sr_set(e, i, SR_JV, s);
            }
        }
        MW_SYNC();
        const T gtol = m.tolerance * T(0.01) * snorm / scale;
        MW_TICK(t_g)
        MW_TOCK(e, L, 3, t_f, t_g)
        T c0, d1, d2;
        line_eval(e, nblk, T(0), quadGauss, &c0, &d1, &d2);
        if (d1 >= 0 || d2 <= 0) { e.I(L.icount + IC_SOLVER_STALL) += 1; return false; }   // not a descent direction (single-precision factor of an ill-conditioned H, or rounding at the optimum): the search is abandoned like in the reference solver, and counted (mw_status)
        T lo = 0, hi = -1, alpha = -d1 / d2;
        int nls = 0;
        for (int it = 0; it < m.sz.ls_iterations; it++) {
            T ca, da, dda, mag;
            nls++;
            line_eval(e, nblk, alpha, quadGauss, &ca, &da, &dda, &mag);
            if (mw_abs(da) < gtol) break;
            // single precision: the derivative cannot be resolved below its own rounding noise (a few ulp of the
            // magnitudes that cancel inside it); without this test ~10 % of the searches ran to ls_iterations and,
            // 64 lanes to a wave, so did every wave
            if (sizeof(T) == 4 && mw_abs(da) <= T(1e-6) * mag) break;
            if (da < 0) lo = alpha; else hi = alpha;
            T an = alpha - da / dda;
            if (hi < 0) { if (an <= lo) an = 2 * alpha; }
            else if (!(an > lo && an < hi)) an = T(0.5) * (lo + hi);
            if (hi > 0 && (hi - lo) <= (sizeof(T) == 8 ? T(1e-16) : T(5e-7)) * hi) { alpha = T(0.5) * (lo + hi); break; }
            if (an == alpha) break;
            alpha = an;
        }
        MW_HIST(1, nls)
        (void)nls;
        MW_TICK(t_h)
        MW_TOCK(e, L, 4, t_g, t_h)
        MW_TADD(e, L, 6, nls)
        if (alpha == 0) return false;
        {          // qacc += alpha s, Ma += alpha Mv: all loads first, then the stores (see integrate_impl)
            T qa[NV], ma[NV], mv[NV];
            vec_load<T, NV>(e, L.qacc, nv, qa); vec_load<T, NV>(e, L.Ma, nv, ma); vec_load<T, NV>(e, L.Mv, nv, mv);
#pragma unroll
            for (int k = 0; k < NV; k++) { qa[k] += alpha * sr[k]; ma[k] += alpha * mv[k]; }
            vec_store<T, NV>(e, L.qacc, nv, qa); vec_store<T, NV>(e, L.Ma, nv, ma);
        }
        MW_SUBS(e, sub) {
            for (int i = sub; i < nefc; i += e.nsub) sr_set(e, i, SR_JAR, sr_get(e, i, SR_JAR) + alpha * sr_get(e, i, SR_JV));
        }
        MW_SYNC();
        const T old = cost;
        cost = update_constraint<T, NV>(e);
        MW_TICK(t_i)
        MW_TOCK(e, L, 5, t_h, t_i)
        MW_TADD(e, L, 7, 1)
        e.I(L.icount + 2) = iter + 1;
        return !(scale * (old - cost) < m.tolerance);
        }();
    }
    MW_HIST(0, e.I(L.icount + 2))
    // efc_force of the rows kept in the scratchpad -> efcX (read by touching_object and through the ABI); fallback rows already are there
    MW_SUBS(e, sub) {
        const int nl = nefc < e.lds_rows ? nefc : e.lds_rows;
        for (int i = sub; i < nl; i += e.nsub) EX(e, i, 5) = e.lds[e.S(i, SR_FORCE) * e.lds_stride];
    }
    MW_SYNC();
}

// the per-environment solver (host build; layouts with fewer than four sub-lanes per environment): its own non-inlined function, so
// that the five per-nv instantiations of solve_impl, their registers and their 2.6 KB frame do not sit in the dispatcher below
template <typename T>
MW_STAGE_FN void solve_env(const Env<T> e_) {
    const Env<T> e = e_.uniform();
    const int nv = e.nv;
    MW_NV_DISPATCH(nv, (solve_impl<T, NVC>(e)))
}

// (inlined into forward_dynamics since round 6: one call frame -- prologue, callee-saved registers -- less per solve)
template <typename T>
MW_HD void solve(const Env<T> e_) {
    const Env<T> e = e_.uniform();
    CLayout& L = e.lay();
    const int nv = e.nv;
    e.I(L.icount + 2) = 0;
    e.I(L.icount + IC_EULER_READY) = 0;          // (set again by solve_wave's finish for THIS solver output; a stale flag must not reach integrate)
    // WAVE-UNIFORM early exit: only when NO environment of the wave has a constraint row.  In a mixed wave the ones
    // without rows go through solve_impl as well (zero rows: qacc = qacc_smooth bit for bit, qfrc_constraint = 0), so that the
    // wave-cooperative Newton direction is entered under a full EXEC mask.  Every Sawyer scene has the mocap weld (nefc >= 6).
    if (!mw_any(e.I(L.icount + 1) != 0)) {
        T x[MAX_NV];
        vec_load<T, MAX_NV>(e, L.qacc_smooth, nv, x);          // (all loads, then all stores)
        vec_store<T, MAX_NV>(e, L.qacc, nv, x);
        for (int k = 0; k < nv; k++) e.R(L.qfrc_c + k) = 0;
        return;
    }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MW_NO_WAVE_SOLVER)
    // every layout with sub-lanes (all bench configurations): the solve in the lane-role layout, entered by all 64 lanes (ghosts
    // included) under a full EXEC mask; the rows and the smooth forces were written by the environment's own sub-lanes
    if (e.nsub >= 4 && nv <= MAX_NV) {
        MW_SYNC();
        if (nv > 16) solve_wave<T, true>(e);
        else solve_wave<T, false>(e);
        return;
    }
#endif
    solve_env(e);
}

// debugging / parity hooks only (lane_debug): copy the rows kept in the scratchpad into the column store (efcJ, efcX), where
// mw_read finds them; the product path never does this
template <typename T>
MW_HD void mirror_rows(const Env<T> e) {
    const int nefc = e.I(e.lay().icount + 1), nl = nefc < e.lds_rows ? nefc : e.lds_rows;
    for (int i = 0; i < nl; i++) {
        for (int k = 0; k < e.nv; k++) EJ(e, i, k) = e.lds[e.S(i, SR_N + k) * e.lds_stride];
        for (int f = 0; f < SR_N; f++)
            if (f != SR_BLK) EX(e, i, sr_slot(f)) = e.lds[e.S(i, f) * e.lds_stride];
    }
}

// ------------------------------------------------------------------ pipeline
#if defined(MW_PROFILE) && !defined(__HIPCC__)
#include <chrono>
inline double* mw_prof() { static double t[8] = {0}; return t; }
#define MW_STAGE(i, call) { auto t0_ = std::chrono::steady_clock::now(); call; mw_prof()[i] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(); }
#else
#define MW_STAGE(i, call) call;
#endif
// mj_forward = kinematics + everything that depends on it.  The second half is its own function because SawyerXYZEnv.step's
// final mj_forward (sawyer_xyz_env.py:620) is only OBSERVED through body / geom / site frames -- and, for the 14 tasks whose
// reward calls touching_object (:401-440), through data.contact / data.efc_force; see env_step.
template <typename T>
MW_HD void forward_dynamics(const Env<T> e_) {
    const Env<T> e = e_.uniform();
    CLayout& L = e.lay();
#if defined(MW_SOLVER_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    // (the bias forces do not depend on the contacts: smooth_forces runs BEFORE make_constraints fills the constraint rows,
    //  because its chain transients use the rows' scratchpad slots, Env::chain_lds; the timer slots keep their meaning)
    MW_TICK(t1) crb(e);
    MW_TICK(t2) smooth_forces(e);
    MW_TICK(t3) collision(e);
    MW_TICK(t4) make_constraints(e);
    MW_TICK(t5) solve(e);
    MW_TICK(t6)
    MW_TOCK(e, L, 9, t1, t2) MW_TOCK(e, L, 12, t2, t3) MW_TOCK(e, L, 10, t3, t4) MW_TOCK(e, L, 11, t4, t5) MW_TOCK(e, L, 13, t5, t6)
#else
    MW_STAGE(1, crb(e))
    MW_STAGE(4, smooth_forces(e))
    MW_STAGE(2, collision(e))
    MW_STAGE(3, make_constraints(e))
    MW_STAGE(5, solve(e))
#endif
    e.I(L.icount + IC_DYN_VALID) = 1;
}
template <typename T>
MW_HD void forward(const Env<T> e_) {
    const Env<T> e = e_.uniform();
#if defined(MW_SOLVER_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    CLayout& L = e.lay();
    MW_TICK(t0) kinematics(e);
    MW_TICK(t1)
    MW_TOCK(e, L, 8, t0, t1)
#else
    MW_STAGE(0, kinematics(e))
#endif
    forward_dynamics(e);
}

// The Euler step of mj_step after the solver: warm start <- qacc; (M + h B) a = qfrc_smooth + qfrc_constraint (implicit joint
// damping); qvel += h a; qpos += h qvel (quaternions of free joints integrated on the sphere).
// Everything is read into registers FIRST and stored afterwards.  The compiler may not move a load of the column store above
// an earlier store to it (it cannot prove that the two do not overlap), so an element-wise `dst[k] = f(src[k])` loop is one
// memory round trip PER ELEMENT (load, s_waitcnt vmcnt(0), store): the former element-wise form of this tail -- copying the
// lower triangle of M into a column qH, adding h B, then qvel / qpos dof by dof -- was ~230 dependent round trips per substep.
// Same operations on the same values in the same order; M + h B now only ever exists in registers.
template <typename T, int NV>
MW_HD void integrate_impl(const Env<T> e) {
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const int nv = e.nv;
    const T h = m.timestep;
    constexpr int NT = NV * (NV + 1) / 2;
    T acc[NV], rhs[NV], fc[NV], qv[NV], qp[NV], H[NT], inv[NV];
    int qa[NV];
    vec_load<T, NV>(e, L.qacc, nv, acc);
    vec_load<T, NV>(e, L.smooth, nv, rhs);
    vec_load<T, NV>(e, L.qfrc_c, nv, fc);
    vec_load<T, NV>(e, L.qvel, nv, qv);
    tri_load<T, NV>(e, L.qM, nv, H);
#pragma unroll
    for (int k = 0; k < NV; k++) {
        qa[k] = k < nv ? m.dof_qposadr[k] : -1;
        qp[k] = e.R(L.qpos + (qa[k] >= 0 ? qa[k] : 0));
    }
    // ---- all loads are issued; from here on only arithmetic and stores ----
    vec_store<T, NV>(e, L.warm, nv, acc);
#pragma unroll
    for (int k = 0; k < NV; k++) {
        if (k < nv) H[tri(k, k)] += h * m.dof_damping[k];
        rhs[k] = rhs[k] + fc[k];
    }
    chol_reg<T, NV>(H, inv);
    chol_solve_reg<T, NV>(H, inv, rhs);
    vec_store<T, NV>(e, L.search, nv, rhs);
#pragma unroll
    for (int k = 0; k < NV; k++) qv[k] += h * rhs[k];
    vec_store<T, NV>(e, L.qvel, nv, qv);
#pragma unroll
    for (int k = 0; k < NV; k++)
        if (qa[k] >= 0) e.R(L.qpos + qa[k]) = qp[k] + h * qv[k];
}

// the same step when the solver has already solved (M + h B) a = qfrc_smooth + qfrc_constraint (solve_wave's finish, IC_EULER_READY):
// warm start <- qacc, qvel += h a, qpos += h qvel; all loads first, then the stores
template <typename T>
MW_HD void integrate_ready(const Env<T> e) {
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const int nv = e.nv;
    const T h = m.timestep;
    constexpr int NV = MAX_NV;
    T acc[NV], a[NV], qv[NV], qp[NV];
    int qa[NV];
    vec_load<T, NV>(e, L.qacc, nv, acc);
    vec_load<T, NV>(e, L.search, nv, a);
    vec_load<T, NV>(e, L.qvel, nv, qv);
#pragma unroll
    for (int k = 0; k < NV; k++) {
        qa[k] = k < nv ? m.dof_qposadr[k] : -1;
        qp[k] = e.R(L.qpos + (qa[k] >= 0 ? qa[k] : 0));
    }
    vec_store<T, NV>(e, L.warm, nv, acc);
#pragma unroll
    for (int k = 0; k < NV; k++) qv[k] += h * a[k];
    vec_store<T, NV>(e, L.qvel, nv, qv);
#pragma unroll
    for (int k = 0; k < NV; k++)
        if (qa[k] >= 0) e.R(L.qpos + qa[k]) = qp[k] + h * qv[k];
}

// the Euler step with its own factorisation (host build; layouts without the lane-role solver): non-inlined, the five per-nv
// instantiations and their 256 + 256 registers stay out of integrate below
template <typename T>
MW_STAGE_FN void integrate_full(const Env<T> e_, bool mine) {
    const Env<T> e = e_.uniform();
    const int nv = e.nv;
    if (mine) { MW_NV_DISPATCH(nv, (integrate_impl<T, NVC>(e))) }
}

// the part of mj_step after mj_forward: Euler step (integrate_impl) + the orientation of the free bodies + time
// (inlined into substep since round 6: with the acceleration ready it is a 17-entry vector update, not worth a call frame)
template <typename T>
MW_HD void integrate(const Env<T> e_) {
    const Env<T> e = e_.uniform();
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const T h = m.timestep;
    // (wave-uniform in practice: solve() takes one path for the whole wave; mw_any keeps the call convergent all the same)
    const bool ready = e.I(L.icount + IC_EULER_READY) == 1;
    if (ready) { e.I(L.icount + IC_EULER_READY) = 0; integrate_ready(e); }
    if (mw_any(!ready)) integrate_full(e, !ready);
    for (int j = 0; j < m.sz.njnt; j++) {          // orientation of the free bodies: q <- q * exp(h w / 2)
        if (m.jnt_type[j] != J_FREE) continue;
        const int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
        V3<T> w = ld3(e, L.qvel + da + 3);
        T wn;
        V3<T> ax = normalized(w, &wn);
        const T ang = wn * h;
        if (ang >= T(1e-15)) {
            const T s = sin(T(0.5) * ang);
            Q4<T> q = qmul(ld4(e, L.qpos + qa + 3), Q4<T>{cos(T(0.5) * ang), s * ax.x, s * ax.y, s * ax.z});
            st4(e, L.qpos + qa + 3, qnormalized(q));
        }
    }
    e.R(L.time) += h;
}

template <typename T>
MW_HD void substep(const Env<T> e_) {
    const Env<T> e = e_.uniform();
    forward(e);
#if defined(MW_STEP_FINE) && defined(MW_SOLVER_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    MW_TICK(t_i0)
    integrate(e);
    MW_TICK(t_i1)
    e.I(e.lay().icount + 4 + 1) += (int)((t_i1 - t_i0) >> 4);          // step-level timers (-DMW_STEP_FINE): slot 1 = integrate
#else
    integrate(e);
#endif
}

// mj_resetData
template <typename T>
MW_HD void reset_data(const Env<T> e) {
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    for (int i = 0; i < m.sz.nq; i++) e.R(L.qpos + i) = m.qpos0[i];
    for (int i = 0; i < m.sz.nv; i++) { e.R(L.qvel + i) = 0; e.R(L.warm + i) = 0; }
    for (int i = 0; i < m.sz.nu; i++) e.R(L.ctrl + i) = 0;
    e.R(L.time) = 0;
    for (int b = 0; b < m.sz.nbody; b++)
        if (m.body_mocap[b]) st3(e, L.mocap, mv3(m.body_pos + 3 * b));
}

}  // namespace mw
