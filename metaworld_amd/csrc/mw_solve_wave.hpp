// mw_solve_wave.hpp -- the constraint solver in the LANE-ROLE layout (device only; included by mw_phys.hpp).
//
// solve_impl (mw_phys.hpp) runs the Newton solver of one environment on that environment's own sub-lanes: every sub-lane holds
// ALL nv entries of qacc / Ma / grad / search / Mv / qfrc_constraint, the vectors travel through the column store between the
// phases (and between the non-inlined update_constraint / newton_direction_wave calls), the butterfly totals go through the LDS
// crossbar (ds_bpermute).  Here the WHOLE solve -- warm start, gradient, Newton direction, exact line search, update_constraint --
// runs in the layout round 4 introduced for the direction alone:
//
//   lane 16 b + i of the wave works for environment (g0 + b) of the workgroup and DOF i          (b = 0..3, i = 0..15)
//
//  * the nv-vectors live ONE DOF PER LANE in registers for the whole solve (qacc, Ma, qfrc_smooth, qacc_smooth, qfrc_constraint,
//    gradient, search, Mv): nothing of the Newton loop touches the column store; the 17th dof of the stick scenes is a border
//    scalar replicated over the block's lanes (BORDER);
//  * row i of the mass matrix sits in lane i's registers (loaded once per solve), M x is 16 FMAs on the gathered x
//    (16 DPP row_newbcast moves); M in the accumulator layout of the matrix instruction is loaded once per solve, not per iteration;
//  * J x (jar at a new point, Jv along a search direction) is ROW PER LANE: lane i takes rows i, i + 16, ..., one scratchpad read
//    per entry, sum in index order (the same sum as solve_impl's);
//  * J' f is DOF PER LANE: one pass over the rows, two scratchpad reads + one FMA per row, no cross-lane sum at all
//    (solve_impl: nv partial sums per sub-lane + an nv-wide butterfly);
//  * update_constraint and the line-search evaluations are BLOCK PER LANE (lane i takes constraint blocks i, i + 16, ... -- the
//    same uc_row / le_row bodies as solve_impl), their totals a 16-lane DPP butterfly (quad_perm xor 1, xor 2, row_half_mirror,
//    row_mirror: the order of the sub-lane butterfly with 16 sub-lanes, so the totals equal the host harness' for MW_NSUB = 16);
//  * the Newton direction is the one rounds 4-5 computed in newton_direction_wave: H = M + J' D J (+ cone blocks) in single precision in the accumulators of
//    v_mfma_f32_16x16x1_4b_f32, right-looking Cholesky in that layout, DPP triangular solves -- with -g and the direction handed
//    over in registers;
//  * per-environment scalars (cost, alpha, the bracket of the line search, ...) are replicated over the block's 16 lanes; control
//    flow is wave-uniform with block-uniform predicates (an environment that has converged waits for the wave, as before).
//
// The warm start (three candidate points) and the Newton iterations are passes of ONE loop, so that the two expensive bodies --
// "apply" (y = M x, rows[field] = J x) and update_constraint -- exist once in the code.
//
// A workgroup with more than four environments (lpb = 8, 16) is solved in groups of four, one after the other.  Layouts with fewer
// than four sub-lanes per environment (lpb >= 32) and the host build keep solve_impl.
#pragma once

namespace mw {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MW_NO_WAVE_SOLVER)
// timing builds (-DMW_SOLVER_TIMING): the eight solver slots hold the phases of solve_impl (warm, Hasm, chol, MvJv, lsrch, update, counts);
// with -DMW_SOLVE_FINE they hold the pieces of solve_wave instead: 0 the WHOLE lane_step (mw_runtime.hpp), 1 coefficient pre-pass, 2 H rows, 3 Cholesky + solves,
// 4 apply (M x, J x), 5 line search, 6 update_constraint's block sweep, 7 J' f   (cycles / 16, every pass incl. the warm start)
#if defined(MW_STEP_FINE) && defined(MW_SOLVER_TIMING)          // (the slots belong to the step-level timers: mw_runtime.hpp lane_step)
#define SW_FINE(slot, t0, t1)
#define SW_COARSE(x)
#elif defined(MW_SOLVE_FINE) && defined(MW_SOLVER_TIMING)
#define SW_FINE(slot, t0, t1) if (on) { rv.I(L.icount + 4 + (slot)) += (int)(((t1) - (t0)) >> 4); }
#define SW_COARSE(x)
#else
#define SW_FINE(slot, t0, t1)
#define SW_COARSE(x) x
#endif

template <int CTRL> __device__ inline double mw_dpp(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
// total over the 16 lanes of a block, in every lane, summed in the order of the sub-lane butterfly (xor 1, 2, 4, 8)
template <typename T> __device__ inline T blk_sum_t(T v) {
    v += mw_dpp<0xB1>(v);          // quad_perm [1, 0, 3, 2]
    v += mw_dpp<0x4E>(v);          // quad_perm [2, 3, 0, 1]
    v += mw_dpp<0x141>(v);         // row_half_mirror: the other quad of the half row (values are quad-uniform by now)
    v += mw_dpp<0x140>(v);         // row_mirror: the other half row
    return v;
}
template <typename T> __device__ inline T blk_bcast_t(T v, int k) {          // lane k of every block to all its lanes (k constant after unrolling)
    switch (k & 15) {
    case 0: return mw_dpp<0x150>(v); case 1: return mw_dpp<0x151>(v); case 2: return mw_dpp<0x152>(v); case 3: return mw_dpp<0x153>(v);
    case 4: return mw_dpp<0x154>(v); case 5: return mw_dpp<0x155>(v); case 6: return mw_dpp<0x156>(v); case 7: return mw_dpp<0x157>(v);
    case 8: return mw_dpp<0x158>(v); case 9: return mw_dpp<0x159>(v); case 10: return mw_dpp<0x15A>(v); case 11: return mw_dpp<0x15B>(v);
    case 12: return mw_dpp<0x15C>(v); case 13: return mw_dpp<0x15D>(v); case 14: return mw_dpp<0x15E>(v); default: return mw_dpp<0x15F>(v);
    }
}
// total over the R blocks that work for the same environment (replicas, see solve_wave): R = 2 -> blocks b and b ^ 2, R = 4 -> all
// four; every replica ends up with the same bits (a + b = b + a)
template <typename T> __device__ inline T rep_sum(T v, int R) {
    if (R >= 2) v += __shfl_xor(v, 32);
    if (R == 4) v += __shfl_xor(v, 16);
    return v;
}
#if !defined(MW_REMAP_MIN_ROWS)
#define MW_REMAP_MIN_ROWS 40          // (measured at MT50 @ 4096 fp64: 24 / 40 / 64 -> see DESIGN.md 5)
#endif
constexpr int REMAP_MIN_ROWS = MW_REMAP_MIN_ROWS;          // solve_wave re-assigns the blocks of finished environments only when an iterating one has at least this many rows
__device__ inline int blk_max4(int v) {          // maximum of the four blocks' (block-uniform) values, wave-uniform
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

template <typename T, bool BORDER>          // BORDER: nv = 17
MW_STAGE_FN void solve_wave(const Env<T> e_) {
    typedef float HT;
    const Env<T> e = e_.uniform();
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const int nv = e.nv, nv16 = nv < 16 ? nv : 16, nslot = mw_uniform(e.nslot);
    const int lane = e.thr & 63, rb = lane >> 4, ri = lane & 15;
    const bool dof = ri < nv16;                      // this lane carries a dof of its block's environment
    const int kd = dof ? ri : 0;
    const T scale = 1 / (m.meaninertia * T(nv > 1 ? nv : 1)), tol = m.tolerance;
    const int max_iter = m.sz.iterations, max_ls = m.sz.ls_iterations;
    const int lds_rows = e.lds_rows, stride = e.lds_stride;
    const bool tri_in_lds = e.lds_perm >= nv16 * (nv16 - 1) / 2;          // finish(): the transposed factor goes through the slots in front of the rows (else by DPP broadcasts)
    const unsigned m0 = rb == 0 ? ~0u : 0u, m1 = rb == 1 ? ~0u : 0u, m2 = rb == 2 ? ~0u : 0u, m3 = rb == 3 ? ~0u : 0u;
    for (int g0 = 0; g0 < nslot; g0 += 4) {
        // REPLICAS: when the group of this pass holds only one or two environments (lpb = 1 / 2, or the tail of a larger workgroup), R = 4 / 2
        // blocks work for each of them: everything that is per dof (vectors, M x, the Newton direction -- the matrix instruction
        // processes four blocks in the time of one anyway) is computed redundantly with identical bits, the row / block sweeps are
        // split over the 16 R lanes (wide lane index w), their totals take one more exchange between the replicas (rep_sum)
        // The assignment can CHANGE during the solve (remap, below): when all but one or two environments of the group have converged, their
        // blocks become replicas of the ones that are still iterating.
        const int left = nslot - g0;
        int R = left == 1 ? 4 : (left == 2 ? 2 : 1), npe = 4 / R;
        int smap[4];                                   // (wave-uniform) slot of the environment block b works for, -1: none
#pragma unroll
        for (int b = 0; b < 4; b++) smap[b] = g0 + (b & (npe - 1)) < nslot ? g0 + (b & (npe - 1)) : -1;
        int slot, rep, w, W, nefc, nblk, nmax, bmax;
        bool on;
        Env<T> rv = e;
        MW_TICK(t_a)
        // ---- once per solve (and again after a remap): row ri of M, the border, the input vectors, M in the accumulator layout ----
        T Mrow[16], Mc16 = 0, M16row[16], m1616 = 1;
        T qs, sm, ws, qs16 = 0, sm16 = 0, ws16 = 0;
        mw_f16v accM;                                  // lane (rb, ri), register 4 blk + v  =  M_blk[4 rb + v][ri]  (identity outside nv)
        HT hbM = 0, etaM = 1;                          // border of the Hessian: H[16][ri], H[16][16]
        auto setup = [&]() __attribute__((always_inline)) {
            const int mine = rb == 0 ? smap[0] : (rb == 1 ? smap[1] : (rb == 2 ? smap[2] : smap[3]));
            on = mine >= 0;
            slot = on ? mine : e.slot;                 // (an idle block looks at its own thread's environment and never stores)
            rep = rb / npe; w = 16 * rep + ri; W = 16 * R;
            rv = env_view(e, slot);
            nefc = on ? (int)rv.I(L.icount + 1) : 0; nblk = on ? (int)rv.I(L.icount + IC_NBLK) : 0;
            nmax = blk_max4(nefc); bmax = blk_max4(nblk);
            {
                T mv[16];
#pragma unroll
                for (int k = 0; k < 16; k++) mv[k] = rv.R(L.qM + kd * nv + (k < nv16 ? k : 0));
#pragma unroll
                for (int k = 0; k < 16; k++) Mrow[k] = (dof && k < nv16) ? mv[k] : T(0);
            }
            if (BORDER) {
                const T c16 = rv.R(L.qM + kd * nv + 16), d16 = rv.R(L.qM + 16 * nv + 16);
                T mv[16];
#pragma unroll
                for (int k = 0; k < 16; k++) mv[k] = rv.R(L.qM + 16 * nv + k);
#pragma unroll
                for (int k = 0; k < 16; k++) M16row[k] = mv[k];
                Mc16 = dof ? c16 : T(0); m1616 = d16;
            }
            {
                const T a = rv.R(L.qacc_smooth + kd), b = rv.R(L.smooth + kd), c = rv.R(L.warm + kd);
                qs = dof ? a : T(0); sm = dof ? b : T(0); ws = dof ? c : T(0);
                if (BORDER) { qs16 = rv.R(L.qacc_smooth + 16); sm16 = rv.R(L.smooth + 16); ws16 = rv.R(L.warm + 16); }
            }
            {          // all sixteen loads issued together
                T mv[16];
                bool ins[16];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int blk = q >> 2, v = q & 3, s2 = smap[blk];
                    const bool on2 = s2 >= 0;
                    const int mrow = 4 * rb + v, hi = mrow > ri ? mrow : ri, lo = mrow > ri ? ri : mrow;
                    ins[q] = on2 && hi < nv16;
                    const int idx = L.qM + (ins[q] ? hi * nv + lo : 0);
                    mv[q] = ((MW_GLOBAL T*)(e.col + ((on2 ? s2 : e.slot) - e.slot)))[(unsigned)idx * e.stride];
                }
#pragma unroll
                for (int q = 0; q < 16; q++) accM[q] = ins[q] ? (HT)mv[q] : ((4 * rb + (q & 3)) == ri ? HT(1) : HT(0));
                if (BORDER) { hbM = on ? (HT)Mc16 : HT(0); etaM = on ? (HT)m1616 : HT(1); }
            }
        };
        setup();
        // ---- state of the solve, one dof per lane ----
        T qa = 0, Ma = 0, qfc = 0, qa16 = 0, Ma16 = 0, qfc16 = 0;
        T cost = 0, cs = 0;
        bool act = on, redo = false;
        int niter = 0, nstall = 0;
        // An environment that is through: its results -- qacc, qfrc_constraint, the iteration count, the stall count, efc_force of the rows
        // kept in the scratchpad -> efcX -- and the acceleration of the semi-implicit Euler step, (M + h B) a = qfrc_smooth + qfrc_constraint
        // (integrate_impl, mw_phys.hpp), which is solved HERE, rows of M + h B one per lane, instead of by every sub-lane of the environment
        // on its own copy of the 120-entry triangle: right-looking Cholesky with the column's multipliers handed round by DPP broadcasts,
        // forward substitution the same way, the factor transposed through the scratchpad slots in front of the rows, backward
        // substitution as a chain.  The operations and their order are chol_reg's / chol_solve_reg's (products subtracted in ascending
        // k, no contraction): the same bits as integrate_impl computes.  integrate_impl finds the result in L.search (IC_EULER_READY).
        auto finish = [&](bool who) __attribute__((always_inline)) {
            if (who) {
                if (dof) { rv.R(L.qacc + ri) = qa; rv.R(L.qfrc_c + ri) = qfc; }
                if (BORDER && ri == 0) { rv.R(L.qacc + 16) = qa16; rv.R(L.qfrc_c + 16) = qfc16; }
                if (ri == 0) {
                    rv.I(L.icount + 2) = niter;
                    if (nstall) rv.I(L.icount + IC_SOLVER_STALL) += nstall;
                }
                {
                    const int nl = nefc < lds_rows ? nefc : lds_rows;
                    for (int i = ri; i < nl; i += 16) EX(rv, i, 5) = rv.lds[rv.S(i, SR_FORCE) * stride];
                }
                MW_SYNC();
              {
                const T h = m.timestep, dmp = m.dof_damping[kd];
                T A[16], A16[16], a1616 = 1, inv_own = 1, inv16 = 1;
#pragma unroll
                for (int k = 0; k < 16; k++) A[k] = ri == k ? (dof ? Mrow[k] + h * dmp : T(1)) : Mrow[k];
                if (BORDER) {
#pragma unroll
                    for (int k = 0; k < 16; k++) A16[k] = M16row[k];
                    a1616 = m1616 + h * m.dof_damping[16];
                }
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const T sj = blk_bcast_t(A[j], j);
                    const T ljj = mw_sqrt(sj < T(1e-15) ? T(1e-15) : sj), invj = T(1) / ljj;
                    A[j] = ri == j ? ljj : A[j] * invj;
                    inv_own = ri == j ? invj : inv_own;
                    T l16j = 0;
                    if (BORDER) { A16[j] = A16[j] * invj; l16j = A16[j]; }
#pragma unroll
                    for (int c = j + 1; c < 16; c++) {
                        const T v = blk_bcast_t(A[j], c);
                        A[c] -= A[j] * v;
                        if (BORDER) A16[c] -= l16j * v;
                    }
                    if (BORDER) a1616 -= l16j * l16j;
                }
                if (BORDER) { const T l = mw_sqrt(a1616 < T(1e-15) ? T(1e-15) : a1616); inv16 = T(1) / l; }
                // forward substitution: lane k's residual is final when its turn comes
                T sres = dof ? sm + qfc : T(0), s16 = BORDER ? sm16 + qfc16 : T(0), yo = 0, xs[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const T xk = blk_bcast_t(sres * inv_own, k);
                    xs[k] = xk;
                    yo = ri == k ? xk : yo;
                    sres -= A[k] * xk;
                    if (BORDER) s16 -= A16[k] * xk;
                }
                const T x16 = BORDER ? s16 * inv16 * inv16 : T(0);          // (y16 = s16 inv16; x16 = y16 inv16)
                // the factor transposed: Lt[k] = L[k][ri] for k > ri (the strict lower triangle, 120 entries).  Through the slots in
                // front of the rows when the environment has them (cdof / qvel / qpos copies, 7 nv + nq >= nv (nv - 1) / 2 slots: dead once the
                // constraint rows are built; the rows themselves stay intact for mw_read / mirror_rows), else by DPP broadcasts.
                T Lt[16], A16i = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) Lt[k] = 0;
                if (tri_in_lds) {
                    MW_LDS T* tp = rv.lds;
#pragma unroll
                    for (int k = 0; k < 15; k++)
                        if (k < ri && dof) tp[(ri * (ri - 1) / 2 + k) * stride] = A[k];          // lane ri's row: L[ri][k], k < ri  (rows >= nv are identity: their entries stay 0)
                    MW_SYNC();
#pragma unroll
                    for (int k = 1; k < 16; k++) {
                        const T v = tp[((k < nv16 ? k * (k - 1) / 2 : 0) + (ri < k ? ri : 0)) * stride];
                        Lt[k] = (ri < k && k < nv16) ? v : T(0);
                    }
                } else {
#pragma unroll
                    for (int k = 1; k < 16; k++)
#pragma unroll
                        for (int i = 0; i < k; i++) {
                            const T u = blk_bcast_t(A[i], k);          // lane k's L[k][i] to the block; lane i keeps it
                            Lt[k] = ri == i ? u : Lt[k];
                        }
                }
                if (BORDER) {
#pragma unroll
                    for (int k = 0; k < 16; k++) A16i = ri == k ? A16[k] : A16i;
                }
                // backward substitution: x_i = (y_i - sum_{k > i} L[k][i] x_k) / L[i][i], the sum in ascending k (the border last)
                T xo = 0;
#pragma unroll
                for (int i = 15; i >= 0; i--) {
                    T t = yo;
#pragma unroll
                    for (int k = i + 1; k < 16; k++) t -= Lt[k] * xs[k];
                    if (BORDER) t -= A16i * x16;
                    const T xi = blk_bcast_t(t * inv_own, i);
                    xs[i] = xi;
                    xo = ri == i ? xi : xo;
                }
                if (dof) rv.R(L.search + ri) = xo;
                if (BORDER && ri == 0) rv.R(L.search + 16) = x16;
                if (ri == 0) rv.I(L.icount + IC_EULER_READY) = 1;
              }
            }
        };
        MW_TICK(t_b)
        int it = -3;
        for (;;) {          // (one trip per assignment of the blocks: the iteration loop leaves either for good or with a remap request)
        bool want_remap = false;
        unsigned long long remap_mask = 0ull;
        for (; it < max_iter; it++) {
            MW_TICK(t_0)
            bool go;                                   // (block-uniform) this environment takes part in this pass
            T x, x16 = 0;                              // the vector of this pass: a candidate point (it < 0) or the search direction
            if (it < 0) {
                // warm start: qacc_smooth, then qacc_warmstart; the better one is kept (ties go to the warm start; an environment without
                // rows takes qacc_smooth, whatever the rounding of the two costs says), i.e. pass -1 repeats the first point where it won
                if (it == -1) { redo = on && (cost > cs || nefc == 0); if (!mw_any(redo)) continue; }
                go = it == -1 ? redo : on;
                x = it == -2 ? ws : qs;
                if (BORDER) x16 = it == -2 ? ws16 : qs16;
            } else {
                // ---- gradient and convergence test ----
                T g = dof ? Ma - sm - qfc : T(0), g16 = BORDER ? Ma16 - sm16 - qfc16 : T(0);
                const T gn = blk_sum_t(g * g) + g16 * g16;
                if (act && scale * mw_sqrt(gn) < tol) act = false;
                if (!mw_any(act)) break;
                // ---- REMAP request: one or two environments are left and they are big -> the blocks of the finished ones become their
                // replicas (below, after the finished ones have handed in their results) ----
                {
                    const unsigned long long am = __builtin_amdgcn_ballot_w64(act && rep == 0 && ri == 0);          // one bit per iterating environment, at lane 16 b of its owner block b
                    const int nact = __builtin_popcountll(am), newR = nact == 1 ? 4 : (nact == 2 ? 2 : 1);
                    if (newR > R && blk_max4(act ? nefc : 0) >= REMAP_MIN_ROWS) { want_remap = true; remap_mask = am; break; }
                }
                go = act;
                // ---- Newton direction s = -H^-1 g (the wave-cooperative direction of rounds 4-5, newton_direction_wave; its description:) ----
                // The same s = -H^-1 g, H = M + J' D J (+ cone blocks), in single precision, computed by the WHOLE WAVE for the workgroup's
                // environments four at a time instead of by every sub-lane of an environment redundantly:
                //  * lane role: lane 16 b + i works for environment (group start + b) and dof i (any lane can address any environment's columns
                //    and scratchpad slice: env_view);
                //  * H is accumulated by v_mfma_f32_16x16x1_4b_f32 -- four independent 16 x 16 rank-1 updates per instruction, one per
                //    environment -- in ONE pass over the constraint rows r = 0 .. nefc-1 (r is wave-uniform, so is the scratchpad / column-store
                //    decision of every access): a quadratic row is the term (D j_r) (x) j_r, one scratchpad read of j per lane.  (Rounds 1-3:
                //    every sub-lane zeroed a 120-153-entry triangle, added its rows with 120-153 FMAs each, and a 4-stage butterfly summed the
                //    triangles: ~4 k wave-instructions per iteration whatever the number of rows.)  Exact f32 products and sums (the f32 MFMA is
                //    an fmaf chain);
                //  * a cone block (state S_CONE, dim rows) is dim + 1 rank-1 terms.  With p_r = sqrt(Dm) fri_r j_r, q^ = (mu / Tn) sum_{r>=1} U_r p_r,
                //    rho = N / (mu Tn), dg = mu^2 - mu N / Tn > 0:   J' Hc J = (p_0 - q^) p_0' + (rho q^ - p_0) q^' + dg sum_{r>=1} p_r p_r'
                //    (the cone Hessian of the per-environment routine, regrouped).  The 2 dim scalars -- w_0 = sqrt(Dm) fri_0 and rho on row 0,
                //    c_r = (mu / Tn) U_r sqrt(Dm) fri_r and g_r = sqrt(Dm dg) fri_r on row r -- are computed in T by the environment's own
                //    sub-lanes (block-parallel, as before) and parked in the rows' AREF and JV fields, which are dead between the warm start
                //    and the line search; the role lanes read them with the row;
                //  * right-looking Cholesky in the accumulator layout: step k fetches row k of the current matrix (one round of lane permutes),
                //    scales it and removes its outer product with one more matrix instruction; lane 16 b + n ends up with row n of the factor;
                //  * the two triangular solves run on that distribution (forward: a lane broadcast + one FMA per step; backward: a 16-lane
                //    DPP sum per step);
                //  * nv = 17 (the stick scenes): the 17th dof is a border -- H = [H16 h; h' eta], factor = [L16 0; l' lam], l = L16^-1 h.
                // Results go to L.search of every active environment.  Called by EVERY lane of the wave (ghost lanes included, Env::ghost) with
                // its environment's `active` flag; environments that are not active are skipped (their lanes help with the others).
                // coefficients of the rank-1 terms, block per lane, parked in the rows' JV / AREF fields (dead here)
                if (go) {
                    auto coef = [&](const auto& rows, int i, int st, int info) {          // (rows: the block's rows in one place, Rows<T, MODE>)
                        const int dim = (info >> 4) & 15;
                        if (st == S_CONE) {
                            ConeEval<T> z = cone_eval<T>(rows, i, dim, T(0));
                            const T Dm = z.D[0] / (z.mu * z.mu * (1 + z.mu * z.mu)), kap = z.mu / z.Tn;
                            const T dg0 = z.mu * z.mu - z.mu * z.N / z.Tn, dg = dg0 > 0 ? dg0 : T(0), sDm = mw_sqrt(Dm);
                            rows.set(i, SR_JV, Dm * z.fri[0] * z.fri[0]); rows.set(i, SR_AREF, z.N / (z.mu * z.Tn));
#pragma unroll
                            for (int r = 1; r < 4; r++)
                                if (r < dim) {
                                    const T g2 = Dm * dg * z.fri[r] * z.fri[r];
                                    rows.set(i + r, SR_JV, r == dim - 1 ? -(g2 > T(1e-30) ? g2 : T(1e-30)) : g2);
                                    rows.set(i + r, SR_AREF, kap * z.U[r] * sDm * z.fri[r]);
                                }
                        } else {          // up to four independent rows: all reads, then the writes
                            const int nr = (info & 15) == C_CONTACT ? dim : 1;
                            T dd[4];
#pragma unroll
                            for (int r = 0; r < 4; r++) dd[r] = rows.get(i + (r < nr ? r : 0), SR_D);
#pragma unroll
                            for (int r = 0; r < 4; r++)
                                if (r < nr) rows.set(i + r, SR_JV, st == S_SATISFIED ? T(0) : dd[r]);
                        }
                    };
                    for (int kb = w; kb < nblk; kb += W) {
                        const int i = block_row(rv, kb);
                        const int st = (int)sr_get(rv, i, SR_STATE), info = (int)sr_get(rv, i, SR_INFO);
                        if (i + 4 <= lds_rows) coef(Rows<T, 1>{rv}, i, st, info);
                        else if (i >= lds_rows) coef(Rows<T, 2>{rv}, i, st, info);
                        else coef(Rows<T, 0>{rv}, i, st, info);
                    }
                }
                MW_SYNC();
                MW_TICK(t_pre)
                SW_FINE(1, t_0, t_pre)
                // REPLICAS share the pass over the rows: replica k takes the row batches k, k + R, ...; its matrix (the accumulator registers
                // of ITS block) starts from M in replica 0 and from zero in the others, and the partial matrices of an environment are
                // added afterwards -- inside every lane: matrix b lives in registers 4 b .. 4 b + 3 of all 64 lanes
                mw_f16v acc;
#pragma unroll
                for (int q = 0; q < 16; q++) acc[q] = (q >> 2) / npe == 0 ? accM[q] : HT(0);
                HT hb = rep == 0 ? hbM : HT(0), eta = rep == 0 ? etaM : HT(0);
                const int ne = go ? nefc : 0, nm = blk_max4(ne);
                auto cone_tail = [&](int r, bool flag, const auto& rows) {          // wave-uniform call; lanes with flag set finish the cone block that ends at row r
                    HT A0 = 0, B0 = 0, A1 = 0, B1 = 0, h0 = 0, e0 = 0;
                    if (flag) {
                        const int dim = ((int)rows.get(r, SR_INFO) >> 4) & 15, r0 = r - dim + 1;
                        T cw[4], cj[4], cj16[4];          // all reads of the block first: row r0's coefficient and rho, the other rows' c_r; the entries of this lane's dof
                        const T rho_ = rows.get(r0, SR_AREF);
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const int rr = r0 + (c < dim ? c : 0);
                            cw[c] = rows.get(rr, c == 0 ? SR_JV : SR_AREF); cj[c] = rows.getj(rr, kd);
                            cj16[c] = BORDER ? rows.getj(rr, 16) : T(0);
                        }
                        const HT w0 = sqrtf((HT)cw[0]), rho = (HT)rho_;
                        const HT P0 = dof ? w0 * (HT)cj[0] : HT(0), P016 = BORDER ? w0 * (HT)cj16[0] : HT(0);
                        HT Qh = 0, Qh16 = 0;
#pragma unroll
                        for (int c = 1; c < 4; c++)
                            if (c < dim) {
                                const HT cc = (HT)cw[c];
                                if (dof) Qh += cc * (HT)cj[c];
                                if (BORDER) Qh16 += cc * (HT)cj16[c];
                            }
                        A0 = -Qh; B0 = P0; A1 = rho * Qh - P0; B1 = Qh;
                        h0 = A0 * P016 + A1 * Qh16; e0 = -Qh16 * P016 + (rho * Qh16 - P016) * Qh16;
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(A0, B0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(A1, B1, acc, 0, 0, 0);
                    if (BORDER) { hb += h0; eta += e0; }
                };
                auto row_term = [&](int r, bool valid, bool lds_site, T t_c, T t_j, T t_j16) {          // (called by every lane of the wave: matrix instructions inside)
                    const bool in = valid && r < ne;
                    const HT cf = in ? (HT)t_c : HT(0), jr = (in && dof) ? (HT)t_j : HT(0), j16 = (BORDER && in) ? (HT)t_j16 : HT(0);
                    const HT A = fabsf(cf) * jr;
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(A, jr, acc, 0, 0, 0);
                    if (BORDER) { hb += A * j16; eta += fabsf(cf) * j16 * j16; }
                    const bool flag = cf < HT(0);
                    if (mw_any(flag)) {          // (where the block's rows -- they end at r and are at most four -- live is decided for the whole wave)
                        if (lds_site) cone_tail(r, flag, Rows<T, 1>{rv});
                        else if (!mw_any(flag && r - 3 < lds_rows)) cone_tail(r, flag, Rows<T, 2>{rv});
                        else cone_tail(r, flag, Rows<T, 0>{rv});
                    }
                };
                {
                    // one pass over the rows: the scratchpad rows eight at a time (all reads of a batch in flight together; a dependent
                    // batch costs one LDS round trip whatever its width), the rows beyond it from the column store sixteen at a time.  The
                    // last batch of either kind is padded with repeats of its last row, masked out in row_term (`ok`).
                    const int nl = nm < lds_rows ? nm : lds_rows;
                    for (int base = 0; base < nl; base += 8 * R) {
                        const int r = base + 8 * rep;
                        T tc[8], tj[8], tj16[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            MW_LDS T* p = rv.lds + rv.S(r + q < nl ? r + q : nl - 1, 0) * stride;
                            tc[q] = p[SR_JV * stride]; tj[q] = p[(SR_N + kd) * stride];
                            tj16[q] = BORDER ? p[(SR_N + 16) * stride] : T(0);
                        }
#pragma unroll
                        for (int q = 0; q < 8; q++)
                            if (base + q < nl) row_term(r + q, r + q < nl, true, tc[q], tj[q], tj16[q]);          // (wave-uniform skip of the padding)
                    }
                    for (int base = nl; base < nm; base += 16 * R) {          // (sixteen rows per trip: a trip is one L2 round trip whatever its width)
                        const int r = base + 16 * rep;
                        T tc[16], tj[16], tj16[16];
#pragma unroll
                        for (int q = 0; q < 16; q++) {
                            const int rr = r + q < nm ? r + q : nm - 1;
                            tc[q] = EX(rv, rr, sr_slot(SR_JV)); tj[q] = EJ(rv, rr, kd);
                            tj16[q] = BORDER ? T(EJ(rv, rr, 16)) : T(0);
                        }
#pragma unroll
                        for (int q = 0; q < 16; q++)
                            if (base + q < nm) row_term(r + q, r + q < nm, false, tc[q], tj[q], tj16[q]);
                    }
                }
                if (R > 1) {          // the partial matrices of an environment -> their sum, in every replica's registers (same bits: a + b = b + a)
                    mw_f16v t = acc;
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const int blk = q >> 2, v = q & 3;
                        acc[q] = R == 2 ? t[q] + t[4 * (blk ^ 2) + v] : (t[4 * (blk & 1) + v] + t[4 * ((blk & 1) ^ 2) + v]) + (t[4 * ((blk & 1) ^ 1) + v] + t[4 * ((blk & 1) ^ 3) + v]);
                    }
                    if (BORDER) { hb = rep_sum(hb, R); eta = rep_sum(eta, R); }
                }
                MW_TICK(t_rows)
                SW_COARSE(if (on) { MW_TOCK(rv, L, 1, t_0, t_rows) })
                SW_FINE(2, t_pre, t_rows)
                // Cholesky in the accumulator layout: lane (rb, ri) collects row ri of the factor of ITS environment
                HT Lr[16], invd = 1;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int src = 16 * (k / 4) + ri;
                    const unsigned u0 = __builtin_bit_cast(unsigned, __shfl(acc[0 + k % 4], src)), u1 = __builtin_bit_cast(unsigned, __shfl(acc[4 + k % 4], src)),
                                   u2 = __builtin_bit_cast(unsigned, __shfl(acc[8 + k % 4], src)), u3 = __builtin_bit_cast(unsigned, __shfl(acc[12 + k % 4], src));
                    const HT hk = __builtin_bit_cast(float, (u0 & m0) | (u1 & m1) | (u2 & m2) | (u3 & m3));
                    const HT d = fmaxf(blk_bcast(hk, k), HT(1e-15));
                    const HT rs = __builtin_amdgcn_rsqf(d);
                    const unsigned keep = (unsigned)((k - 1 - ri) >> 31);
                    const HT l = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, (ri == k ? d : hk) * rs) & keep);
                    Lr[k] = l;
                    invd = ri == k ? rs : invd;
                    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(-l, l, acc, 0, 0, 0);
                }
                HT yg = (go && dof) ? (HT)(-g) : HT(0), yh = hb;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const HT bg = blk_bcast(yg * invd, k);
                    const HT ng = yg - Lr[k] * bg;
                    yg = ri == k ? bg : ng;
                    if (BORDER) {
                        const HT bh = blk_bcast(yh * invd, k);
                        const HT nh = yh - Lr[k] * bh;
                        yh = ri == k ? bh : nh;
                    }
                }
                HT x17 = 0;
                if (BORDER) {
                    const HT lam2 = fmaxf(eta - blk_sum(yh * yh), HT(1e-15));
                    const HT il = __builtin_amdgcn_rsqf(lam2);
                    const HT y17 = ((go ? (HT)(-g16) : HT(0)) - blk_sum(yh * yg)) * il;
                    x17 = y17 * il;
                    yg -= yh * x17;
                }
                HT xd = 0;
#pragma unroll
                for (int k = 15; k >= 0; k--) {
                    const HT sum = blk_sum(Lr[k] * xd);
                    const HT xk = (yg - sum) * invd;
                    xd = ri == k ? xk : xd;
                }
                x = dof ? (T)xd : T(0);
                if (BORDER) x16 = (T)x17;
                MW_TICK(t_chol)
                SW_COARSE(if (on) { MW_TOCK(rv, L, 2, t_rows, t_chol) })
                SW_FINE(3, t_rows, t_chol)
            }
            MW_TICK(t_1)
            // ---- apply: y = M x (row per lane on the gathered x), rows[field] = J x (- aref at a candidate point), row per lane ----
            T y, y16 = 0;
            {
                T xs[16];
#pragma unroll
                for (int k = 0; k < 16; k++) xs[k] = blk_bcast_t(x, k);
                y = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) y += Mrow[k] * xs[k];
                if (BORDER) {
                    y += Mc16 * x16;
#pragma unroll
                    for (int k = 0; k < 16; k++) y16 += M16row[k] * xs[k];
                    y16 += m1616 * x16;
                }
                const bool point = it < 0;
                for (int r0 = 0; r0 < nmax; r0 += W) {
                    const int r = r0 + w;
                    if (go && r < nefc) {
                        T j[16], j16 = 0, s;
                        if (r < lds_rows) {
                            MW_LDS T* p = rv.lds + rv.S(r, 0) * stride;
#pragma unroll
                            for (int k = 0; k < 16; k++) j[k] = p[(SR_N + (k < nv16 ? k : 0)) * stride];
                            if (BORDER) j16 = p[(SR_N + 16) * stride];
                            s = point ? -p[SR_AREF * stride] : T(0);
                        } else {
#pragma unroll
                            for (int k = 0; k < 16; k++) j[k] = EJ(rv, r, k < nv16 ? k : 0);
                            if (BORDER) j16 = EJ(rv, r, 16);
                            s = point ? -EX(rv, r, sr_slot(SR_AREF)) : T(0);
                        }
#pragma unroll
                        for (int k = 0; k < 16; k++) s += j[k] * xs[k];          // (xs is 0 beyond nv)
                        if (BORDER) s += j16 * x16;
                        sr_set(rv, r, point ? SR_JAR : SR_JV, s);
                    }
                }
            }
            MW_SYNC();
            MW_TICK(t_2)
            SW_FINE(4, t_1, t_2)
            if (it < 0) {
                if (go) { qa = x; Ma = y; qa16 = x16; Ma16 = y16; }
            } else {
                SW_COARSE(if (on) { MW_TOCK(rv, L, 3, t_1, t_2) })
                // ---- exact line search along x (safeguarded Newton on the 1-D convex cost) ----
                const T r_ = dof ? Ma - sm : T(0), dq = dof ? qa - qs : T(0), r16 = BORDER ? Ma16 - sm16 : T(0), dq16 = BORDER ? qa16 - qs16 : T(0);
                const T snorm = mw_sqrt(blk_sum_t(x * x) + x16 * x16);
                T quadGauss[3];
                quadGauss[0] = blk_sum_t(T(0.5) * r_ * dq) + T(0.5) * r16 * dq16;
                quadGauss[1] = blk_sum_t(x * r_) + x16 * r16;
                quadGauss[2] = blk_sum_t(T(0.5) * x * y) + T(0.5) * x16 * y16;
                if (go && snorm < T(1e-15)) { go = false; act = false; }
                const T gtol = tol * T(0.01) * snorm / scale;
                T alpha = 0, lo = 0, hi = -1;
                bool ls = go;                          // the search of this block is still running
                int nls = 0;
                for (int lit = -1; lit < max_ls; lit++) {
                    if (!mw_any(ls)) break;
                    // constraint part of the cost at jar + alpha Jv with its derivatives, block per lane
                    T C = 0, D1 = 0, D2 = 0, A1 = 0;
                    for (int kb0 = 0; kb0 < bmax; kb0 += W) {
                        const int kb = kb0 + w;
                        if (ls && kb < nblk) {
                            const int i = block_row(rv, kb);
                            if (i + 4 <= lds_rows) le_row<T>(Rows<T, 1>{rv}, i, alpha, &C, &D1, &D2, &A1);
                            else if (i >= lds_rows) le_row<T>(Rows<T, 2>{rv}, i, alpha, &C, &D1, &D2, &A1);
                            else le_row<T>(Rows<T, 0>{rv}, i, alpha, &C, &D1, &D2, &A1);
                        }
                    }
                    C = rep_sum(blk_sum_t(C), R); D1 = rep_sum(blk_sum_t(D1), R); D2 = rep_sum(blk_sum_t(D2), R); A1 = rep_sum(blk_sum_t(A1), R);
                    const T da = D1 + (2 * alpha * quadGauss[2] + quadGauss[1]), dda = D2 + 2 * quadGauss[2];
                    const T mag = A1 + mw_abs(2 * alpha * quadGauss[2]) + mw_abs(quadGauss[1]);
                    (void)C;
                    if (!ls) continue;
                    if (lit < 0) {                     // the evaluation at alpha = 0: first Newton step of the search
                        if (da >= 0 || dda <= 0) { nstall++; ls = false; go = false; act = false; }   // not a descent direction: the search is abandoned, and counted (mw_status)
                        else alpha = -da / dda;
                        continue;
                    }
                    nls++;
                    if (mw_abs(da) < gtol) { ls = false; continue; }
                    if (sizeof(T) == 4 && mw_abs(da) <= T(1e-6) * mag) { ls = false; continue; }
                    if (da < 0) lo = alpha; else hi = alpha;
                    T an = alpha - da / dda;
                    if (hi < 0) { if (an <= lo) an = 2 * alpha; }
                    else if (!(an > lo && an < hi)) an = T(0.5) * (lo + hi);
                    if (hi > 0 && (hi - lo) <= (sizeof(T) == 8 ? T(1e-16) : T(5e-7)) * hi) { alpha = T(0.5) * (lo + hi); ls = false; continue; }
                    if (an == alpha) { ls = false; continue; }
                    alpha = an;
                }
                MW_TICK(t_3)
                SW_COARSE(if (on) { MW_TOCK(rv, L, 4, t_2, t_3) MW_TADD(rv, L, 6, nls) })
                SW_FINE(5, t_2, t_3)
                (void)nls;
                if (go && alpha == 0) { go = false; act = false; }
                if (go) { qa += alpha * x; Ma += alpha * y; if (BORDER) { qa16 += alpha * x16; Ma16 += alpha * y16; } }
                for (int r0 = 0; r0 < nmax; r0 += W) {
                    const int r = r0 + w;
                    if (go && r < nefc) sr_set(rv, r, SR_JAR, sr_get(rv, r, SR_JAR) + alpha * sr_get(rv, r, SR_JV));
                }
                MW_SYNC();
            }
            MW_TICK(t_4)
            // ---- update_constraint at the current jar: force / state / cost block per lane, Gauss term, qfrc_constraint = J' force ----
            T cnew;
            {
                T c = 0;
                for (int kb0 = 0; kb0 < bmax; kb0 += W) {
                    const int kb = kb0 + w;
                    if (go && kb < nblk) {
                        const int i = block_row(rv, kb);
                        if (i + 4 <= lds_rows) uc_row<T>(Rows<T, 1>{rv}, rv, i, &c);
                        else if (i >= lds_rows) uc_row<T>(Rows<T, 2>{rv}, rv, i, &c);
                        else uc_row<T>(Rows<T, 0>{rv}, rv, i, &c);
                    }
                }
                c = rep_sum(blk_sum_t(c), R);
                const T gs = blk_sum_t(dof ? (Ma - sm) * (qa - qs) : T(0)) + (BORDER ? (Ma16 - sm16) * (qa16 - qs16) : T(0));
                cnew = c + T(0.5) * gs;
                MW_SYNC();
                MW_TICK(t_uc)
                SW_FINE(6, t_4, t_uc)
                // J' f, dof per lane: every lane of the block reads the row's force (one address: an LDS broadcast) and its own entry
                const int ne = go ? nefc : 0, nm = blk_max4(ne), nl = nm < lds_rows ? nm : lds_rows;
                // (replica k takes the row batches k, k + R, ...; the partial sums meet in rep_sum; batches as in the H pass: eight
                //  scratchpad rows / sixteen column-store rows per trip, the last one padded and masked)
                T q = 0, q16 = 0;
                for (int r = 8 * rep; r < nl; r += 8 * R) {
                    T tf[8], tj[8], tj16[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        MW_LDS T* p = rv.lds + rv.S(r + u < nl ? r + u : nl - 1, 0) * stride;
                        tf[u] = p[SR_FORCE * stride]; tj[u] = p[(SR_N + kd) * stride];
                        tj16[u] = BORDER ? p[(SR_N + 16) * stride] : T(0);
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const bool in = r + u < nl && r + u < ne && tf[u] != 0;
                        q += in ? tj[u] * tf[u] : T(0);
                        if (BORDER) q16 += in ? tj16[u] * tf[u] : T(0);
                    }
                }
                for (int r = nl + 16 * rep; r < nm; r += 16 * R) {
                    T tf[16], tj[16], tj16[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const int rr = r + u < nm ? r + u : nm - 1;
                        tf[u] = EX(rv, rr, 5); tj[u] = EJ(rv, rr, kd);
                        tj16[u] = BORDER ? T(EJ(rv, rr, 16)) : T(0);
                    }
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const bool in = r + u < nm && r + u < ne && tf[u] != 0;
                        q += in ? tj[u] * tf[u] : T(0);
                        if (BORDER) q16 += in ? tj16[u] * tf[u] : T(0);
                    }
                }
                q = rep_sum(q, R);
                if (BORDER) q16 = rep_sum(q16, R);
                if (go) { qfc = dof ? q : T(0); qfc16 = q16; }
                MW_TICK(t_jf)
                SW_FINE(7, t_uc, t_jf)
            }
            MW_SYNC();
            MW_TICK(t_5)
            if (it == -3) cs = cnew;
            else if (it == -2) cost = cnew;
            else if (it == -1) { if (go) cost = cnew; }
            else {
                SW_COARSE(if (on) { MW_TOCK(rv, L, 5, t_4, t_5) MW_TADD(rv, L, 7, go ? 1 : 0) })
                if (go) {
                    niter = it + 1;
                    if (scale * (cost - cnew) < tol) act = false;
                    cost = cnew;
                }
            }
            SW_COARSE(if (it < 0 && on) { MW_TOCK(rv, L, 0, t_0, t_5) })
        }
        // ---- the environments that are through (all of them, or before a remap the finished ones) hand in their results ----
        finish(want_remap ? (on && rep == 0 && !act) : (on && rep == 0));
        if (!want_remap) break;
        MW_SYNC();
        {
            const unsigned long long am = remap_mask;
            const int nact = __builtin_popcountll(am), newR = nact == 1 ? 4 : 2;
            const int b0 = __builtin_ctzll(am) >> 4, b1 = nact == 2 ? (__builtin_ctzll(am & (am - 1)) >> 4) : b0;
            const int s0 = __builtin_amdgcn_readlane(slot, 16 * b0), s1 = __builtin_amdgcn_readlane(slot, 16 * b1);
            const int from = 16 * ((newR == 4 || !(rb & 1)) ? b0 : b1) + ri;          // the lane that holds this lane's dof of the adopted environment
            qa = __shfl(qa, from); Ma = __shfl(Ma, from); qfc = __shfl(qfc, from); cost = __shfl(cost, from);
            niter = __shfl(niter, from); nstall = __shfl(nstall, from);
            if (BORDER) { qa16 = __shfl(qa16, from); Ma16 = __shfl(Ma16, from); qfc16 = __shfl(qfc16, from); }
            act = true;
            R = newR; npe = 4 / R;
            smap[0] = s0; smap[1] = newR == 4 ? s0 : s1; smap[2] = s0; smap[3] = smap[1];
            setup();          // (the iteration the request came from starts again: gradient of the adopted environment, same bits as its owner's)
        }
        }
        SW_COARSE(if (on) { MW_TOCK(rv, L, 0, t_a, t_b) })
        MW_SYNC();
    }
}
#endif
}  // namespace mw
