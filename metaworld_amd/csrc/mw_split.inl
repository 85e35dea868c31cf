// mw_split.inl -- SPLIT COLLISION: the narrow phase of every dynamics evaluation as batch-wide kernels over (environment, candidate
// pair) work items, between the lane kernels (included by mw_runtime.hpp inside namespace mw; option "split_collision").
//
// The fused step kernel (lane_step) runs an environment's narrow phase on that environment's own sub-lanes: one 512-register wave
// per SIMD, every wave executing every geom-type branch some lane takes, lanes idle whenever an environment has fewer candidates
// than sub-lanes.  Here one VectorEnv.step is a SEQUENCE of launches on the context's stream:
//
//   lane_phase BEGIN : action -> mocap / ctrl, kinematics, mass matrix, bias forces of substep 1               (lane kernel, sub-lanes)
//   5 x { mid_phase  : one wave per environment over its static pair list -> ordered candidate list per environment (L.ipair) +
//                      one work item per candidate, appended to the work list of its geom-type CLASS            (flat, coalesced)
//         narrow     : persistent waves, each over 64 items of ONE class (type-uniform: one branch of collide_pair per wave); the
//                      hits go to the item's slot of a batch-wide hit table                                       (flat)
//         lane_phase : contacts appended in candidate order from the hit table (collision_gather: the contact list equals the
//                      fused kernel's), constraint rows, solver, Euler step; then kinematics / mass matrix / bias forces of the
//                      NEXT substep -- or, after the fifth, the final kinematics, and for the environments whose reward reads
//                      contact forces the first half of the lazy final dynamics }
//   mid_phase + narrow + lane_phase FINAL for those environments only; observation, reward, wrappers, auto-reset (step_outputs).
//
// Same stage functions on the same inputs in the same order as lane_step: results are bit-identical on the host harness
// (tests/test_split_collision.py) and within rounding of fused-multiply-add placement on the device.

constexpr int NP_NCLASS = 5;          // geom-type classes of the narrow phase: a wave of the narrow kernel serves ONE class
constexpr int NP_HIT_W = 7;           // reals per hit: dist, pos[3], normal[3]
constexpr int NP_MAXHIT = 16;         // hits per pair (the size of collide_pair's output array)
enum { SC_TOTAL = 0, SC_CLASS0 = 1, SC_OVERFLOW = 1 + NP_NCLASS, SC_WORDS = 8 };   // words of SplitBuf::counts

// class of a geom-type pair (t1 <= t2 as in the model's pair list): which branch of collide_pair the pair takes
MW_HD int np_class(int t1, int t2) {
    if (t1 == G_PLANE || t1 == G_SPHERE || (t1 == G_CAPSULE && (t2 == G_CAPSULE || t2 == G_BOX))) return 0;   // closed forms
    if (t1 == G_BOX && t2 == G_BOX) return 1;                                                                      // SAT + face clipping (thread-private polygons)
    if (t2 == G_BOX || t1 == G_BOX) return t1 == G_MESH || t2 == G_MESH ? 3 : 2;   // box face axes, then portal refinement: cylinder (2, + face upgrade) / hull (3)
    return 4;                                                                       // portal refinement only (hull-hull, cylinder-hull, capsule-hull, ...)
}

template <typename T>
struct SplitBuf {
    int* counts;        // [SC_WORDS]: items allocated by the mid phase, items per class, overflow flag (zeroed before every mid phase)
    int* item_env;      // [cap] (group << 24) | lane of the item's environment
    int* item_pair;     // [cap] index into the model's pair list
    int* class_list;    // [NP_NCLASS][cap] item ids per class
    int* hit_n;         // [cap] hits the narrow phase found for the item
    T* hits;            // [NP_MAXHIT * NP_HIT_W][cap] struct-of-arrays: component c of hit i of item k at hits[(i * NP_HIT_W + c) * cap + k]
    int cap;
};

// the environment `lane` of group g as seen from ANY thread (no scratchpad, no sub-lanes): the flat kernels' view
template <typename T>
MW_HD Env<T> env_at(const World<T>& w, int g, int lane) {
    const GroupDev<T>& G = w.groups[g];
    Env<T> e{};
    const int lpb = G.lpb, lin = lane % lpb;
    const size_t chunk = (size_t)(lane / lpb);
    e.col = G.col + chunk * G.L.nreal * lpb + lin;
    e.icol = G.icol + chunk * G.L.nint * lpb + lin;
    e.m = &G.m; e.stride = (unsigned)lpb;
    e.cache_layout(G.L, G.m.sz.nv);
    e.lds = nullptr; e.tls = nullptr; e.lds_stride = 1; e.tls_stride = 1; e.sub = 0; e.nsub = 1; e.thr = 0;
    e.slot = 0; e.nslot = 1; e.ghost = 0; e.lds_rows = 0; e.lds_w = SR_N + G.m.sz.nv; e.lds_perm = 0; e.chain_lds = 0;
#if defined(MW_BOUNDS)
    e.nreal_b = (unsigned)G.L.nreal; e.nint_b = (unsigned)G.L.nint; e.oob = w.io.status;
#endif
    return e;
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline int sc_atomic_add(int* p, int v) { return atomicAdd(p, v); }
#else
inline int sc_atomic_add(int* p, int v) { const int o = *p; *p = o + v; return o; }
#endif

// ---- mid phase: ONE WAVE PER ENVIRONMENT (device) over the model's static pair list, 64 pairs per trip; the survivors of a trip are
// compacted in pair order with a ballot, so the environment's candidate list (L.ipair) equals the fused kernel's.  Every candidate
// becomes a work item: its id goes to L.iitem (where collision_gather looks for the hits) and to the work list of its class.
// `only_pending`: the lazy final dynamics -- only environments that wait for it (IC_PENDING).  Host harness: one call per environment.
template <typename T>
MW_HD void mid_phase_env(const World<T>& w, const SplitBuf<T>& sb, int flat_env, int tid, bool only_pending) {
    int g = 0;
    while (g + 1 < w.ngroups && flat_env >= w.groups[g + 1].env0) g++;
    const int lane = flat_env - w.groups[g].env0;
    if (lane >= w.groups[g].nenv) return;
    Env<T> e = env_at(w, g, lane);
#if defined(__HIP_DEVICE_COMPILE__)
    e = e.uniform();
    e.col = (T*)mw_uniform((unsigned long long)e.col); e.icol = (int*)mw_uniform((unsigned long long)e.icol);
#endif
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    if (only_pending && !e.I(L.icount + IC_PENDING)) return;
    const int npair = m.sz.npair, packed = (g << 24) | lane;
    int ncand = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long below = tid == 0 ? 0ull : (~0ull >> (64 - tid));
    for (int p0 = 0; p0 < npair; p0 += 64) {
        const int p = p0 + tid;
        const bool ok = p < npair && pair_near(e, p);
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(ok);
        if (bal == 0ull) continue;
        const int tot = __popcll(bal), rank = __popcll(bal & below);
        int base = 0;
        if (tid == 0) base = sc_atomic_add(sb.counts + SC_TOTAL, tot);
        base = mw_uniform(base);
        int item = base + rank, cls = -1;
        if (ok) {
            if (item >= sb.cap) { item = -1; sb.counts[SC_OVERFLOW] = 1; }
            else cls = np_class(m.geom_type[m.pair_geom[2 * p]], m.geom_type[m.pair_geom[2 * p + 1]]);
        }
        for (int c = 0; c < NP_NCLASS; c++) {
            const unsigned long long bc = __builtin_amdgcn_ballot_w64(cls == c);
            if (bc == 0ull) continue;
            int cb = 0;
            if (tid == 0) cb = sc_atomic_add(sb.counts + SC_CLASS0 + c, __popcll(bc));
            cb = mw_uniform(cb);
            if (cls == c) sb.class_list[(size_t)c * sb.cap + cb + __popcll(bc & below)] = item;
        }
        if (ok) {
            e.I(L.ipair + ncand + rank) = p;
            e.I(L.iitem + ncand + rank) = item;
            if (item >= 0) { sb.item_env[item] = packed; sb.item_pair[item] = p; }
        }
        ncand += tot;
    }
    if (tid == 0) e.I(L.icount + IC_NCAND) = ncand;
#else
    (void)tid;
    for (int p = 0; p < npair; p++) {
        if (!pair_near(e, p)) continue;
        int item = sc_atomic_add(sb.counts + SC_TOTAL, 1);
        if (item >= sb.cap) { item = -1; sb.counts[SC_OVERFLOW] = 1; }
        else {
            const int cls = np_class(m.geom_type[m.pair_geom[2 * p]], m.geom_type[m.pair_geom[2 * p + 1]]);
            sb.class_list[(size_t)cls * sb.cap + sc_atomic_add(sb.counts + SC_CLASS0 + cls, 1)] = item;
            sb.item_env[item] = packed; sb.item_pair[item] = p;
        }
        e.I(L.ipair + ncand) = p;
        e.I(L.iitem + ncand) = item;
        ncand++;
    }
    e.I(L.icount + IC_NCAND) = ncand;
#endif
}

// ---- narrow phase of ONE work item: the fused kernel's collide_pair on the item's two shapes (built from the environment's geom
// frames in the column store + its model's tables), hits into the item's slot of the hit table
template <typename T>
MW_HD void narrow_item(const World<T>& w, const SplitBuf<T>& sb, int item, bool active, MW_LDS T* tls, int tls_stride) {
    const int packed = sb.item_env[item], p = sb.item_pair[item];
    const Env<T> e = env_at(w, packed >> 24, packed & 0xffffff);
    CModel<T>& m = e.model();
    const int g1 = m.pair_geom[2 * p], g2 = m.pair_geom[2 * p + 1];
    const T margin = mw_max(m.geom_margin[g1], m.geom_margin[g2]);
    Hit<T> h[NP_MAXHIT];
#if defined(MW_COLL_TIMING) && defined(MW_SOLVER_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    int tstat[4] = {0, 0, 0, 0};
#endif
    int n = collide_pair<T, false>(make_shape(e, g1), make_shape(e, g2), margin, h, active, tls, tls_stride MW_CP_PASS(tstat));
    if (!active) return;
    if (n < 0) n = 0;
    sb.hit_n[item] = n;
    for (int i = 0; i < n; i++) {
        T* o = sb.hits + (size_t)(i * NP_HIT_W) * sb.cap + item;
        o[0] = h[i].dist;
        o[(size_t)sb.cap] = h[i].pos.x; o[(size_t)2 * sb.cap] = h[i].pos.y; o[(size_t)3 * sb.cap] = h[i].pos.z;
        o[(size_t)4 * sb.cap] = h[i].normal.x; o[(size_t)5 * sb.cap] = h[i].normal.y; o[(size_t)6 * sb.cap] = h[i].normal.z;
    }
}

// persistent waves (device): virtual wave v serves 64 consecutive items of one class list; the classes are laid out one after the
// other, so all of them run concurrently and every wave is type-uniform.  Host harness: one call that walks all items in order.
template <typename T>
MW_HD void narrow_wave(const World<T>& w, const SplitBuf<T>& sb, int wave, int tid, int nwaves, MW_LDS T* lds) {
#if defined(__HIP_DEVICE_COMPILE__)
    int cnt[NP_NCLASS];
    for (int c = 0; c < NP_NCLASS; c++) {
        const int v = sb.counts[SC_CLASS0 + c];
        cnt[c] = mw_uniform(v < sb.cap ? v : sb.cap);
    }
    for (int v = wave;; v += nwaves) {
        int c = 0, first = v;
        for (; c < NP_NCLASS; c++) {
            const int nw = (cnt[c] + 63) >> 6;
            if (first < nw) break;
            first -= nw;
        }
        if (c == NP_NCLASS) break;
        const int idx = first * 64 + tid;
        const bool act = idx < cnt[c];
        const int item = sb.class_list[(size_t)c * sb.cap + (act ? idx : first * 64)];
        narrow_item(w, sb, item, act, lds + tid, 64);
    }
#else
    (void)wave; (void)tid; (void)nwaves;
    const int total = sb.counts[SC_TOTAL] < sb.cap ? sb.counts[SC_TOTAL] : sb.cap;
    for (int item = 0; item < total; item++) narrow_item(w, sb, item, true, lds, 1);
#endif
}

// ---- contacts from the hit table, appended in candidate order by the environment's sub-lanes: the second half of collision()
// (mw_collide.hpp) with "read the item's hits" in the place of "call collide_pair"
template <typename T>
MW_STAGE_FN void collision_gather(const Env<T> e_, const SplitBuf<T> sb) {
    const Env<T> e = e_.uniform();
    CModel<T>& m = e.model();
    CLayout& L = e.lay();
    const int maxcon = m.sz.maxcon, ncand = e.I(L.icount + IC_NCAND);
    int ncon = 0, flags = 0, want = 0;
    for (int c0 = 0; mw_any(c0 < ncand); c0 += e.nsub) {
        Hit<T> h[MW_NSLOT][NP_MAXHIT];
        int n[MW_NSLOT], off[MW_NSLOT], pp[MW_NSLOT];
        MW_SUBS(e, sub) {
            const bool act = c0 + sub < ncand;
            const int item = act ? e.I(L.iitem + c0 + sub) : -1;
            int cnt = 0;
            if (item >= 0) cnt = sb.hit_n[item];
            else if (act) flags |= ST_CON_OVERFLOW;          // the work-item table was full: the pair's contacts are dropped (flagged)
            Hit<T>* hh = h[MW_SLOT(sub)];
            for (int i = 0; i < cnt; i++) {
                const T* o = sb.hits + (size_t)(i * NP_HIT_W) * sb.cap + item;
                T x[NP_HIT_W];
#pragma unroll
                for (int k = 0; k < NP_HIT_W; k++) x[k] = o[(size_t)k * sb.cap];
                hh[i].dist = x[0]; hh[i].pos = v3(x[1], x[2], x[3]); hh[i].normal = v3(x[4], x[5], x[6]);
            }
            n[MW_SLOT(sub)] = cnt; pp[MW_SLOT(sub)] = act ? e.I(L.ipair + c0 + sub) : 0;
        }
        const int total = sub_scan(e, n, off);
        MW_SUBS(e, sub) {
            if (n[MW_SLOT(sub)] > 0) append_contacts(e, pp[MW_SLOT(sub)], n[MW_SLOT(sub)], h[MW_SLOT(sub)], ncon + off[MW_SLOT(sub)], maxcon);
        }
        ncon += total;
        want += total;
        if (ncon > maxcon) { ncon = maxcon; flags |= ST_CON_OVERFLOW; }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    {          // (a dropped item is seen by one sub-lane only: every sub-lane of the environment must hold the same flags)
        for (int o = e.lds_stride; o < 64; o <<= 1) flags |= __shfl_xor(flags, o);
    }
#endif
    if (sub_disagree(e, ncon) || sub_disagree(e, want) || sub_disagree(e, flags)) flags |= ST_DIVERGED;
    e.I(L.icount) = ncon;
    if (want > e.I(L.icount + IC_WANT_CON)) e.I(L.icount + IC_WANT_CON) = want;
    if (flags) e.I(L.icount + 3) |= flags;
    MW_SYNC();
}

// ---- the lane kernels between the collision kernels
enum { PH_BEGIN = 0, PH_MID = 1, PH_LAST = 2, PH_FINAL = 3 };

// The scratchpad does not survive a kernel boundary: the copies of cdof / qvel / qpos in front of the constraint rows (Env::lds_perm,
// written by the kinematics of the PREVIOUS lane kernel and read by make_constraints) are staged in again from the column store,
// which holds the same values (kinematics_impl writes both).
template <typename T>
MW_HD void restage_perm(const Env<T> e) {
    if (!e.chain_lds) return;
    CLayout& L = e.lay();
    stage_in(e, cdof_view<T, true>(e), L.cdof, 6 * e.nv);
    stage_in(e, qvel_view<T, true>(e), L.qvel, e.nv);
    stage_in(e, qpos_view<T, true>(e), L.qpos, e.model().sz.nq);
    MW_SYNC();
}

template <typename T>
MW_HD void split_finish(const World<T>& w, const Env<T> e, int gid, int task, const TaskDesc<T>& td) {
    T act[4], obs[39], reward, success;
    Info info;
    for (int k = 0; k < 4; k++) act[k] = (T)w.io.act[(size_t)gid * 4 + k];
    get_obs(e, td, obs);
    clip_obs(td, obs);
    if (w.reward_v1) task_evaluate_v1(e, td, obs, act, &reward, &success, &info);
    else task_evaluate(e, td, obs, act, &reward, &success, &info);
    step_outputs(w, e, gid, task, td, obs, reward, success, info);
}

template <typename T>
MW_HD void lane_phase(const World<T>& w, const SplitBuf<T> sb, int phase, int block, int thread, Scratchpad sp) {
    Env<T> e; int gid;
    if (!locate(w, block, thread, sp, &e, &gid)) return;
    const int task = (int)TK(e, TK_TASK);
    const TaskDesc<T>& td = w.tasks[task];
    CLayout& L = e.lay();
    if (phase == PH_BEGIN) {          // env_step up to the first narrow phase
        T act[4];
        for (int k = 0; k < 4; k++) act[k] = (T)w.io.act[(size_t)gid * 4 + k];
        e.I(L.icount + 3) = 0; e.I(L.icount + IC_SOLVER_STALL) = 0; e.I(L.icount + IC_PENDING) = 0;
        for (int k = 0; k < 3; k++) {
            const float a = fminf(fmaxf(float(act[k]), -1.0f), 1.0f);
            const float delta = a * 0.01f;
            e.R(L.mocap + k) = mw_clamp(e.R(L.mocap + k) + T(delta), td.mocap_low[k], td.mocap_high[k]);
        }
        e.R(L.ctrl) = act[3]; e.R(L.ctrl + 1) = -act[3];
        kinematics(e); crb(e); smooth_forces(e);
        return;
    }
    if (phase == PH_FINAL) {          // second half of the lazy final dynamics, then the outputs
        if (!e.I(L.icount + IC_PENDING)) return;          // (set wave-uniformly in PH_LAST)
        restage_perm(e);
        collision_gather(e, sb); make_constraints(e); solve(e);
        e.I(L.icount + IC_DYN_VALID) = 1; e.I(L.icount + IC_PENDING) = 0;
    } else {
        // PH_MID / PH_LAST: the rest of the running substep ...
        restage_perm(e);
        collision_gather(e, sb); make_constraints(e); solve(e);
        e.I(L.icount + IC_DYN_VALID) = 1;
        integrate(e);
        if (phase == PH_MID) { kinematics(e); crb(e); smooth_forces(e); return; }          // ... and the next one up to its narrow phase
        TK(e, TK_PATHLEN) += 1;
        kinematics(e);
        e.I(L.icount + IC_DYN_VALID) = 0;
        if (mw_any(w.full_forward != 0 || task_touches(td.kind))) {          // (wave-uniform, as in env_step)
            crb(e); smooth_forces(e);
            e.I(L.icount + IC_PENDING) = 1;
            return;
        }
    }
    split_finish(w, e, gid, task, td);          // (ONE call site: the observation / reward code of all tasks is inlined here)
}
