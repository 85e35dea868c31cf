// mw_runtime.hpp -- context / group / snapshot management and the lane programs that the
// kernels run.  Included by mwgpu.hip (device backend; the product) and by
// tests/host_harness.cpp (host backend; CPU test harness for the same lane code).
//
// The including file must define, before inclusion, a `Backend` struct with:
//   static void* alloc(size_t bytes);  static void free(void*);  static void zero(void*, size_t);
//   static void h2d(void* dst, const void* src, size_t);  static void d2h(void* dst, const void* src, size_t);
//   template <class F> static void launch(int nblocks, F lane_program);   // F(block, thread, Scratchpad) for 64 threads / block
//   static void sync();
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <set>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mwgpu.h"   // mw_device_out, mw_bookkeeping (the C ABI types are plain structs)
#include "mw_collide.hpp"
#include "mw_common.hpp"
#include "mw_phys.hpp"
#include "mw_policies_gen.hpp"
#include "mw_tasks.hpp"

namespace mw {

constexpr int BLOCK = 64;   // one wavefront per workgroup: spreads small batches over as many CUs as possible

// MW_VERBOSE >= 2: the address range of every device buffer, so that the address of a GPU memory access fault can be attributed
inline void mw_log_range(const char* what, int id, const void* p, size_t bytes) {
    static const int verbose = getenv("MW_VERBOSE") ? atoi(getenv("MW_VERBOSE")) : 0;
    if (verbose >= 2) fprintf(stderr, "[mwgpu] %-10s %3d  %p .. %p  (%zu bytes)\n", what, id, p, (const void*)((const char*)p + bytes), bytes);
}

// ------------------------------------------------------------------ host-side model description
struct ModelData {
    std::map<std::string, std::vector<int>> ints;
    std::map<std::string, std::vector<double>> reals;
    Sizes sz{};
    double timestep = 0.0025, tolerance = 1e-10, meaninertia = 1, gravity[3] = {0, 0, -9.81};
    double reset_tolerance = 0;   // solver tolerance of the (double precision) reset-snapshot build; 0 = same as tolerance
    int lanes_per_block = 0;      // environments per workgroup for this model's group; 0 = the runtime's own choice (finalize)
    double step_ms_lpb4 = 0, step_ms_lpb8 = 0;   // measured late-episode step time of this scene at 4 / 8 lanes per workgroup (0 = unknown)
    const std::vector<int>& I(const std::string& k) const {
        auto it = ints.find(k);
        if (it == ints.end()) throw std::runtime_error("model is missing int field " + k);
        return it->second;
    }
    const std::vector<double>& Rr(const std::string& k) const {
        auto it = reals.find(k);
        if (it == reals.end()) throw std::runtime_error("model is missing real field " + k);
        return it->second;
    }
    void finalize() {
        sz.nq = (int)Rr("qpos0").size(); sz.nv = (int)I("dof_bodyid").size(); sz.nbody = (int)I("body_parentid").size();
        sz.njnt = (int)I("jnt_type").size(); sz.ngeom = (int)I("geom_type").size(); sz.nmesh = (int)I("mesh_vertnum").size();
        sz.nmeshvert = (int)Rr("mesh_vert").size() / 3; sz.npair = (int)I("pair_geom").size() / 2;
        sz.nu = (int)I("act_dofid").size(); sz.neq = (int)I("eq_body1").size(); sz.nprobe = (int)I("probe_body").size();
        sz.nsite = 0;
        if (sz.nv > MAX_NV) throw std::runtime_error("nv exceeds MAX_NV");
        if (sz.nu != 2) throw std::runtime_error("expected the two finger position actuators");
    }
};

template <typename T, typename Backend>
struct DeviceModel {
    Model<T> m{};
    int* iblob = nullptr;
    T* rblob = nullptr;
    explicit DeviceModel(const ModelData& d) {
        struct IF { const char* name; CP<int> Model<T>::*p; };
        struct RF { const char* name; CP<T> Model<T>::*p; };
        const IF ifs[] = {{"body_parentid", &Model<T>::body_parentid}, {"body_mocap", &Model<T>::body_mocap},
            {"body_jntadr", &Model<T>::body_jntadr}, {"body_jntnum", &Model<T>::body_jntnum}, {"body_lastdof", &Model<T>::body_lastdof},
            {"body_relocid", &Model<T>::body_relocid}, {"jnt_type", &Model<T>::jnt_type}, {"jnt_bodyid", &Model<T>::jnt_bodyid},
            {"jnt_qposadr", &Model<T>::jnt_qposadr}, {"jnt_dofadr", &Model<T>::jnt_dofadr}, {"jnt_limited", &Model<T>::jnt_limited},
            {"dof_bodyid", &Model<T>::dof_bodyid}, {"dof_jntid", &Model<T>::dof_jntid}, {"dof_parentid", &Model<T>::dof_parentid},
            {"geom_type", &Model<T>::geom_type}, {"geom_bodyid", &Model<T>::geom_bodyid}, {"geom_meshid", &Model<T>::geom_meshid},
            {"geom_condim", &Model<T>::geom_condim}, {"mesh_vertadr", &Model<T>::mesh_vertadr}, {"mesh_vertnum", &Model<T>::mesh_vertnum},
            {"pair_geom", &Model<T>::pair_geom}, {"act_dofid", &Model<T>::act_dofid}, {"act_qposid", &Model<T>::act_qposid},
            {"eq_body1", &Model<T>::eq_body1}, {"eq_body2", &Model<T>::eq_body2}, {"probe_body", &Model<T>::probe_body}};
        const RF rfs[] = {{"body_pos", &Model<T>::body_pos}, {"body_quat", &Model<T>::body_quat}, {"body_ipos", &Model<T>::body_ipos},
            {"body_iquat", &Model<T>::body_iquat}, {"body_mass", &Model<T>::body_mass}, {"body_inertia", &Model<T>::body_inertia},
            {"jnt_pos", &Model<T>::jnt_pos}, {"jnt_axis", &Model<T>::jnt_axis}, {"jnt_range", &Model<T>::jnt_range},
            {"jnt_stiffness", &Model<T>::jnt_stiffness}, {"jnt_springref", &Model<T>::jnt_springref}, {"jnt_solref", &Model<T>::jnt_solref},
            {"jnt_solimp", &Model<T>::jnt_solimp}, {"jnt_margin", &Model<T>::jnt_margin}, {"dof_armature", &Model<T>::dof_armature},
            {"dof_damping", &Model<T>::dof_damping}, {"dof_invweight0", &Model<T>::dof_invweight0}, {"qpos0", &Model<T>::qpos0},
            {"geom_size", &Model<T>::geom_size}, {"geom_pos", &Model<T>::geom_pos}, {"geom_quat", &Model<T>::geom_quat},
            {"geom_friction", &Model<T>::geom_friction}, {"geom_solref", &Model<T>::geom_solref}, {"geom_solimp", &Model<T>::geom_solimp},
            {"geom_solmix", &Model<T>::geom_solmix}, {"geom_margin", &Model<T>::geom_margin}, {"geom_gap", &Model<T>::geom_gap},
            {"geom_rbound", &Model<T>::geom_rbound}, {"geom_invweight0", &Model<T>::geom_invweight0}, {"geom_aabb", &Model<T>::geom_aabb}, {"mesh_vert", &Model<T>::mesh_vert},
            {"act_kp", &Model<T>::act_kp}, {"act_ctrlrange", &Model<T>::act_ctrlrange}, {"eq_solref", &Model<T>::eq_solref},
            {"eq_solimp", &Model<T>::eq_solimp}, {"eq_data", &Model<T>::eq_data}, {"eq_invweight0", &Model<T>::eq_invweight0},
            {"probe_pos", &Model<T>::probe_pos}, {"probe_quat", &Model<T>::probe_quat}};
        std::vector<int> ib;
        std::vector<T> rb;
        std::vector<size_t> ioff, roff;
        for (auto& f : ifs) { ioff.push_back(ib.size()); const auto& v = d.I(f.name); ib.insert(ib.end(), v.begin(), v.end()); ib.push_back(0); }
        for (auto& f : rfs) { roff.push_back(rb.size()); const auto& v = d.Rr(f.name); for (double x : v) rb.push_back((T)x); rb.push_back(0); }
        // derived hull tables (see Model::mesh_cellxyz): the coordinates of every support cell's vertex list at a fixed place
        size_t o_cellovf, o_cellxyz, o_ovfxyz, o_dofmask, o_dofqpos;
        {
            const auto &vadr = d.I("mesh_vertadr"), &vnum = d.I("mesh_vertnum"), &cadr = d.I("mesh_celladr"), &cid = d.I("mesh_cellid");
            const auto& vert = d.Rr("mesh_vert");
            const size_t nmesh = vnum.size();
            if (cadr.size() != nmesh * CELL_N + 1) throw std::runtime_error("mesh_celladr: expected nmesh * CELL_N + 1 entries (metaworld_amd/hullcells.py GRID != CELL_GRID?)");
            std::vector<int> covf(nmesh * CELL_N + 1, -1);
            std::vector<T> cxyz(nmesh * CELL_N * 3 * CELL_K + 1, T(0)), oxyz;
            for (size_t mi = 0; mi < nmesh; mi++)
                for (size_t c = 0; c < (size_t)CELL_N; c++) {
                    const int j0 = cadr[mi * CELL_N + c], j1 = cadr[mi * CELL_N + c + 1];
                    if (j1 <= j0) throw std::runtime_error("mesh_celladr: empty support cell");
                    auto put = [&](T* dst, int j) {          // entry j of the list, the last entry beyond its end
                        const int id = cid[j < j1 ? j : j1 - 1];
                        if (id < 0 || id >= vnum[mi]) throw std::runtime_error("mesh_cellid: vertex id out of range");
                        for (int k = 0; k < 3; k++) dst[k] = (T)vert[3 * (size_t)(vadr[mi] + id) + k];
                    };
                    for (int q = 0; q < CELL_K; q++) put(&cxyz[((mi * CELL_N + c) * CELL_K + q) * 3], j0 + q);
                    if (j1 - j0 > CELL_K) {
                        const int nb = (j1 - j0 - 1) / CELL_K;          // further batches
                        if (nb > 15) throw std::runtime_error("support cell list longer than 16 batches");
                        const size_t b0 = oxyz.size() / (3 * CELL_K);
                        covf[mi * CELL_N + c] = (int)(b0 * 16 + nb);
                        oxyz.resize(oxyz.size() + (size_t)nb * 3 * CELL_K);
                        for (int q = 0; q < nb * CELL_K; q++) put(&oxyz[(b0 * CELL_K + q) * 3], j0 + CELL_K + q);
                    }
                }
            oxyz.push_back(T(0));
            o_cellovf = ib.size(); ib.insert(ib.end(), covf.begin(), covf.end());
            // chain mask of every body (see Model::body_dofmask): the Jacobian rows are then filled dof by dof without walking
            // dof_parentid (a chain of dependent loads per row and body)
            const auto &lastdof = d.I("body_lastdof"), &dpar = d.I("dof_parentid");
            if (d.sz.nv > 31) throw std::runtime_error("body_dofmask: more than 31 dofs");
            std::vector<int> dm(lastdof.size() + 1, 0);
            for (size_t b = 0; b < lastdof.size(); b++)
                for (int i = lastdof[b]; i >= 0; i = dpar[i]) dm[b] |= 1 << i;
            o_dofmask = ib.size(); ib.insert(ib.end(), dm.begin(), dm.end());
            // qpos element of every dof (see Model::dof_qposadr): the Euler step then updates qpos dof by dof from registers
            const auto &djnt = d.I("dof_jntid"), &jtype = d.I("jnt_type"), &jqa = d.I("jnt_qposadr"), &jda = d.I("jnt_dofadr");
            std::vector<int> dq(djnt.size() + 1, -1);
            for (size_t i = 0; i < djnt.size(); i++) {
                const int j = djnt[i], k = (int)i - jda[j];
                dq[i] = jtype[j] == J_FREE ? (k < 3 ? jqa[j] + k : -1) : jqa[j];
            }
            o_dofqpos = ib.size(); ib.insert(ib.end(), dq.begin(), dq.end());
            o_cellxyz = rb.size(); rb.insert(rb.end(), cxyz.begin(), cxyz.end());
            o_ovfxyz = rb.size(); rb.insert(rb.end(), oxyz.begin(), oxyz.end());
        }
        iblob = (int*)Backend::alloc(ib.size() * sizeof(int));
        rblob = (T*)Backend::alloc(rb.size() * sizeof(T));
        mw_log_range("model.int", d.sz.nv, iblob, ib.size() * sizeof(int)); mw_log_range("model.real", d.sz.nv, rblob, rb.size() * sizeof(T));
        Backend::h2d(iblob, ib.data(), ib.size() * sizeof(int));
        Backend::h2d(rblob, rb.data(), rb.size() * sizeof(T));
        size_t k = 0;
        for (auto& f : ifs) { const int* q = iblob + ioff[k++]; ::memcpy(&(m.*(f.p)), &q, sizeof(q)); }
        k = 0;
        for (auto& f : rfs) { const T* q = rblob + roff[k++]; ::memcpy(&(m.*(f.p)), &q, sizeof(q)); }
        { const int* q = iblob + o_cellovf; ::memcpy(&m.mesh_cellovf, &q, sizeof(q)); }
        { const int* q = iblob + o_dofmask; ::memcpy(&m.body_dofmask, &q, sizeof(q)); }
        { const int* q = iblob + o_dofqpos; ::memcpy(&m.dof_qposadr, &q, sizeof(q)); }
        { const T* q = rblob + o_cellxyz; ::memcpy(&m.mesh_cellxyz, &q, sizeof(q)); }
        { const T* q = rblob + o_ovfxyz; ::memcpy(&m.mesh_ovfxyz, &q, sizeof(q)); }
        m.sz = d.sz;
        m.L = make_layout(d.sz);
        m.timestep = (T)d.timestep; m.tolerance = (T)d.tolerance; m.meaninertia = (T)d.meaninertia;
        for (int c = 0; c < 3; c++) m.gravity[c] = (T)d.gravity[c];
    }
    ~DeviceModel() { Backend::free(iblob); Backend::free(rblob); }
    DeviceModel(const DeviceModel&) = delete;
};

// ------------------------------------------------------------------ device-visible world description
template <typename T>
struct GroupDev {
    Model<T> m;
    Layout L;
    T* col;
    int* icol;
    int lpb;          // environments per workgroup; the column store is chunked per workgroup: [chunk][element][lpb]
    int nenv, block0;
    int env0;         // environments of the groups before this one (flat environment index of the split-collision kernels)
    const int* gid;   // lane -> global env index
};

struct IOPtrs {
    const float* act;        // [N][4]
    const int* next_goal;    // [N] goal index to use at the next auto-reset of each env
    double* obs;             // [N][D]
    double* reward;          // [N]
    uint8_t *terminated, *truncated, *success, *done;   // [N]
    float* info;             // [N][6]
    double* final_obs;       // [N][D] (valid where done)
    double* ep_ret;          // [N]   (valid where done)
    int* ep_len;             // [N]
    mw_bookkeeping* book;    // [N] packed per-step record for the cross-rank gather, or null
    int* status;             // [MW_STATUS_WORDS] context status: OR of the per-env flags, env-steps with row overflow / contact overflow / instability / sub-lane divergence, solver stalls
    int D;
    // goal schedule of the resident loop (mw_set_goal_schedule; RandomTaskSelectWrapper.reset, wrappers.py:116-119, inside the kernel):
    // the k-th auto-reset of env i since the schedule was set takes sched[min(k, sched_K - 1)][i]; null = next_goal[i]
    const int* sched;        // [sched_K][N]
    int* sched_pos;          // [N] auto-resets of env i since the schedule was set
    int sched_K, N;
};

// (per-env status bits of one step: mw_common.hpp, ST_*; sticky in the context status word until mw_status clears it)

template <typename T>
struct World {
    const GroupDev<T>* groups;
    int ngroups;
    const TaskDesc<T>* tasks;
    const T* snap;            // reset snapshots
    const long long* snap_off;  // per task: element offset of goal 0
    const int* snap_stride;   // per task: elements per snapshot (= nstate + 39)
    const int* snap_ngoal;    // per task: number of goals (snapshots)
    int max_episode_steps, terminate_on_success, one_hot, num_tasks, full_forward;
    int reward_v1;            // 1 = the reference's reward_function_version="v1" branches (mw_tasks_v1.hpp) for every task of the context
    IOPtrs io;
};

template <typename T>
MW_HD bool locate(const World<T>& w, int block, int thread, Scratchpad sp, Env<T>* e, int* gid) {
    int g = 0;
    while (g + 1 < w.ngroups && block >= w.groups[g + 1].block0) g++;
    const GroupDev<T>& G = w.groups[g];
    const int lpb = G.lpb;                    // environments per workgroup of this group (64, or fewer + sub-lanes)
    const bool host = sp.host_nsub > 0;       // host harness: one call per environment
    if (host && thread >= lpb) return false;
    int lin = host ? thread : thread % lpb;
    const int nin = G.nenv - (block - G.block0) * lpb < lpb ? G.nenv - (block - G.block0) * lpb : lpb;   // environments this workgroup really holds
    e->ghost = 0;
    if (lin >= nin) {
        if (host) return false;
        lin = nin - 1; e->ghost = 1;          // device: the surplus threads of a group's last workgroup duplicate its last environment (Env::ghost)
    }
    const int lane = (block - G.block0) * lpb + lin;
    e->slot = lin; e->nslot = nin;
    e->set_scratchpad(sp, thread, lpb, G.m.sz.nv, G.m.sz.nbody, G.m.sz.nq, lin);
    // element i of this environment: chunk base + i * lpb + (lane in chunk) -- consecutive elements of a workgroup's
    // environments are adjacent in memory (a Jacobian row of 8 environments is a few cache lines, not one line per entry)
    const size_t chunk = (size_t)(block - G.block0);
    e->col = G.col + chunk * G.L.nreal * lpb + lin;
    e->icol = G.icol + chunk * G.L.nint * lpb + lin;
    e->m = &G.m; e->stride = (unsigned)lpb;
    e->cache_layout(G.L, G.m.sz.nv);
#if defined(MW_BOUNDS)
    e->nreal_b = (unsigned)G.L.nreal; e->nint_b = (unsigned)G.L.nint; e->oob = w.io.status;
#endif
    *gid = G.gid[lane];
    return true;
}

template <typename T>
MW_HD void write_obs(const World<T>& w, const TaskDesc<T>& td, double* dst, const T* obs39, int onehot_id) {
    for (int k = 0; k < 39; k++) dst[k] = (double)obs39[k];
    if (w.one_hot)
        for (int k = 0; k < w.num_tasks; k++) dst[39 + k] = k == onehot_id ? 1.0 : 0.0;
}

template <typename T>
MW_HD void load_snapshot(const World<T>& w, const Env<T> e, int task, int goal, T* obs39) {
    const int ng = w.snap_ngoal[task];
    goal = goal < 0 ? 0 : (goal >= ng ? ng - 1 : goal);   // device-pointer entry points cannot be range-checked on the host
    const T* s = w.snap + w.snap_off[task] + (long long)goal * w.snap_stride[task];
    const int ns = e.lay().nstate;
    const V3<T> persist = tk3(e, TK_PERSIST0);
    for (int k0 = 0; k0 < ns; k0 += 16) {          // in batches of 16: loads first, then stores (an element-wise copy is one round trip per element)
        T x[16];
#pragma unroll
        for (int q = 0; q < 16; q++) x[q] = s[k0 + q < ns ? k0 + q : ns - 1];
#pragma unroll
        for (int q = 0; q < 16; q++)
            if (k0 + q < ns) e.R(k0 + q) = x[q];
    }
    for (int k = 0; k < 39; k++) obs39[k] = s[ns + k];
    if (w.tasks[task].kind == 1) task_after_reset(e, w.tasks[task], persist, obs39);
}

// ---- lane programs -------------------------------------------------------------------------
// faithful reset (slow path; builds snapshots and serves explicit mw_reset_full)
template <typename T>
MW_HD void lane_reset_full(const World<T>& w, int block, int thread, Scratchpad sp) {
    Env<T> e; int gid;
    if (!locate(w, block, thread, sp, &e, &gid)) return;
    const int task = (int)TK(e, TK_TASK);
    const TaskDesc<T>& td = w.tasks[task];
    T obs[39];
    env_reset(e, td, obs, w.reward_v1 != 0);
    if (w.io.obs && e.sub == 0) write_obs(w, td, w.io.obs + (size_t)gid * w.io.D, obs, (int)td.c[15]);
}

// reset from snapshot for masked envs (mask may be null = all); goal from io.next_goal
template <typename T>
MW_HD void lane_reset_snap(const World<T>& w, const uint8_t* mask, int block, int thread, Scratchpad sp) {
    Env<T> e; int gid;
    if (!locate(w, block, thread, sp, &e, &gid)) return;
    if (mask && !mask[gid]) return;
    const int task = (int)TK(e, TK_TASK);
    const TaskDesc<T>& td = w.tasks[task];
    T obs[39];
    load_snapshot(w, e, task, w.io.next_goal[gid], obs);
    if (w.io.obs && e.sub == 0) write_obs(w, td, w.io.obs + (size_t)gid * w.io.D, obs, (int)td.c[15]);
}

// RecordEpisodeStatistics sums the float64 rewards in float64 (gymnasium); a single-precision context keeps the running
// return as an unevaluated (hi, lo) pair of floats in the task block, so that it is summed in double there as well
template <typename T>
MW_HD double ep_return_add(const Env<T> e, T reward) {
    const double acc = (double)TK(e, TK_EPRET) + (double)TK(e, TK_EPRET_LO) + (double)reward;
    const T hi = (T)acc;
    TK(e, TK_EPRET) = hi; TK(e, TK_EPRET_LO) = (T)(acc - (double)hi);
    return acc;
}
MW_HD bool mw_finite(double x) { return x - x == 0; }

// everything of one VectorEnv.step that follows SawyerXYZEnv.step's (obs, reward, success, info): instability guard, TimeLimit,
// AutoTerminateOnSuccess, OneHot, RecordEpisodeStatistics, the outputs and the SAME_STEP auto-reset (metaworld/__init__.py:430-454, :465)
template <typename T>
MW_HD void step_outputs(const World<T>& w, const Env<T> e, int gid, int task, const TaskDesc<T>& td, T* obs, T reward, T success, Info info) {
    // Instability guard (the intent of the reference's dead `_did_see_sim_exception` branch, sawyer_xyz_env.py:603-619, and
    // of MuJoCo's own reset on a bad QACC): a non-finite step returns the last stable observation with reward 0, ends the
    // episode as truncated (so the SAME_STEP auto-reset below restores a valid state) and raises ST_UNSTABLE.
    int flags = e.I(e.lay().icount + 3);
    // canary (mw_common.hpp, sub_disagree): the sub-lanes of an environment must arrive here with bit-identical results
    {
        unsigned sig = (unsigned)(canary_bits(reward) ^ canary_bits(success));          // (unsigned: the hash wraps around)
        for (int k = 0; k < 18; k++) sig = sig * 31u + (unsigned)canary_bits(obs[k]);
        if (sub_disagree(e, (int)sig) || sub_disagree(e, flags)) flags |= ST_DIVERGED;
    }
    const int nstall = e.I(e.lay().icount + IC_SOLVER_STALL);
    bool bad = !mw_finite((double)reward);
    for (int k = 0; k < 18; k++) bad |= !mw_finite((double)obs[k]);
    if (bad) {
        flags |= ST_UNSTABLE;
        for (int k = 0; k < 18; k++) { obs[k] = obs[18 + k]; TK(e, TK_PREVOBS + k) = obs[18 + k]; }
        reward = 0; success = 0; info = Info{0, 0, 0, 0, 0, 0};
    }
    TK(e, TK_SUCCESS) = success;
    TK(e, TK_ELAPSED) += 1; TK(e, TK_EPLEN) += 1;
    const double ep_ret = ep_return_add(e, reward);
    const bool truncated = bad || TK(e, TK_PATHLEN) >= td.max_path_length || TK(e, TK_ELAPSED) >= w.max_episode_steps;
    const bool terminated = w.terminate_on_success && success == T(1);
    const bool done = terminated || truncated;
    const IOPtrs& io = w.io;
    const bool writer = e.sub == 0 && !e.ghost;          // the sub-lanes of an environment hold identical values: one of them stores (and counts)
    const int ep_len = (int)TK(e, TK_EPLEN);
    if (writer) {
        io.reward[gid] = (double)reward;
        io.terminated[gid] = terminated; io.truncated[gid] = truncated; io.success[gid] = success == T(1);
        io.done[gid] = done;
        if (io.info) {
            float* f = io.info + (size_t)gid * 6;
            f[0] = info.near_object; f[1] = info.grasp_success; f[2] = info.grasp_reward; f[3] = info.in_place_reward;
            f[4] = info.obj_to_target; f[5] = info.unscaled_reward;
        }
        if (io.book) {
            mw_bookkeeping b;
            b.done = done; b.success = success == T(1); b.task_id = (int16_t)td.kind; b.episode_return = (float)ep_ret; b.episode_length = ep_len;
            io.book[gid] = b;
        }
        if (flags && io.status) {
#if defined(__HIP_DEVICE_COMPILE__)
            atomicOr(io.status, flags);
            if (flags & ST_ROW_OVERFLOW) atomicAdd(io.status + 1, 1);
            if (flags & ST_CON_OVERFLOW) atomicAdd(io.status + 2, 1);
            if (flags & ST_UNSTABLE) atomicAdd(io.status + 3, 1);
            if (flags & ST_DIVERGED) atomicAdd(io.status + 4, 1);
#else
#pragma omp critical(mw_status)
            { io.status[0] |= flags; io.status[1] += (flags & ST_ROW_OVERFLOW) != 0; io.status[2] += (flags & ST_CON_OVERFLOW) != 0; io.status[3] += (flags & ST_UNSTABLE) != 0; io.status[4] += (flags & ST_DIVERGED) != 0; }
#endif
        }
        if (nstall && io.status) {          // informational counter (no flag bit): see solve_impl
#if defined(__HIP_DEVICE_COMPILE__)
            atomicAdd(io.status + 5, nstall);
#else
#pragma omp critical(mw_status)
            { io.status[5] += nstall; }
#endif
        }
    }
    const int oh = (int)td.c[15];
    if (done) {
        if (writer) {
            if (io.final_obs) write_obs(w, td, io.final_obs + (size_t)gid * io.D, obs, oh);
            io.ep_ret[gid] = ep_ret; io.ep_len[gid] = ep_len;
        }
        int goal = io.next_goal[gid];
        if (io.sched) {          // (the sub-lanes of an environment read the position before its writer advances it: one wave, in lockstep)
            const int k = io.sched_pos[gid];
            goal = io.sched[(size_t)(k < io.sched_K ? k : io.sched_K - 1) * io.N + gid];
            if (writer) io.sched_pos[gid] = k + 1;
        }
        load_snapshot(w, e, task, goal, obs);
    }
    if (writer) write_obs(w, td, io.obs + (size_t)gid * io.D, obs, oh);
}

// one VectorEnv.step for one env: SawyerXYZEnv.step + the wrapper stack (step_outputs)
template <typename T>
MW_HD void lane_step(const World<T>& w, int block, int thread, Scratchpad sp) {
    Env<T> e; int gid;
    if (!locate(w, block, thread, sp, &e, &gid)) return;
    const int task = (int)TK(e, TK_TASK);
    const TaskDesc<T>& td = w.tasks[task];
    T act[4], obs[39], reward, success;
    Info info;
    for (int k = 0; k < 4; k++) act[k] = (T)w.io.act[(size_t)gid * 4 + k];
    e.I(e.lay().icount + 3) = 0; e.I(e.lay().icount + IC_SOLVER_STALL) = 0;
    MW_TICK(t_ls0)
    env_step(e, td, act, obs, &reward, &success, &info, w.full_forward != 0, w.reward_v1 != 0);
    MW_TICK(t_ls05)
    step_outputs(w, e, gid, task, td, obs, reward, success, info);
#if (defined(MW_SOLVE_FINE) || defined(MW_STEP_FINE)) && defined(MW_SOLVER_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    MW_TICK(t_ls1)
    e.I(e.lay().icount + 4) += (int)((t_ls1 - t_ls0) >> 4);          // slot 0 of the fine timers: the whole step of this wave
#if defined(MW_STEP_FINE)
    e.I(e.lay().icount + 4 + 4) += (int)((t_ls1 - t_ls05) >> 4);     // slot 4: step_outputs (wrappers, auto-reset, output stores)
#endif
#endif
}

// Split collision (narrow phase as batch-wide kernels between the lane kernels): measured 25-37 % SLOWER than the fused kernel
// (profiles/r05_split_collision_*), so it is compiled only into -DMW_SPLIT_COLLISION variants of the library
// (tools/build_variants.sh, tests/test_split_collision.py); the default library carries neither its kernels nor its buffers.
#if defined(MW_SPLIT_COLLISION)
#include "mw_split.inl"
#endif

// debugging / parity hooks: run raw physics on every lane
template <typename T>
MW_HD void lane_debug(const World<T>& w, int what, int n, int block, int thread, Scratchpad sp) {
    Env<T> e; int gid;
    if (!locate(w, block, thread, sp, &e, &gid)) return;
    if (what == 0) forward(e);
    else if (what == 1) for (int k = 0; k < n; k++) substep(e);
    else if (what == 2) reset_data(e);
    else if (what == 3) kinematics(e);
    else if (what >= 10)            // stage timing: run stages 0..(what-10) of one dynamics evaluation, n times
        for (int it = 0; it < n; it++) {
            const int k = what - 10;
            kinematics(e);          // (stages in the order of forward_dynamics: the bias forces before the constraint rows)
            if (k >= 1) crb(e);
            if (k >= 2) smooth_forces(e);
            if (k >= 3) collision(e);
            if (k >= 4) make_constraints(e);
            if (k >= 5) solve(e);
        }
    if (what == 0 || what == 1 || what >= 10) mirror_rows(e);   // the constraint rows of the last evaluation, for mw_read("efcJ" / "efcX")
}

// ------------------------------------------------------------------ host-side context
struct Config {
    int precision;   // 0 = fp32, 1 = fp64
    int device_id, rank, world_size;
    int max_episode_steps, terminate_on_success, one_hot, num_tasks;
    int full_forward;   // 1 = complete final mj_forward for every task (see env_step)
    int reward_version; // 1 = v1 reward functions, anything else = v2 (mw_config.reward_version)
};

struct TaskSpec {      // precision-independent TaskDesc
    int kind, probe[P_COUNT], nobj, quat_mode[2], qadr[4], dadr[4], geom[4], reloc[2], partially_observable, max_path_length;
    double hand_init[3], mocap_low[3], mocap_high[3], goal_low[3], goal_high[3], obj_off[2][3], c[16];
    int model;         // index into Context::models
    std::vector<double> goals;   // [ngoals][6]
};

class ContextBase {
public:
    virtual ~ContextBase() {}
    std::string error;
    Config cfg{};
    virtual void finalize() = 0;
    virtual void build_snapshots() = 0;
    virtual void reset(const uint8_t* mask, const int* goal_idx, double* obs_out) = 0;
    virtual void step(const float* act, const int* next_goal, double* obs, double* reward, uint8_t* term, uint8_t* trunc,
                      uint8_t* success, float* info, double* final_obs, double* ep_ret, int* ep_len) = 0;
    virtual void step_device_only(const float* d_act, int nsteps, int act_stride_steps, float* kernel_ms, bool gather = false) = 0;
    virtual void step_device(const float* d_act, const int* d_next_goal, const mw_device_out* out, bool async = false, void* caller_stream = nullptr) = 0;
    virtual const uint8_t* wait_done() = 0;
    virtual void reset_device(const uint8_t* d_mask, const int* d_goal_idx, double* d_obs) = 0;
    virtual void policy_actions(const int* policy_id, const double* obs, float* act) = 0;
    virtual void policy_rollout(const int* policy_id, const int* schedule, int K, int nsteps, int* episodes, int* successes, float* kernel_ms, int per_launch = 1) = 0;
    virtual void upload_actions(const float* act, int nsteps) = 0;
    virtual void step_resident_gather(int nsteps, int act_stride_steps, float* kernel_ms) = 0;
    virtual void step_fused(int nsteps, int act_stride_steps, int per_launch, float* kernel_ms) = 0;
    virtual void comm_init(const void* id128, int rank, int world) = 0;
    virtual void comm_info(int* out /*[4]*/) = 0;
    virtual void gather_bookkeeping(mw_bookkeeping* out, int out_on_device) = 0;
    virtual void status(int* out, int n, int clear) = 0;
    virtual int launch_times(float* out, int cap) = 0;
    virtual void set_option(const std::string& name, double value) = 0;
    virtual void set_episode_phase(const int* elapsed) = 0;
    virtual void set_goal_schedule(const int* schedule, int K) = 0;
    virtual void goal_schedule_pos(int* out) = 0;
    virtual void state_all(double* buf, int stride, bool write) = 0;
    virtual void read_col(int gid, const char* what, int n, double* out) = 0;
    virtual void write_col(int gid, const char* what, int n, const double* in) = 0;
    virtual void read_icol(int gid, const char* what, int n, int* out) = 0;
    virtual void debug(int what, int n) = 0;
    virtual int layout_size(int gid, const char* what) = 0;
    std::vector<std::shared_ptr<ModelData>> models;
    std::vector<TaskSpec> tasks;
    std::vector<int> env_task;   // per global env: task index
    int obs_dim() const { return 39 + (cfg.one_hot ? cfg.num_tasks : 0); }
};

template <typename T, typename Backend>
class Context : public ContextBase {
    template <typename, typename> friend class Context;
    struct Group {
        std::unique_ptr<DeviceModel<T, Backend>> dm;
        Layout L;
        T* col = nullptr;
        int* icol = nullptr;
        int* gid_dev = nullptr;
        int lpb = BLOCK;
        size_t nchunk = 0;
        size_t at(int i, int lane) const { return ((size_t)(lane / lpb) * L.nreal + i) * lpb + lane % lpb; }     // real column element
        size_t iat(int i, int lane) const { return ((size_t)(lane / lpb) * L.nint + i) * lpb + lane % lpb; }     // int column element
        size_t nreal_total() const { return nchunk * lpb * (size_t)L.nreal; }
        size_t nint_total() const { return nchunk * lpb * (size_t)L.nint; }
        int nenv = 0, block0 = 0, model = 0, env0 = 0;
        std::vector<int> gid;
    };
    std::vector<Group> groups_;
    std::vector<int> env_group_, env_lane_;
    GroupDev<T>* d_groups_ = nullptr;
    TaskDesc<T>* d_tasks_ = nullptr;
    T* d_snap_ = nullptr;
    long long* d_snap_off_ = nullptr;
    int* d_snap_stride_ = nullptr;
    int nblocks_ = 0, N_ = 0;
    // io buffers (device) + host staging
    float* d_act_ = nullptr; size_t act_capacity_steps_ = 0;
    int* d_next_goal_ = nullptr;
    int *d_sched_ = nullptr, *d_sched_pos_ = nullptr;   // goal schedule of the resident loop (set_goal_schedule)
    int sched_K_ = 0;
    uint8_t* d_mask_ = nullptr;
    double *d_obs_ = nullptr, *d_reward_ = nullptr, *d_final_ = nullptr, *d_epret_ = nullptr;
    uint8_t* d_flags_ = nullptr;   // terminated, truncated, success, done : 4 x N
    float* d_info_ = nullptr;
    int* d_eplen_ = nullptr;
    std::vector<long long> snap_off_;
    std::vector<int> snap_stride_;
    int* d_snap_ngoal_ = nullptr;
    int* d_status_ = nullptr;                 // [MW_STATUS_WORDS], see mw_status
    mw_bookkeeping* d_book_ = nullptr;        // [2][N]: the step kernel writes slot (step parity), the gather reads it
    mw_bookkeeping* d_book_all_ = nullptr;    // [2][world][N] gathered records
    int book_slot_ = 0;                       // slot the LAST step wrote
    typename Backend::Comm* comm_ = nullptr;
    std::vector<int> h_next_goal_;            // host mirror of d_next_goal_ (masked resets update only their envs)
    std::vector<uint8_t> was_reset_;          // step() before the first reset() of an env is an error
    int world_size() const { return comm_ ? comm_->world : 1; }
    uint8_t* h_done_ = nullptr;               // pinned host copy of the `done` row of the last mw_step_device_on
    int split_collision_ = 0;                 // mw_set_option("split_collision"): 1 = the narrow phase runs as its own batch-wide kernels (launch_step); -DMW_SPLIT_COLLISION builds only
#if defined(MW_SPLIT_COLLISION)
    SplitBuf<T> sb_{};                        // work items / hit table of the split collision (allocated on first use)
    bool any_lazy_dynamics_ = false;          // some task's reward reads contact forces (task_touches) or cfg.full_forward: a sixth narrow phase per step
#else
    void free_split_buffers() {}
#endif
#if defined(MW_SPLIT_COLLISION)
    void ensure_split_buffers() {
        if (sb_.counts) return;
        if (groups_.size() > 127) throw std::runtime_error("split_collision: more than 127 model groups");
        for (auto& g : groups_) if (g.nenv >= (1 << 24)) throw std::runtime_error("split_collision: more than 2^24 environments in one group");
        const char* ov = getenv("MW_SPLIT_ITEMS_PER_ENV");
        const size_t cap = (size_t)N_ * (ov ? atoi(ov) : 48);
        sb_.cap = (int)cap;
        sb_.counts = (int*)Backend::alloc(sizeof(int) * SC_WORDS); Backend::zero(sb_.counts, sizeof(int) * SC_WORDS);
        sb_.item_env = (int*)Backend::alloc(sizeof(int) * cap); sb_.item_pair = (int*)Backend::alloc(sizeof(int) * cap);
        sb_.class_list = (int*)Backend::alloc(sizeof(int) * cap * NP_NCLASS);
        sb_.hit_n = (int*)Backend::alloc(sizeof(int) * cap);
        sb_.hits = (T*)Backend::alloc(sizeof(T) * cap * NP_MAXHIT * NP_HIT_W);
        any_lazy_dynamics_ = cfg.full_forward != 0;
        for (int t : env_task) any_lazy_dynamics_ |= task_touches(tasks[t].kind);
    }
    void free_split_buffers() {
        Backend::free(sb_.counts); Backend::free(sb_.item_env); Backend::free(sb_.item_pair); Backend::free(sb_.class_list); Backend::free(sb_.hit_n); Backend::free(sb_.hits);
        sb_ = SplitBuf<T>{};
    }
    // one mid phase + narrow phase over the whole batch (or, only_pending, over the environments that wait for the lazy final dynamics)
    void launch_collision(const World<T>& w, bool only_pending) {
        const SplitBuf<T> sb = sb_;
        Backend::zero(sb.counts, sizeof(int) * SC_WORDS);
        Backend::launch_waves(N_, 0, [w, sb, only_pending] MW_LAMBDA(int wave, int tid, int nwaves, WaveLds lds) {
            (void)nwaves; (void)lds;
            mid_phase_env(w, sb, wave, tid, only_pending);
        });
        Backend::launch_waves(Backend::narrow_waves(), (int)(TLS_SLOTS * 64 * sizeof(T)), [w, sb] MW_LAMBDA(int wave, int tid, int nwaves, WaveLds lds) {
            narrow_wave(w, sb, wave, tid, nwaves, (MW_LDS T*)lds.base);
        });
    }
    // (ONE kernel for the four phases, selected at run time: one copy of the lane programs in the code object)
    void phase_kernel(const World<T>& w, const SplitBuf<T>& sb, int phase) {
        Backend::launch(nblocks_, [w, sb, phase] MW_LAMBDA(int b, int t, Scratchpad sp) { lane_phase(w, sb, phase, b, t, sp); });
    }
#endif
    // ONE VectorEnv.step of the whole batch on the context's stream: the fused kernel, or the split-collision sequence (mw_split.inl)
    void launch_step(const World<T>& w) {
        if (!split_collision_) {
            Backend::launch(nblocks_, [w] MW_LAMBDA(int b, int t, Scratchpad sp) { lane_step(w, b, t, sp); });
            return;
        }
#if defined(MW_SPLIT_COLLISION)
        ensure_split_buffers();
        const SplitBuf<T> sb = sb_;
        phase_kernel(w, sb, PH_BEGIN);
        for (int k = 0; k < 5; k++) {
            launch_collision(w, false);
            phase_kernel(w, sb, k < 4 ? PH_MID : PH_LAST);
        }
        if (any_lazy_dynamics_) {
            launch_collision(w, true);
            phase_kernel(w, sb, PH_FINAL);
        }
#endif
    }

    World<T> world(bool with_io = true) const {
        World<T> w{};
        w.groups = d_groups_; w.ngroups = (int)groups_.size(); w.tasks = d_tasks_;
        w.snap = d_snap_; w.snap_off = d_snap_off_; w.snap_stride = d_snap_stride_; w.snap_ngoal = d_snap_ngoal_;
        w.max_episode_steps = cfg.max_episode_steps; w.terminate_on_success = cfg.terminate_on_success;
        w.one_hot = cfg.one_hot; w.num_tasks = cfg.num_tasks; w.full_forward = cfg.full_forward; w.reward_v1 = cfg.reward_version == 1;
        if (with_io) {
            w.io.act = d_act_; w.io.next_goal = d_next_goal_; w.io.obs = d_obs_; w.io.reward = d_reward_;
            w.io.terminated = d_flags_; w.io.truncated = d_flags_ + N_; w.io.success = d_flags_ + 2 * N_; w.io.done = d_flags_ + 3 * N_;
            w.io.info = d_info_; w.io.final_obs = d_final_; w.io.ep_ret = d_epret_; w.io.ep_len = d_eplen_; w.io.D = obs_dim();
            w.io.status = d_status_; w.io.book = d_book_ ? d_book_ + (size_t)book_slot_ * N_ : nullptr;
            w.io.N = N_;          // (io.sched stays null: only the resident loop hands the goal schedule to the kernel)
        }
        return w;
    }
    static TaskDesc<T> to_desc(const TaskSpec& s) {
        TaskDesc<T> d{};
        d.kind = s.kind; d.nobj = s.nobj; d.partially_observable = s.partially_observable; d.max_path_length = s.max_path_length;
        for (int k = 0; k < P_COUNT; k++) d.probe[k] = s.probe[k];
        for (int k = 0; k < 2; k++) { d.quat_mode[k] = s.quat_mode[k]; d.reloc[k] = s.reloc[k]; for (int c = 0; c < 3; c++) d.obj_off[k][c] = (T)s.obj_off[k][c]; }
        for (int k = 0; k < 4; k++) { d.qadr[k] = s.qadr[k]; d.dadr[k] = s.dadr[k]; d.geom[k] = s.geom[k]; }
        for (int k = 0; k < 3; k++) {
            d.hand_init[k] = (T)s.hand_init[k]; d.mocap_low[k] = (T)s.mocap_low[k]; d.mocap_high[k] = (T)s.mocap_high[k];
            d.goal_low[k] = (T)s.goal_low[k]; d.goal_high[k] = (T)s.goal_high[k];
        }
        for (int k = 0; k < 16; k++) d.c[k] = (T)s.c[k];
        return d;
    }
    void make_group(Group& g, int model, const std::vector<int>& gids, int lpb) {
        g.model = model;
        g.dm.reset(new DeviceModel<T, Backend>(*models[model]));
        g.L = make_layout(models[model]->sz);
        g.nenv = (int)gids.size();
        g.gid = gids;
        g.lpb = lpb;
        g.nchunk = (size_t)((g.nenv + lpb - 1) / lpb);
        g.col = (T*)Backend::alloc(sizeof(T) * g.nreal_total());
        g.icol = (int*)Backend::alloc(sizeof(int) * g.nint_total());
        Backend::zero(g.col, sizeof(T) * g.nreal_total());
        Backend::zero(g.icol, sizeof(int) * g.nint_total());
        g.gid_dev = (int*)Backend::alloc(sizeof(int) * g.nenv);
        mw_log_range("col", model, g.col, sizeof(T) * g.nreal_total()); mw_log_range("icol", model, g.icol, sizeof(int) * g.nint_total());
        mw_log_range("gid", model, g.gid_dev, sizeof(int) * g.nenv);
        Backend::h2d(g.gid_dev, gids.data(), sizeof(int) * g.nenv);
    }
    void free_group(Group& g) { Backend::free(g.col); Backend::free(g.icol); Backend::free(g.gid_dev); g.col = nullptr; }
    GroupDev<T> dev_of(const Group& g) const {
        GroupDev<T> d{};
        d.m = g.dm->m; d.L = g.L; d.col = g.col; d.icol = g.icol; d.lpb = g.lpb; d.nenv = g.nenv; d.block0 = g.block0; d.env0 = g.env0; d.gid = g.gid_dev;
        return d;
    }
    void set_task_field(Group& g, int lane, int k, double v) {
        T x = (T)v;
        Backend::h2d(g.col + g.at(g.L.task + k, lane), &x, sizeof(T));
    }

public:
    ~Context() override {
        for (auto& g : groups_) free_group(g);
        Backend::free(d_groups_); Backend::free(d_tasks_); Backend::free(d_snap_); Backend::free(d_snap_off_); Backend::free(d_snap_stride_);
        Backend::free(d_sched_); Backend::free(d_sched_pos_);
        Backend::free(d_act_); Backend::free(d_next_goal_); Backend::free(d_mask_); Backend::free(d_obs_); Backend::free(d_reward_);
        Backend::free(d_final_); Backend::free(d_epret_); Backend::free(d_flags_); Backend::free(d_info_); Backend::free(d_eplen_);
        Backend::free(d_snap_ngoal_); Backend::free(d_status_); Backend::free(d_book_); Backend::free(d_book_all_);
        free_split_buffers();
        Backend::free_host(h_done_);
        Backend::comm_free(comm_);
    }

    void finalize() override {
        N_ = (int)env_task.size();
        if (N_ == 0) throw std::runtime_error("no environments");
        // one group per model, lanes in global-env order
        std::map<int, std::vector<int>> by_model;
        for (int i = 0; i < N_; i++) by_model[tasks.at(env_task[i]).model].push_back(i);
        env_group_.assign(N_, 0); env_lane_.assign(N_, 0);
        groups_.resize(by_model.size());
        // Lanes per workgroup, per group.  The lane programs are latency-bound and a wave runs as long as its slowest lane,
        // so while the batch does not fill the chip (one wave per SIMD: the lane programs use all 512 VGPRs) the groups
        // are spread over MORE, emptier waves whose idle threads become sub-lanes.  The launch lasts as long as its slowest
        // wave, so the waves go where they shorten the slowest group: each model carries two measured numbers (model options
        // step_ms_lpb4 / step_ms_lpb8, shipped in metaworld_amd/data/model_caps.json): its critical-path weight at 4 lanes per
        // workgroup inside the MT50 @ 4096 workload (tools/mix_timing.py) and the same scaled by its 8-lane / 4-lane step-time
        // ratio (tools/per_task_timing.py).  Greedy: halve the lanes of the group with the largest predicted time while the grid
        // still fits the one-wave-per-SIMD budget; a group that no longer fits is frozen and the next one is tried.  A model
        // without measurements is ranked by its row capacity.  MW_LANES_PER_BLOCK forces one value for every group.
        std::map<int, int> lpb_of;
        {
            const char* ov = getenv("MW_LANES_PER_BLOCK");
            // MILD OVERSUBSCRIPTION (round 5): the budget is 1.2 x the SIMDs (MW_OVERSUBSCRIBE = f overrides; 1 = rounds 1-4: one workgroup
            // per SIMD at most).  The launch lasts as long as its slowest wave while the mean wave is busy ~40 % of it; with a few more,
            // smaller workgroups than wave slots the dispatcher hands the next workgroup to whichever SIMD frees up first, and with the
            // groups ORDERED by predicted time (longest first: list scheduling, below) the short ones fill the tail.  Measured at MT50 @
            // 4096 fp64 inside one call (profiles/r05_oversubscribe_ab.txt): f = 1 1.134 M env-steps/s (lanes per workgroup 1 / 2 / 4 / 8
            // for 1 / 5 / 15 / 17 scenes, 1021 workgroups), 1.1 1.156 M, **1.2 1.200 M** (1 / 2 / 4 for 1 / 5 / 32 scenes, 1221
            // workgroups: the light scenes go from 8 to 4 environments per workgroup), 1.25 1.196 M, 1.3 1.182 M, 1.4 1.138 M, 1.5 1.112 M,
            // 2 1.02 M, 3 0.87 M -- beyond ~1.3 the extra chains cost more than the shorter critical path gains.  (Round 3 measured "4
            // lanes everywhere" at -8 %: 1040 workgroups in model order, whose last 16 started when the first wave had finished.)
            const double oversub = getenv("MW_OVERSUBSCRIBE") ? atof(getenv("MW_OVERSUBSCRIBE")) : 1.2;
            const int budget = (int)(4 * Backend::compute_units() * (oversub > 1.0 ? oversub : 1.0));
            auto blocks = [&]() { int nb = 0; for (auto& kv : by_model) nb += ((int)kv.second.size() + lpb_of[kv.first] - 1) / lpb_of[kv.first]; return nb; };
            auto predicted = [&](int model, int l) {
                const ModelData& md = *models[model];
                double t4 = md.step_ms_lpb4, t8 = md.step_ms_lpb8;
                if (!(t4 > 0 && t8 >= t4)) { t4 = 1.0 + md.sz.maxefc / 150.0; t8 = 1.3 * t4; }
                // measured shape of t(lanes) relative to the step t8 - t4 (box-close 2.7 / 3.1 / 3.6 / 5.5 / 8.9 ms at 1 / 2 / 4 / 8 / 16
                // lanes, reach 1.45 / 1.49 / 1.67 / 2.0 at 2 / 4 / 8 / 16): linear above 4, flattening below (more butterfly stages)
                const double f = l >= 4 ? (l - 4) / 4.0 * (l > 8 ? 0.93 : 1.0) : (l == 2 ? -0.4 : -0.55);
                return t4 + (t8 - t4) * f;
            };
            // upper end of the search: the smallest uniform value that fits the budget (MT50 @ 4096: 8).  Going further (light scenes
            // at 16+ lanes to give more heavy ones 2) looks better on the per-scene measurements and is worse on the whole
            // batch: MT50 @ 4096 fp64 ran at 873 k env-steps/s with {2, 4, 8}, 864 k with 4 everywhere, 722 k once 16 was
            // allowed (late in an episode EVERY scene has environments with expensive mesh contacts, tools/mix_timing.py)
            int start = 1;
            for (;; start *= 2) {
                for (auto& kv : by_model) lpb_of[kv.first] = start;
                if (start >= BLOCK || blocks() <= budget) break;
            }
            if (const char* mx = getenv("MW_LPB_MAX")) {          // experiments: upper end of the search
                const int v = atoi(mx);
                if (v != 1 && v != 2 && v != 4 && v != 8 && v != 16 && v != 32 && v != 64) throw std::runtime_error("MW_LPB_MAX must be a power of two in [1, 64]");
                start = v;
            }
            for (auto& kv : by_model) lpb_of[kv.first] = start;
            std::set<int> frozen;
            for (;;) {
                int best = -1; double bw = -1;
                for (auto& kv : by_model) {
                    const int l = lpb_of[kv.first];
                    if (l <= 1 || frozen.count(kv.first)) continue;
                    const double wgt = predicted(kv.first, l);
                    if (wgt > bw) { bw = wgt; best = kv.first; }
                }
                if (best < 0) break;
                lpb_of[best] /= 2;
                if (blocks() > budget) { lpb_of[best] *= 2; frozen.insert(best); }
            }
            // Optional second pass (minimax exchange, MW_LPB_EXCHANGE=1): the greedy above stops when the grid is full; the slowest
            // predicted group can still be halved if lighter groups give up waves by DOUBLING their lanes, as long as none of them
            // comes within 3 % of the time that is being removed.  OFF by default: measured on MT50 @ 4096 fp64 (round 3) it moved 8
            // light scenes to 16 lanes and one heavy scene to 2 and was 2 % SLOWER (817 k vs 835 k env-steps/s inside one GPU call);
            // a table re-measured with 4 / 8 lanes everywhere was 3 % slower than the shipped one (1.04 M vs 1.07 M), 4 lanes
            // everywhere 8 % slower.  The slowest waves are set by ONE environment's serial narrow phase or solver, which more
            // sub-lanes do not shorten, while every doubled scene pays for the longer union of its lanes' iterations.
            if (getenv("MW_LPB_EXCHANGE") && atoi(getenv("MW_LPB_EXCHANGE")) == 1) {
                for (int round = 0; round < 64; round++) {
                    int gmax = -1; double tmax = -1;
                    for (auto& kv : by_model) {
                        const double t = predicted(kv.first, lpb_of[kv.first]);
                        if (t > tmax) { tmax = t; gmax = kv.first; }
                    }
                    if (gmax < 0 || lpb_of[gmax] <= 1) break;
                    const double tnew = predicted(gmax, lpb_of[gmax] / 2), bound = 0.97 * tmax;
                    std::map<int, int> trial = lpb_of;
                    trial[gmax] /= 2;
                    auto blocks_of = [&](const std::map<int, int>& l) { int nb = 0; for (auto& kv : by_model) nb += ((int)kv.second.size() + l.at(kv.first) - 1) / l.at(kv.first); return nb; };
                    std::set<int> used;
                    while (blocks_of(trial) > budget) {
                        int donor = -1; double best = 1e300;
                        for (auto& kv : by_model) {
                            if (kv.first == gmax || used.count(kv.first) || trial[kv.first] >= BLOCK) continue;
                            const double t = predicted(kv.first, trial[kv.first] * 2);
                            if (t < bound && t < best) { best = t; donor = kv.first; }
                        }
                        if (donor < 0) break;
                        trial[donor] *= 2; used.insert(donor);
                    }
                    if (blocks_of(trial) > budget || !(tnew < tmax)) break;
                    lpb_of = trial;
                }
            }
            for (auto& kv : by_model) {          // an explicit per-model choice (model option "lanes_per_block") wins over the proxy
                const int l = models[kv.first]->lanes_per_block;
                if (l == 0) continue;
                if (l != 1 && l != 2 && l != 4 && l != 8 && l != 16 && l != 32 && l != 64) throw std::runtime_error("lanes_per_block must be a power of two <= 64");
                lpb_of[kv.first] = l;
            }
            if (ov) {
                const int l = atoi(ov);
                if (l != 1 && l != 2 && l != 4 && l != 8 && l != 16 && l != 32 && l != 64) throw std::runtime_error("lanes per block must be a power of two <= 64");
                for (auto& kv : by_model) lpb_of[kv.first] = l;
            }
        }
        if (getenv("MW_VERBOSE")) {
            std::map<int, int> hist;
            int nb = 0;
            for (auto& kv : by_model) { hist[lpb_of[kv.first]]++; nb += ((int)kv.second.size() + lpb_of[kv.first] - 1) / lpb_of[kv.first]; }
            fprintf(stderr, "[mwgpu] lanes per workgroup -> number of scenes:");
            for (auto& kv : hist) fprintf(stderr, " %d:%d", kv.first, kv.second);
            fprintf(stderr, "  (%d workgroups)\n", nb);
        }
        // block order = group order: by predicted workgroup time, longest first (by model index with MW_OVERSUBSCRIBE <= 1)
        std::vector<std::pair<int, std::vector<int>>> ordered(by_model.begin(), by_model.end());
        if (!getenv("MW_OVERSUBSCRIBE") || atof(getenv("MW_OVERSUBSCRIBE")) > 1.0) {
            auto weight = [&](int model) {
                const ModelData& md = *models[model];
                double t4 = md.step_ms_lpb4, t8 = md.step_ms_lpb8;
                if (!(t4 > 0 && t8 >= t4)) { t4 = 1.0 + md.sz.maxefc / 150.0; t8 = 1.3 * t4; }
                const int l = lpb_of[model];
                const double f = l >= 4 ? (l - 4) / 4.0 * (l > 8 ? 0.93 : 1.0) : (l == 2 ? -0.4 : -0.55);
                return t4 + (t8 - t4) * f;
            };
            std::stable_sort(ordered.begin(), ordered.end(), [&](const auto& a, const auto& b) { return weight(a.first) > weight(b.first); });
        }
        int gi = 0, blk = 0, env0 = 0;
        for (auto& kv : ordered) {
            Group& g = groups_[gi];
            make_group(g, kv.first, kv.second, lpb_of[kv.first]);
            g.block0 = blk;
            g.env0 = env0; env0 += g.nenv;
            blk += (g.nenv + g.lpb - 1) / g.lpb;
            for (int l = 0; l < g.nenv; l++) { env_group_[kv.second[l]] = gi; env_lane_[kv.second[l]] = l; }
            gi++;
        }
        nblocks_ = blk;
        std::vector<GroupDev<T>> gd;
        for (auto& g : groups_) gd.push_back(dev_of(g));
        d_groups_ = (GroupDev<T>*)Backend::alloc(sizeof(GroupDev<T>) * gd.size());
        Backend::h2d(d_groups_, gd.data(), sizeof(GroupDev<T>) * gd.size());
        upload_tasks();
        const int D = obs_dim();
        d_next_goal_ = (int*)Backend::alloc(sizeof(int) * N_); Backend::zero(d_next_goal_, sizeof(int) * N_);
        d_mask_ = (uint8_t*)Backend::alloc(N_);
        d_obs_ = (double*)Backend::alloc(sizeof(double) * N_ * D); d_final_ = (double*)Backend::alloc(sizeof(double) * N_ * D);
        Backend::zero(d_final_, sizeof(double) * N_ * D);
        d_reward_ = (double*)Backend::alloc(sizeof(double) * N_); d_epret_ = (double*)Backend::alloc(sizeof(double) * N_);
        Backend::zero(d_epret_, sizeof(double) * N_);
        d_flags_ = (uint8_t*)Backend::alloc(4 * N_); d_info_ = (float*)Backend::alloc(sizeof(float) * 6 * N_);
        d_eplen_ = (int*)Backend::alloc(sizeof(int) * N_); Backend::zero(d_eplen_, sizeof(int) * N_);
        d_act_ = (float*)Backend::alloc(sizeof(float) * 4 * N_); act_capacity_steps_ = 1;
        d_status_ = (int*)Backend::alloc(sizeof(int) * MW_STATUS_WORDS); Backend::zero(d_status_, sizeof(int) * MW_STATUS_WORDS);
        d_book_ = (mw_bookkeeping*)Backend::alloc(sizeof(mw_bookkeeping) * 2 * N_); Backend::zero(d_book_, sizeof(mw_bookkeeping) * 2 * N_);
        d_book_all_ = (mw_bookkeeping*)Backend::alloc(sizeof(mw_bookkeeping) * 2 * N_); Backend::zero(d_book_all_, sizeof(mw_bookkeeping) * 2 * N_);
        h_next_goal_.assign(N_, 0); was_reset_.assign(N_, 0);
        mw_log_range("obs", 0, d_obs_, sizeof(double) * N_ * D); mw_log_range("final_obs", 0, d_final_, sizeof(double) * N_ * D);
        mw_log_range("act", 0, d_act_, sizeof(float) * 4 * N_); mw_log_range("groups", 0, d_groups_, sizeof(GroupDev<T>) * gd.size());
        mw_log_range("tasks", 0, d_tasks_, sizeof(TaskDesc<T>) * tasks.size());
        for (int i = 0; i < N_; i++) set_task_field(groups_[env_group_[i]], env_lane_[i], TK_TASK, env_task[i]);
        build_snapshots();
    }

    void upload_tasks() {
        std::vector<TaskDesc<T>> td;
        for (auto& s : tasks) td.push_back(to_desc(s));
        Backend::free(d_tasks_);
        d_tasks_ = (TaskDesc<T>*)Backend::alloc(sizeof(TaskDesc<T>) * td.size());
        Backend::h2d(d_tasks_, td.data(), sizeof(TaskDesc<T>) * td.size());
    }

    // run the faithful reset once per (task, goal) and keep the resulting persistent state + reset observation
    // The snapshots are always computed in double precision with the model's own solver tolerance, also for an fp32
    // context: every episode then starts from the reference's reset state (to ~1e-7 after the cast) instead of from a
    // single-precision replay of the 500 settling substeps.
    void build_snapshots() override {
        Backend::free(d_snap_); Backend::free(d_snap_off_); Backend::free(d_snap_stride_); Backend::free(d_snap_ngoal_);
        std::vector<T> snap;
        if (sizeof(T) == 8) compute_snapshots(snap);
        else {
            Context<double, Backend> c64;
            c64.cfg = cfg;
            c64.tasks = tasks;
            for (auto& m : models) {
                auto md = std::make_shared<ModelData>(*m);
                if (md->reset_tolerance > 0) md->tolerance = md->reset_tolerance;
                c64.models.push_back(md);
            }
            c64.upload_tasks();
            std::vector<double> s64;
            c64.compute_snapshots(s64);
            snap.assign(s64.begin(), s64.end());
            snap_off_ = c64.snap_off_; snap_stride_ = c64.snap_stride_;
        }
        const long long total = (long long)snap.size();
        d_snap_ = (T*)Backend::alloc(sizeof(T) * (size_t)(total > 0 ? total : 1));
        Backend::h2d(d_snap_, snap.data(), sizeof(T) * (size_t)total);
        mw_log_range("snapshots", 0, d_snap_, sizeof(T) * (size_t)total);
        d_snap_off_ = (long long*)Backend::alloc(sizeof(long long) * tasks.size());
        Backend::h2d(d_snap_off_, snap_off_.data(), sizeof(long long) * tasks.size());
        d_snap_stride_ = (int*)Backend::alloc(sizeof(int) * tasks.size());
        Backend::h2d(d_snap_stride_, snap_stride_.data(), sizeof(int) * tasks.size());
        std::vector<int> ngoal;
        for (auto& t : tasks) ngoal.push_back((int)(t.goals.size() / 6));
        d_snap_ngoal_ = (int*)Backend::alloc(sizeof(int) * tasks.size());
        Backend::h2d(d_snap_ngoal_, ngoal.data(), sizeof(int) * tasks.size());
    }
    void compute_snapshots(std::vector<T>& snap) {
        snap_off_.assign(tasks.size(), 0); snap_stride_.assign(tasks.size(), 0);
        long long total = 0;
        for (size_t t = 0; t < tasks.size(); t++) {
            const int ns = make_layout(models[tasks[t].model]->sz).nstate + 39;
            snap_off_[t] = total; snap_stride_[t] = ns;
            total += (long long)ns * (tasks[t].goals.size() / 6);
        }
        snap.assign((size_t)total, (T)0);
        // temporary groups: one per TASK, one lane per goal.  (One group per model would put two tasks of a shared scene into one
        // wave; reset_model is task-specific code around the non-inlined physics stages, which would then be entered under a
        // partial EXEC mask -- every wave of the faithful reset is task-uniform instead.)
        std::map<int, std::vector<std::pair<int, int>>> by_task;
        for (size_t t = 0; t < tasks.size(); t++)
            for (size_t g = 0; g < tasks[t].goals.size() / 6; g++) by_task[(int)t].push_back({(int)t, (int)g});
        for (auto& kv : by_task) {
            Group g;
            std::vector<int> gids(kv.second.size());
            for (size_t i = 0; i < gids.size(); i++) gids[i] = (int)i;
            make_group(g, tasks[kv.first].model, gids, BLOCK);
            // seed the task block of each lane: task id, goal idx, rand_vec
            std::vector<T> host(g.nreal_total(), (T)0);
            for (size_t l = 0; l < kv.second.size(); l++) {
                const int t = kv.second[l].first, go = kv.second[l].second;
                host[g.at(g.L.task + TK_TASK, (int)l)] = (T)t;
                host[g.at(g.L.task + TK_GOAL, (int)l)] = (T)go;
                for (int k = 0; k < 6; k++) host[g.at(g.L.task + TK_RANDVEC + k, (int)l)] = (T)tasks[t].goals[6 * go + k];
            }
            Backend::h2d(g.col, host.data(), host.size() * sizeof(T));
            GroupDev<T> gd = dev_of(g);
            GroupDev<T>* d_g = (GroupDev<T>*)Backend::alloc(sizeof(GroupDev<T>));
            Backend::h2d(d_g, &gd, sizeof(gd));
            const int D = obs_dim();
            double* d_o = (double*)Backend::alloc(sizeof(double) * g.nenv * D);
            World<T> w = world(false);
            w.groups = d_g; w.ngroups = 1; w.io.obs = d_o; w.io.D = D;
            Backend::launch((g.nenv + BLOCK - 1) / BLOCK, [w] MW_LAMBDA(int b, int t, Scratchpad sp) { lane_reset_full(w, b, t, sp); });
            Backend::sync();
            Backend::d2h(host.data(), g.col, host.size() * sizeof(T));
            std::vector<double> obs((size_t)g.nenv * D);
            Backend::d2h(obs.data(), d_o, obs.size() * sizeof(double));
            for (size_t l = 0; l < kv.second.size(); l++) {
                const int t = kv.second[l].first, go = kv.second[l].second;
                T* dst = snap.data() + snap_off_[t] + (long long)go * snap_stride_[t];
                for (int k = 0; k < g.L.nstate; k++) dst[k] = host[g.at(k, (int)l)];
                for (int k = 0; k < 39; k++) dst[g.L.nstate + k] = (T)obs[l * D + k];
            }
            Backend::free(d_o); Backend::free(d_g);
            free_group(g);
        }
    }

    int ngoals_of(int env) const { return (int)(tasks[env_task[env]].goals.size() / 6); }
    void check_goals(const int* goal_idx, const uint8_t* mask, const char* who) const {
        for (int i = 0; i < N_; i++) {
            if (mask && !mask[i]) continue;
            if (goal_idx[i] < 0 || goal_idx[i] >= ngoals_of(i))
                throw std::out_of_range(std::string(who) + ": goal index " + std::to_string(goal_idx[i]) + " of env " + std::to_string(i) +
                                        " is outside its task's goal table [0, " + std::to_string(ngoals_of(i)) + ")");
        }
    }
    void reset(const uint8_t* mask, const int* goal_idx, double* obs_out) override {
        if (!goal_idx) throw std::invalid_argument("reset: goal_idx is required");
        check_goals(goal_idx, mask, "reset");
        for (int i = 0; i < N_; i++)
            if (!mask || mask[i]) { h_next_goal_[i] = goal_idx[i]; was_reset_[i] = 1; }     // a masked reset leaves the other envs' look-ahead goals alone
        Backend::h2d(d_next_goal_, h_next_goal_.data(), sizeof(int) * N_);
        const uint8_t* dm = nullptr;
        if (mask) { Backend::h2d(d_mask_, mask, N_); dm = d_mask_; }
        World<T> w = world();
        Backend::launch(nblocks_, [w, dm] MW_LAMBDA(int b, int t, Scratchpad sp) { lane_reset_snap(w, dm, b, t, sp); });
        Backend::sync();
        if (obs_out) Backend::d2h(obs_out, d_obs_, sizeof(double) * N_ * obs_dim());
    }
    void need_reset_done(const char* who) const {
        for (int i = 0; i < N_; i++)
            if (!was_reset_[i]) throw std::logic_error(std::string(who) + ": env " + std::to_string(i) + " has never been reset (call reset first)");
    }

    void step(const float* act, const int* next_goal, double* obs, double* reward, uint8_t* term, uint8_t* trunc,
              uint8_t* success, float* info, double* final_obs, double* ep_ret, int* ep_len) override {
        if (!act) throw std::invalid_argument("step: actions are required");
        need_reset_done("step");
        if (next_goal) { check_goals(next_goal, nullptr, "step"); h_next_goal_.assign(next_goal, next_goal + N_); }
        Backend::h2d_async(d_act_, act, sizeof(float) * 4 * N_);
        if (next_goal) Backend::h2d_async(d_next_goal_, h_next_goal_.data(), sizeof(int) * N_);
        book_slot_ ^= 1;
        World<T> w = world();
        launch_step(w);
        // all outputs are queued behind the kernel on the context's stream and drained by ONE synchronisation
        const int D = obs_dim();
        if (obs) Backend::d2h_async(obs, d_obs_, sizeof(double) * N_ * D);
        if (reward) Backend::d2h_async(reward, d_reward_, sizeof(double) * N_);
        if (term) Backend::d2h_async(term, d_flags_, N_);
        if (trunc) Backend::d2h_async(trunc, d_flags_ + N_, N_);
        if (success) Backend::d2h_async(success, d_flags_ + 2 * N_, N_);
        if (info) Backend::d2h_async(info, d_info_, sizeof(float) * 6 * N_);
        if (final_obs) Backend::d2h_async(final_obs, d_final_, sizeof(double) * N_ * D);
        if (ep_ret) Backend::d2h_async(ep_ret, d_epret_, sizeof(double) * N_);
        if (ep_len) Backend::d2h_async(ep_len, d_eplen_, sizeof(int) * N_);
        Backend::sync();
    }

    // device-resident boundary: actions / goal indices are device pointers, outputs are written straight into the caller's
    // device buffers (any of them null = the context's own buffer); no host copies, one stream sync before returning
    // async (mw_step_device_on): ordered against the caller's stream by events instead of a host synchronisation; the `done` row is
    // copied to pinned host memory behind the kernel, wait_done() hands it over when it has landed
    void step_device(const float* d_act, const int* d_next_goal, const mw_device_out* out, bool async, void* caller_stream) override {
        need_reset_done("step_device");
        if (async) Backend::wait_for_caller(caller_stream);
        book_slot_ ^= 1;
        World<T> w = world();
        w.io.act = d_act;
        if (d_next_goal) w.io.next_goal = const_cast<int*>(d_next_goal);
        if (out) {
            if (out->obs) w.io.obs = out->obs;
            if (out->reward) w.io.reward = out->reward;
            if (out->flags) { w.io.terminated = out->flags; w.io.truncated = out->flags + N_; w.io.success = out->flags + 2 * N_; w.io.done = out->flags + 3 * N_; }
            if (out->info) w.io.info = out->info;
            if (out->final_obs) w.io.final_obs = out->final_obs;
            if (out->episode_return) w.io.ep_ret = out->episode_return;
            if (out->episode_length) w.io.ep_len = out->episode_length;
        }
        launch_step(w);
        if (!async) { Backend::sync(); return; }
        if (!h_done_) h_done_ = (uint8_t*)Backend::alloc_host(N_);
        Backend::d2h_async(h_done_, w.io.done, N_);
        Backend::record_done();
        Backend::caller_waits_for_us(caller_stream);
    }
    const uint8_t* wait_done() override {
        if (!h_done_) throw std::logic_error("wait_done: no mw_step_device_on call to wait for");
        Backend::wait_done();
        return h_done_;
    }
    void reset_device(const uint8_t* d_mask, const int* d_goal_idx, double* d_obs) override {
        // a full reset (no mask) satisfies the step-before-reset guard; a DEVICE mask cannot be inspected without a copy: it is copied
        // back (N bytes, blocking) ONLY while some env has never been reset -- once all have, the guard has nothing left to learn and
        // a masked device reset costs no host round trip
        if (!d_mask) was_reset_.assign(N_, 1);
        else if (std::find(was_reset_.begin(), was_reset_.end(), (uint8_t)0) != was_reset_.end()) {
            std::vector<uint8_t> hm(N_);
            Backend::d2h(hm.data(), d_mask, N_);
            for (int i = 0; i < N_; i++) if (hm[i]) was_reset_[i] = 1;
        }
        World<T> w = world();
        w.io.next_goal = const_cast<int*>(d_goal_idx);
        if (d_obs) w.io.obs = d_obs;
        Backend::launch(nblocks_, [w, d_mask] MW_LAMBDA(int b, int t, Scratchpad sp) { lane_reset_snap(w, d_mask, b, t, sp); });
        Backend::sync();
    }

    // ---- scripted policies on the device (generated from metaworld_amd/policies.py) ----
    struct PolicyState {       // per-env device arrays of a closed-loop rollout
        int* policy_id; int* schedule; int K; int* episodes; int* successes; uint8_t* ever;
    };
    // one thread per env: book the step that just ran (an episode counts as solved if success was ever 1 in it), pick the goal
    // of the env's next auto-reset from its schedule, then act on the observation the step returned
    static MW_HD void policy_thread(const IOPtrs& io, const PolicyState& ps, int N, int gid, bool account, bool act) {
        if (account) {
            uint8_t ever = ps.ever[gid] | io.success[gid];
            if (io.done[gid]) { ps.successes[gid] += ever ? 1 : 0; ps.episodes[gid] += 1; ever = 0; }
            ps.ever[gid] = ever;
        }
        if (!act) return;
        int k = ps.episodes[gid] + 1;
        const_cast<int*>(io.next_goal)[gid] = ps.schedule[(size_t)(k < ps.K ? k : ps.K - 1) * N + gid];
        scripted_policy(ps.policy_id[gid], io.obs + (size_t)gid * io.D, const_cast<float*>(io.act) + (size_t)gid * 4);
    }

    void policy_actions(const int* policy_id, const double* obs, float* act) override {
        const int D = obs_dim();
        int* d_pid = (int*)Backend::alloc(sizeof(int) * N_);
        Backend::h2d(d_pid, policy_id, sizeof(int) * N_);
        Backend::h2d(d_obs_, obs, sizeof(double) * N_ * D);
        const double* d_o = d_obs_; float* d_a = d_act_;
        Backend::launch_flat(N_, [d_pid, d_o, d_a, D] MW_LAMBDA(int gid) { scripted_policy(d_pid[gid], d_o + (size_t)gid * D, d_a + (size_t)gid * 4); });
        Backend::sync();
        Backend::d2h(act, d_act_, sizeof(float) * 4 * N_);
        Backend::free(d_pid);
    }

    // closed loop entirely on the device: reset to schedule[0], then nsteps x (policy kernel -> step kernel); the k-th
    // auto-reset of env i takes goal schedule[min(k, K-1)][i].  No host round trip inside the loop.
    void policy_rollout(const int* policy_id, const int* schedule, int K, int nsteps, int* episodes, int* successes, float* kernel_ms, int per_launch) override {
        if (K < 1) throw std::invalid_argument("policy_rollout: the goal schedule needs at least one row");
        PolicyState ps{};
        ps.policy_id = (int*)Backend::alloc(sizeof(int) * N_); ps.schedule = (int*)Backend::alloc(sizeof(int) * N_ * K); ps.K = K;
        ps.episodes = (int*)Backend::alloc(sizeof(int) * N_); ps.successes = (int*)Backend::alloc(sizeof(int) * N_);
        ps.ever = (uint8_t*)Backend::alloc(N_);
        Backend::h2d(ps.policy_id, policy_id, sizeof(int) * N_);
        Backend::h2d(ps.schedule, schedule, sizeof(int) * N_ * K);
        Backend::zero(ps.episodes, sizeof(int) * N_); Backend::zero(ps.successes, sizeof(int) * N_); Backend::zero(ps.ever, N_);
        reset(nullptr, schedule, nullptr);
        World<T> w = world();
        const IOPtrs io = w.io; const int N = N_;
        Backend::timed_begin();
        if (per_launch <= 1) {
            for (int s = 0; s < nsteps; s++) {
                const bool account = s > 0;
                Backend::launch_flat(N_, [io, ps, N, account] MW_LAMBDA(int gid) { policy_thread(io, ps, N, gid, account, true); });
                launch_step(w);
            }
        } else {
            // several (policy, step) pairs per launch: the policy of an environment is evaluated by the environment's own writer thread
            // between two of its steps (the observation it reads and the action it writes travel through the same buffers as in the
            // two-kernel loop; its sub-lanes are lanes of the same wave: MW_SYNC orders them).  Same results, no batch-wide
            // synchronisation between the steps of a launch.
            for (int s0 = 0; s0 < nsteps; s0 += per_launch) {
                const int n = nsteps - s0 < per_launch ? nsteps - s0 : per_launch;
                Backend::launch(nblocks_, [w, ps, N, s0, n] MW_LAMBDA(int b, int t, Scratchpad sp) {
                    Env<T> e; int gid;
                    const bool have = locate(w, b, t, sp, &e, &gid);
                    for (int k = 0; k < n; k++) {
                        if (have && e.sub == 0 && !e.ghost) policy_thread(w.io, ps, N, gid, s0 + k > 0, true);
                        MW_SYNC();
                        lane_step(w, b, t, sp);
                        MW_SYNC();
                    }
                });
            }
        }
        if (nsteps > 0) Backend::launch_flat(N_, [io, ps, N] MW_LAMBDA(int gid) { policy_thread(io, ps, N, gid, true, false); });
        const float ms = Backend::timed_end();
        if (kernel_ms) *kernel_ms = ms;
        if (episodes) Backend::d2h(episodes, ps.episodes, sizeof(int) * N_);
        if (successes) Backend::d2h(successes, ps.successes, sizeof(int) * N_);
        Backend::free(ps.policy_id); Backend::free(ps.schedule); Backend::free(ps.episodes); Backend::free(ps.successes); Backend::free(ps.ever);
    }

    void upload_actions(const float* act, int nsteps) override {
        if ((size_t)nsteps > act_capacity_steps_) {
            Backend::free(d_act_);
            d_act_ = (float*)Backend::alloc(sizeof(float) * 4 * N_ * nsteps);
            mw_log_range("act", nsteps, d_act_, sizeof(float) * 4 * N_ * nsteps);
            act_capacity_steps_ = nsteps;
        }
        Backend::h2d(d_act_, act, sizeof(float) * 4 * N_ * nsteps);
    }

    // bench path: actions already resident on the device; outputs stay on the device.  With gather = true the per-step
    // cross-rank bookkeeping all-gather (RCCL, side stream) is part of the loop: the kernel of step k+1 overlaps the
    // collective of step k (two record slots), and the call returns when both streams have drained.
    void step_device_only(const float* d_act, int nsteps, int act_steps, float* kernel_ms, bool gather = false) override {
        need_reset_done("step_resident");
        if (gather && world_size() > 1 && !comm_) throw std::logic_error("step_resident_gather: call mw_comm_init first");
        const float* base = d_act ? d_act : d_act_;
        Backend::timed_begin();
        for (int s = 0; s < nsteps; s++) {
            book_slot_ ^= 1;
            // two record slots: the kernel of step k rewrites the slot the gather of step k-2 read.  That gather runs on the side
            // stream and may lag (first-call RCCL setup, a slow rank): the main stream waits for its "gather done" event.
            if (gather) Backend::wait_gather_done(book_slot_);
            World<T> w = world();
            w.io.act = base + (size_t)(act_steps > 0 ? s % act_steps : 0) * 4 * N_;
            if (d_sched_) { w.io.sched = d_sched_; w.io.sched_pos = d_sched_pos_; w.io.sched_K = sched_K_; }
            launch_step(w);
            Backend::timed_mark();
            if (gather) gather_async();
        }
        float ms = Backend::timed_end();
        if (gather) Backend::sync_side();
        if (kernel_ms) *kernel_ms = ms;
    }
    void step_resident_gather(int nsteps, int act_steps, float* kernel_ms) override { step_device_only(nullptr, nsteps, act_steps, kernel_ms, true); }

    // OPEN-LOOP rollout with several steps per launch: every thread runs `per_launch` consecutive steps of its environment before
    // the kernel ends.  Environments are independent (the only cross-environment effect of a step is the status word), so the
    // results equal those of the per-step loop bit for bit -- but the batch is synchronised once per launch instead of once per
    // step: "the slowest wave's SUM over the steps" instead of "the sum over the steps of the slowest wave".  Only for callers that
    // do not look at the batch between steps (pre-uploaded actions; the observation / reward / flag buffers and the bookkeeping
    // record hold the LAST step of the launch, auto-resets follow the goal schedule or next_goal as in step_device_only).
    void step_fused(int nsteps, int act_steps, int per_launch, float* kernel_ms) override {
        need_reset_done("step_resident_fused");
        if (per_launch < 1) throw std::invalid_argument("step_resident_fused: steps_per_launch must be >= 1");
        const float* base = d_act_;
        const int N = N_;
        Backend::timed_begin();
        for (int s0 = 0; s0 < nsteps; s0 += per_launch) {
            const int n = nsteps - s0 < per_launch ? nsteps - s0 : per_launch;
            World<T> w = world();
            if (d_sched_) { w.io.sched = d_sched_; w.io.sched_pos = d_sched_pos_; w.io.sched_K = sched_K_; }
            Backend::launch(nblocks_, [w, base, s0, n, act_steps, N] MW_LAMBDA(int b, int t, Scratchpad sp) {
                World<T> ws = w;
                for (int k = 0; k < n; k++) {
                    ws.io.act = base + (size_t)(act_steps > 0 ? (s0 + k) % act_steps : 0) * 4 * N;
                    lane_step(ws, b, t, sp);
                    MW_SYNC();          // step boundary: the sub-lanes' read-then-advance of sched_pos, the task-block counters and the snapshot copy of an auto-reset are ordered for the next step (as in policy_rollout's fused loop)
                }
            });
        }
        const float ms = Backend::timed_end();
        if (kernel_ms) *kernel_ms = ms;
    }

    // ---- cross-rank bookkeeping ----
    void comm_init(const void* id128, int rank, int world) override {
        if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("comm_init: bad rank / world size");
        Backend::comm_free(comm_); comm_ = nullptr;
        // world size 1 needs no communicator (the gather is a device copy); MW_COMM_FORCE_RCCL=1 creates a one-rank RCCL communicator
        // anyway, so that ncclCommInitRank / ncclAllGather run on a single-GPU box (tests/test_gpu_fullsize.py)
        if (world > 1 || (id128 && getenv("MW_COMM_FORCE_RCCL"))) comm_ = Backend::comm_init(id128, rank, world);
        Backend::free(d_book_all_);
        d_book_all_ = (mw_bookkeeping*)Backend::alloc(sizeof(mw_bookkeeping) * 2 * (size_t)world * N_);
        Backend::zero(d_book_all_, sizeof(mw_bookkeeping) * 2 * (size_t)world * N_);
        Backend::sync();
        cfg.rank = rank; cfg.world_size = world;
    }
    void comm_info(int* out) override { Backend::comm_info(comm_, out); }
    mw_bookkeeping* gathered() const { return d_book_all_ + (size_t)book_slot_ * world_size() * N_; }
    void gather_async() {          // records of the step just queued on the main stream -> every rank, on the side stream
        const mw_bookkeeping* src = d_book_ + (size_t)book_slot_ * N_;
        Backend::allgather_side(comm_, src, gathered(), sizeof(mw_bookkeeping) * (size_t)N_, book_slot_);
    }
    void gather_bookkeeping(mw_bookkeeping* out, int out_on_device) override {
        gather_async();
        const size_t bytes = sizeof(mw_bookkeeping) * (size_t)world_size() * N_;
        if (out) Backend::copy_side(out, gathered(), bytes, out_on_device != 0);
        Backend::sync_side();
    }
    // TimeLimit._elapsed_steps and curr_path_length of every env := elapsed[i]: env i then truncates (and auto-resets) after
    // max_episode_steps - elapsed[i] more steps.  Staggers the synchronised auto-reset waves of a freshly reset batch.
    void set_episode_phase(const int* elapsed) override {
        for (int i = 0; i < N_; i++) {
            if (elapsed[i] < 0 || elapsed[i] >= cfg.max_episode_steps) throw std::out_of_range("set_episode_phase: elapsed steps outside [0, max_episode_steps)");
            set_task_field(groups_[env_group_[i]], env_lane_[i], TK_ELAPSED, elapsed[i]);
            set_task_field(groups_[env_group_[i]], env_lane_[i], TK_PATHLEN, elapsed[i]);
        }
    }
    // goal schedule of the resident loop (IOPtrs::sched): K rows of N goal indices, or null / K = 0 to go back to next_goal
    void set_goal_schedule(const int* schedule, int K) override {
        Backend::sync();
        Backend::free(d_sched_); d_sched_ = nullptr; sched_K_ = 0;
        if (!schedule || K <= 0) return;
        for (int k = 0; k < K; k++) check_goals(schedule + (size_t)k * N_, nullptr, "set_goal_schedule");
        d_sched_ = (int*)Backend::alloc(sizeof(int) * (size_t)N_ * K);
        Backend::h2d(d_sched_, schedule, sizeof(int) * (size_t)N_ * K);
        if (!d_sched_pos_) d_sched_pos_ = (int*)Backend::alloc(sizeof(int) * N_);
        Backend::zero(d_sched_pos_, sizeof(int) * N_);
        Backend::sync();
        sched_K_ = K;
    }
    void goal_schedule_pos(int* out) override {
        if (!d_sched_pos_) { for (int i = 0; i < N_; i++) out[i] = 0; return; }
        Backend::sync();
        Backend::d2h(out, d_sched_pos_, sizeof(int) * N_);
    }
    void status(int* out, int n, int clear) override {
        int all[MW_STATUS_WORDS];
        Backend::d2h(all, d_status_, sizeof(int) * MW_STATUS_WORDS);
        for (int k = 0; k < n; k++) out[k] = k < MW_STATUS_WORDS ? all[k] : 0;          // the caller says how many words it has room for
        if (clear) { Backend::zero(d_status_, sizeof(int) * MW_STATUS_WORDS); Backend::sync(); }
    }
    int launch_times(float* out, int cap) override { return Backend::launch_times(out, cap); }
    void set_option(const std::string& name, double value) override {
        if (name == "split_collision") {
#if defined(MW_SPLIT_COLLISION)
            Backend::sync(); split_collision_ = value != 0 ? 1 : 0;
#else
            if (value != 0) throw std::invalid_argument("set_option: split_collision needs a library built with -DMW_SPLIT_COLLISION (tools/build_variants.sh)");
#endif
        }
        else throw std::invalid_argument("set_option: unknown option " + name);
    }

    void debug(int what, int n) override {
        World<T> w = world();
        Backend::launch(nblocks_, [w, what, n] MW_LAMBDA(int b, int t, Scratchpad sp) { lane_debug(w, what, n, b, t, sp); });
        Backend::sync();
    }

    static int offset_of(const Layout& L, const Sizes& s, const std::string& k, int* n) {
        struct E { const char* name; int off, n; };
        const E tab[] = {{"qpos", L.qpos, s.nq}, {"qvel", L.qvel, s.nv}, {"warm", L.warm, s.nv}, {"ctrl", L.ctrl, s.nu}, {"mocap", L.mocap, 3},
            {"reloc", L.reloc, 3 * (s.nreloc > 0 ? s.nreloc : 1)}, {"time", L.time, 1}, {"task", L.task, TASK_NREAL}, {"state", 0, L.nstate},
            {"xpos", L.xpos, 3 * s.nbody}, {"xquat", L.xquat, 4 * s.nbody}, {"xmat", L.xmat, 9 * s.nbody}, {"xipos", L.xipos, 3 * s.nbody},
            {"geom_xpos", L.geom_xpos, 3 * s.ngeom}, {"geom_xmat", L.geom_xmat, 9 * s.ngeom}, {"cdof", L.cdof, 6 * s.nv},
            {"qM", L.qM, s.nv * s.nv}, {"qL", L.qL, s.nv * s.nv}, {"qH", L.qH, s.nv * s.nv}, {"bias", L.bias, s.nv}, {"smooth", L.smooth, s.nv},
            {"qacc_smooth", L.qacc_smooth, s.nv}, {"qfrc_constraint", L.qfrc_c, s.nv}, {"qacc", L.qacc, s.nv},
            {"con", L.con, CON_STRIDE * s.maxcon}, {"efcJ", L.efcJ, s.nv * s.maxefc}, {"efcX", L.efcX, EFC_EXTRA * s.maxefc}};
        for (auto& e : tab) if (k == e.name) { *n = e.n; return e.off; }
        throw std::runtime_error("unknown column " + k);
    }
    static int ioffset_of(const Layout& L, const Sizes& s, const std::string& k, int* n) {
        if (k == "icon") { *n = CON_ISTRIDE * s.maxcon; return L.icon; }
        if (k == "iefc") { *n = EFC_ISTRIDE * s.maxefc; return L.iefc; }
        if (k == "icount") { *n = IC_SIZE; return L.icount; }
        throw std::runtime_error("unknown int column " + k);
    }
    int layout_size(int gid, const char* what) override {
        int n; const Group& g = groups_.at(env_group_.at(gid));
        try { offset_of(g.L, models[g.model]->sz, what, &n); } catch (...) { ioffset_of(g.L, models[g.model]->sz, what, &n); }
        return n;
    }
    // persistent state (Layout::nstate reals: qpos, qvel, warm start, ctrl, mocap, relocated bodies, time, task / episode block) of
    // EVERY environment in one pass: the whole column store of a group travels once, instead of one copy per element and env
    void state_all(double* buf, int stride, bool write) override {
        Backend::sync();
        for (auto& g : groups_) {
            std::vector<T> host(g.nreal_total());
            Backend::d2h(host.data(), g.col, sizeof(T) * host.size());
            if (g.L.nstate > stride) throw std::invalid_argument("state_all: stride smaller than an environment's state (mw_column_size(env, \"state\"))");
            for (int l = 0; l < g.nenv; l++) {
                double* row = buf + (size_t)g.gid[l] * stride;
                if (write) for (int k = 0; k < g.L.nstate; k++) host[g.at(k, l)] = (T)row[k];
                else for (int k = 0; k < g.L.nstate; k++) row[k] = (double)host[g.at(k, l)];
            }
            if (write) Backend::h2d(g.col, host.data(), sizeof(T) * host.size());
        }
    }
    void read_col(int gid, const char* what, int n, double* out) override {
        const Group& g = groups_.at(env_group_.at(gid));
        int cnt; const int off = offset_of(g.L, models[g.model]->sz, what, &cnt);
        if (n > cnt) n = cnt;
        for (int k = 0; k < n; k++) { T x; Backend::d2h(&x, g.col + g.at(off + k, env_lane_[gid]), sizeof(T)); out[k] = (double)x; }
    }
    void write_col(int gid, const char* what, int n, const double* in) override {
        Group& g = groups_.at(env_group_.at(gid));
        int cnt; const int off = offset_of(g.L, models[g.model]->sz, what, &cnt);
        if (n > cnt) n = cnt;
        for (int k = 0; k < n; k++) { T x = (T)in[k]; Backend::h2d(g.col + g.at(off + k, env_lane_[gid]), &x, sizeof(T)); }
    }
    void read_icol(int gid, const char* what, int n, int* out) override {
        const Group& g = groups_.at(env_group_.at(gid));
        int cnt; const int off = ioffset_of(g.L, models[g.model]->sz, what, &cnt);
        if (n > cnt) n = cnt;
        for (int k = 0; k < n; k++) Backend::d2h(out + k, g.icol + g.iat(off + k, env_lane_[gid]), sizeof(int));
    }
};

}  // namespace mw
