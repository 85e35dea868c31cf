// mw_common.hpp -- shared definitions for the MI355X batched Meta-World runtime.
//
// Execution model: ONE ENVIRONMENT PER LANE, PLUS SUB-LANES.  A workgroup is one 64-thread wave carrying `lpb`
// environments; thread t works for environment t % lpb as its sub-lane t / lpb (lpb = 64: no sub-lanes).  All
// per-environment data (persistent state and scratch) lives in a struct-of-arrays "column store" chunked per workgroup:
// element `i` of the environment in lane `l` of chunk `c` is at  col[(c * nreal + i) * lpb + l]  (Env::R), so a wave's
// access to one element is one contiguous request and consecutive elements are adjacent.  Every wavefront is
// model-uniform (groups are per compiled model), so the model tables are read at wave-uniform addresses (scalar loads).
// The solver's per-row scalars live in an LDS scratchpad, its dense algebra in registers (mw_phys.hpp); the row /
// pair / dof sweeps of one environment are split over its sub-lanes (MW_SUBS below).
//
// The same headers compile for the device (hipcc, gfx950) and -- for tests only --
// for the host (g++), where a plain loop over lanes replaces the wavefront.  The
// host build is a test harness (tests/host_harness.cpp); the product path is the
// HIP library and nothing falls back to the CPU.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MW_HD __host__ __device__ inline
#define MW_STAGE_FN __host__ __device__ __attribute__((noinline))   // the big stages: one copy each (code size, compile time)
#define MW_FP_EXACT _Pragma("clang fp contract(off)")   // first statement of a block: no fused multiply-add (bit-exact restatements)
#else
#define MW_HD inline
#define MW_STAGE_FN inline
#define MW_FP_EXACT   // the host harness is built with -ffp-contract=off
#endif

namespace mw {

// solver phase timers (shader clock), only in -DMW_SOLVER_TIMING device builds: accumulated in icount[4..11]
#if defined(MW_SOLVER_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define MW_TICK(var) const long long var = (long long)__builtin_amdgcn_s_memtime();
#if defined(MW_COLL_TIMING)
#define MW_TOCK(e, L, slot, t0, t1) if ((slot) >= 8) e.I(L.icount + 4 + (slot)) += (int)(((t1) - (t0)) >> 4);
#define MW_TADD(e, L, slot, v)
#else
#define MW_TOCK(e, L, slot, t0, t1) e.I(L.icount + 4 + (slot)) += (int)(((t1) - (t0)) >> 4);
#define MW_TADD(e, L, slot, v) e.I(L.icount + 4 + (slot)) += (v);
#endif
#else
#define MW_TADD(e, L, slot, v)
#define MW_TICK(var)
#define MW_TOCK(e, L, slot, t0, t1)
#endif

// -DMW_COLL_TIMING (with -DMW_SOLVER_TIMING): the eight solver-phase slots are given to the collision stage instead
// (0 mid phase, 1 narrow phase [cycles / 16]; 2 candidate pairs, 3 narrow-phase rounds [counts]); tools/experiments/coll_timing.py
#if defined(MW_COLL_TIMING) && defined(MW_SOLVER_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define MW_CTICK(var) const long long var = (long long)__builtin_amdgcn_s_memtime();
#define MW_CTOCK(e, L, slot, t0, t1) e.I(L.icount + 4 + (slot)) += (int)(((t1) - (t0)) >> 4);
#define MW_CADD(e, L, slot, v) e.I(L.icount + 4 + (slot)) += (v);
// per-branch cycles inside collide_pair (slots 4 box-box, 5 portal refinement, 6 the other closed-form routines, 7 face upgrade):
// a lane records the duration of the branches IT takes; the wave executes every branch some lane takes, one after the other
#define MW_CP_EXTRA , int* tstat
#define MW_CP_PASS(x) , x
#define MW_CSTAT(k, t0, t1) tstat[k] += (int)(((t1) - (t0)) >> 4);
#else
#define MW_CP_EXTRA
#define MW_CP_PASS(x)
#define MW_CSTAT(k, t0, t1)
#define MW_CTICK(var)
#define MW_CTOCK(e, L, slot, t0, t1)
#define MW_CADD(e, L, slot, v)
#endif
// iteration counters / histograms of the host profile build (tests/host_harness.cpp with -DMW_PROFILE)
#if defined(MW_PROFILE) && !defined(__HIPCC__)
inline long* mw_cnt() { static long c[8] = {0}; return c; }
#define MW_COUNT(i) mw_cnt()[i]++;
inline long* mw_hist() { static long h[4 * 64] = {0}; return h; }
#define MW_HIST(w, v) mw_hist()[(w) * 64 + ((v) < 63 ? (v) : 63)]++;
// per geom-type-pair narrow-phase statistics: [t1][t2][0 calls, 1 portal iterations, 2 hill-climb steps, 3 re-shot mpr() runs]
inline long* mw_pairstat() { static long p[8 * 8 * 4] = {0}; return p; }
inline int& mw_pair_cur() { static thread_local int c = 0; return c; }
#define MW_PAIR_BEGIN(t1, t2) { mw_pair_cur() = ((t1) * 8 + (t2)) * 4; mw_pairstat()[mw_pair_cur()]++; }
#define MW_PAIR_ADD(k, v) { mw_pairstat()[mw_pair_cur() + (k)] += (v); }
#else
#define MW_COUNT(i) {}
#define MW_HIST(w, v) {}
#define MW_PAIR_BEGIN(t1, t2) {}
#define MW_PAIR_ADD(k, v) {}
#endif


// Address spaces and uniformity (device only).  Pointers reach the lane code as generic (flat) pointers, which cost
// flat_load/flat_store and hide wave-uniformity from the compiler.  The column store is therefore accessed through
// references into the GLOBAL address space (global_load/store), and the model tables -- read at wave-uniform
// addresses, every wave being model-uniform -- through the CONSTANT address space (scalar s_load into SGPRs).
// mw_uniform() (v_readfirstlane) tells the compiler a value is wave-uniform.
#if defined(__HIP_DEVICE_COMPILE__)
#define MW_GLOBAL __attribute__((address_space(1)))
#define MW_LDS __attribute__((address_space(3)))
#define MW_CONST __attribute__((address_space(4)))
__device__ inline unsigned mw_uniform(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
__device__ inline int mw_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ inline unsigned long long mw_uniform(unsigned long long v) {
    const unsigned lo = mw_uniform((unsigned)v), hi = mw_uniform((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ inline float mw_uniform(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }
__device__ inline double mw_uniform(double x) { return __builtin_bit_cast(double, mw_uniform(__builtin_bit_cast(unsigned long long, x))); }
#else
#define MW_GLOBAL
#define MW_CONST
#define MW_LDS
inline float mw_uniform(float x) { return x; }
inline double mw_uniform(double x) { return x; }
inline unsigned mw_uniform(unsigned x) { return x; }
inline int mw_uniform(int x) { return x; }
inline unsigned long long mw_uniform(unsigned long long v) { return v; }
#endif
template <typename T> using GRef = MW_GLOBAL T&;          // reference into the column store
template <typename T> using CP = const MW_CONST T*;       // read-only model table

enum { G_PLANE = 0, G_HFIELD, G_SPHERE, G_CAPSULE, G_ELLIPSOID, G_CYLINDER, G_BOX, G_MESH };
enum { J_FREE = 0, J_BALL, J_SLIDE, J_HINGE };
enum { C_EQUALITY = 0, C_LIMIT = 3, C_CONTACT = 7 };
enum { S_SATISFIED = 0, S_QUADRATIC = 1, S_CONE = 4 };

constexpr int MAX_NV = 17;       // register-resident solver arrays are sized by this (NV_LARGE in mw_phys.hpp)
constexpr int CON_STRIDE = 26;   // reals per contact record
constexpr int CON_ISTRIDE = 4;   // ints per contact record
constexpr int EFC_EXTRA = 11;    // reals per constraint row besides J: pos, margin, R, D, aref, force, jar, Jv, fri, info, state
constexpr int SR_N = 9;          // solver row scalars kept in the LDS scratchpad: D, jar, Jv, fri, info, state, force, block list, aref; the row's Jacobian follows them
constexpr int CELL_GRID = 16;     // support cells of a hull: cube map of directions, 6 faces x CELL_GRID x CELL_GRID (metaworld_amd/hullcells.py GRID)
constexpr int CELL_N = 6 * CELL_GRID * CELL_GRID;
constexpr int CELL_K = 8;         // vertices of a cell's list stored at the cell's fixed place (Model::mesh_cellxyz); longer lists continue in Model::mesh_ovfxyz
constexpr int TLS_SLOTS = 60;    // thread-private scratchpad slots of the narrow phase: two polygons of up to 10 vertices (box-box face clipping)
constexpr int IC_NBLK = 18;      // icount slot: number of constraint blocks (single rows / contact cones)
constexpr int IC_SIZE = 26;      // ints in icount: 0 ncon, 1 nefc, 2 niter, 3 flags, 4..17 phase timers (timing builds), 18 nblk, 19 dynamics valid, 20 / 21 demand, 22 Euler acceleration ready, 23 solver stalls, 24 / 25 split collision
constexpr int IC_NCAND = 24;     // icount slot (split collision, mw_split.hpp): candidate pairs the mid-phase kernel listed for this environment (L.ipair / L.iitem)
constexpr int IC_PENDING = 25;   // icount slot (split collision): 1 = the step waits for the lazy final dynamics (its narrow phase is on the way)
constexpr int IC_DYN_VALID = 19;  // icount slot: 1 = contacts / constraint rows / solver output belong to the CURRENT qpos (see env_step)
constexpr int IC_WANT_CON = 20, IC_WANT_EFC = 21;   // running maxima of the contacts / constraint rows a step WANTED (capacity planning)
constexpr int EFC_ISTRIDE = 5;   // type, id, state, first and last dof with a non-zero Jacobian entry
// per-env status bits of one step (icount[3]); OR-ed into the context status word (mw_status) by lane_step:
//   1 / 2  constraint-row / contact capacity exceeded (rows / contacts were dropped)
//   4      non-finite state, the env was truncated and reset
//   8      CANARY: the copies of a redundantly computed value held by the sub-lanes of one environment disagreed (sub_disagree)
//  16      (MW_BOUNDS debug builds) a column-store / scratchpad index was out of range
enum { ST_ROW_OVERFLOW = 1, ST_CON_OVERFLOW = 2, ST_UNSTABLE = 4, ST_DIVERGED = 8, ST_OOB = 16 };
constexpr int IC_EULER_READY = 22;    // icount slot: 1 = L.search holds the acceleration of the semi-implicit Euler step for the CURRENT solver output (solve_wave's finish); integrate consumes it
constexpr int IC_SOLVER_STALL = 23;   // icount slot: line searches abandoned on a non-descent direction (this step)

struct Sizes {
    int nq, nv, nbody, njnt, ngeom, nsite, nmesh, nmeshvert, npair, nu, neq, nprobe, nreloc;
    int maxcon, maxefc, iterations, ls_iterations;
};

// Offsets (in elements) of every per-environment array inside the column store.
struct Layout {
    // persistent state
    int qpos, qvel, warm, ctrl, mocap, reloc, time, task;   // task: TASK_NREAL reals of task/episode state
    int nstate;                                              // number of persistent reals
    // scratch
    int xpos, xquat, xmat, xipos, cinert, crb, geom_xpos, geom_xmat, cdof, qM, qL, qH;
    int cvel, cacc, cfrc, bias, smooth, qacc_smooth, qfrc_c, qacc, Ma, grad, search, Mv;
    int con, efcJ, efcX;
    int nreal;
    // int columns
    int icon, iefc, icount;   // icount: ncon, nefc, niter, flags
    int iwork;                // constraint work items of make_constraints: (kind, id, first row) x maxefc
    int ipair;                // collision candidates (pair indices that passed the broad / mid phase), npair
    int iitem;                // split collision: work-item id of every candidate (where the narrow-phase kernel left its hits), npair
    int nint;
};

// Device-resident model tables (one per compiled MJCF scene).  Field names follow
// metaworld_amd/mjcf.py (which follows MuJoCo's mjModel).
template <typename T>
struct Model {
    Sizes sz;
    T timestep, tolerance, meaninertia, gravity[3];
    // ints
    CP<int> body_parentid, body_mocap, body_jntadr, body_jntnum, body_lastdof, body_relocid;
    CP<int> jnt_type, jnt_bodyid, jnt_qposadr, jnt_dofadr, jnt_limited;
    CP<int> dof_bodyid, dof_jntid, dof_parentid;
    CP<int> geom_type, geom_bodyid, geom_meshid, geom_condim;
    CP<int> mesh_vertadr, mesh_vertnum, pair_geom;
    CP<int> act_dofid, act_qposid, eq_body1, eq_body2, probe_body;
    // reals
    CP<T> body_pos, body_quat, body_ipos, body_iquat, body_mass, body_inertia;
    CP<T> jnt_pos, jnt_axis, jnt_range, jnt_stiffness, jnt_springref, jnt_solref, jnt_solimp, jnt_margin;
    CP<T> dof_armature, dof_damping, dof_invweight0, qpos0;
    CP<T> geom_size, geom_pos, geom_quat, geom_friction, geom_solref, geom_solimp, geom_solmix;
    CP<T> geom_margin, geom_gap, geom_rbound, geom_invweight0, geom_aabb;
    CP<T> mesh_vert, act_kp, act_ctrlrange, eq_solref, eq_solimp, eq_data, eq_invweight0;
    CP<T> probe_pos, probe_quat;
    // derived at upload (DeviceModel) from the support cells of the hulls (metaworld_amd/hullcells.py: per cube-map cell of
    // directions, the ascending list of the vertices that can be the support vertex for a direction of the cell): the
    // COORDINATES of the first CELL_K vertices of every list at a fixed place, mesh_cellxyz[(mesh * CELL_N + cell) * 3 CELL_K ...]
    // (a short list is padded with copies of its last vertex), so that a support call is ONE memory round trip at an address
    // computed from the direction alone.  The few longer lists continue in mesh_ovfxyz in batches of CELL_K (padded likewise):
    // mesh_cellovf[mesh * CELL_N + cell] = -1 or (first batch) * 16 + (number of batches).
    CP<int> mesh_cellovf;
    CP<T> mesh_cellxyz, mesh_ovfxyz;
    CP<int> body_dofmask;   // derived at upload: bit i = dof i lies on the chain from body b to the root (nv <= 31)
    CP<int> dof_qposadr;    // derived at upload: the qpos element that dof i integrates into by qpos += h * qvel (slide / hinge joints, the three translational dofs of a free joint); -1 for the rotational dofs of a free joint
    Layout L;        // make_layout(sz)
};

constexpr int TASK_NREAL = 96;   // per-env task/episode block, see mw_tasks.hpp

inline Layout make_layout(const Sizes& s) {
    Layout L;
    int o = 0;
    auto take = [&](int n) { int r = o; o += n; return r; };
    L.qpos = take(s.nq); L.qvel = take(s.nv); L.warm = take(s.nv); L.ctrl = take(s.nu > 0 ? s.nu : 1);
    L.mocap = take(3); L.reloc = take(3 * (s.nreloc > 0 ? s.nreloc : 1)); L.time = take(1); L.task = take(TASK_NREAL);
    L.nstate = o;
    L.xpos = take(3 * s.nbody); L.xquat = take(4 * s.nbody); L.xmat = take(9 * s.nbody); L.xipos = take(3 * s.nbody);
    L.cinert = take(10 * s.nbody); L.crb = take(10 * s.nbody);
    L.geom_xpos = take(3 * s.ngeom); L.geom_xmat = take(9 * s.ngeom);
    L.cdof = take(6 * s.nv); L.qM = take(s.nv * s.nv); L.qL = take(s.nv * s.nv); L.qH = take(s.nv * s.nv);
    L.cvel = take(6 * s.nbody); L.cacc = take(6 * s.nbody); L.cfrc = take(6 * s.nbody);
    L.bias = take(s.nv); L.smooth = take(s.nv); L.qacc_smooth = take(s.nv); L.qfrc_c = take(s.nv); L.qacc = take(s.nv);
    L.Ma = take(s.nv); L.grad = take(s.nv); L.search = take(s.nv); L.Mv = take(s.nv);
    L.con = take(CON_STRIDE * s.maxcon);
    L.efcJ = take(s.nv * s.maxefc);
    L.efcX = take(EFC_EXTRA * s.maxefc);
    L.nreal = o;
    o = 0;
    L.icon = take(CON_ISTRIDE * s.maxcon); L.iefc = take(EFC_ISTRIDE * s.maxefc); L.icount = take(IC_SIZE); L.iwork = take(3 * s.maxefc); L.ipair = take(s.npair > 0 ? s.npair : 1); L.iitem = take(s.npair > 0 ? s.npair : 1);   // + 16 phase timers (MW_SOLVER_TIMING builds): 8 solver phases, 6 pipeline stages
    L.nint = o;
    return L;
}

// workgroup scratchpad handed to every lane program: LDS on the device (stride = lanes per workgroup), a private
// buffer per host thread in the test harness (stride 1)
struct Scratchpad { MW_LDS void* base; int block_words, host_nsub, max_rows, chain; };   // 4-byte words of the whole workgroup; host_nsub > 0: host harness (one call per env, that many emulated sub-lanes); max_rows > 0: cap on the rows kept in the scratchpad (tests of the fallback rows); chain: 0 = never stage the body-level chains in the scratchpad (Env::chain_lds), else when they fit

// LDS of a flat wave kernel (mw_split.inl).  A struct, not a bare pointer parameter: the address-space qualifier exists only in the
// device pass, and a lambda whose PARAMETER TYPES differ between the host and the device pass gets two different kernel symbol names
struct WaveLds { MW_LDS void* base; };

template <typename T> using CModel = const MW_CONST Model<T>;
using CLayout = const MW_CONST Layout;

// Per-lane view of one environment.  Passed BY VALUE so the fields stay in registers across the non-inlined stage
// functions; each stage starts with `e = e_.uniform()`, which marks everything except the two lane pointers as
// wave-uniform ONCE (readfirstlane is not hoisted out of divergent control flow, so it must not sit in accessors).
template <typename T>
struct Env {
    const Model<T>* m;
    T* col;        // real column store, already offset by the lane
    int* icol;     // int column store, already offset by the lane
    unsigned stride;   // 32-bit index arithmetic: nreal * stride < 2^32 (checked at group creation)
    int nv, o_efcJ, o_efcX, o_con, o_icon, o_iefc, o_icount, o_task;   // hot layout offsets (copied from Layout)
    MW_LDS T* lds;     // this lane's slice of the workgroup scratchpad (LDS on the device), slot k at lds[k * lds_stride]
    int lds_stride;
    int sub, nsub;     // sub-lane of this thread and sub-lanes per environment (cooperative row sweeps, see below)
    int thr;           // thread index inside the workgroup
    // slot = which of the workgroup's environments this thread works for; nslot = how many the workgroup really holds.  The last
    // workgroup of a group can hold fewer than lpb: its surplus threads do not leave the kernel, they run as GHOSTS -- exact
    // duplicates of the threads of the last real environment (same slot, same sub-lane index, same values, same stores), so that
    // EVERY lane of every wave stays alive: the wave-cooperative sections (solve_wave, mw_solve_wave.hpp: matrix instructions and
    // lane-role loads that need all 64 lanes) and every non-inlined stage function then run under a full EXEC mask.
    int slot, nslot, ghost;
    int lds_rows;      // constraint rows that fit in the scratchpad: scalars + Jacobian row (the rest stay in the column store)
    int lds_w;         // scratchpad slots per row = SR_N + nv, rounded up to an odd number
    // BODY-LEVEL CHAINS IN THE SCRATCHPAD.  Kinematics, the composite inertias and the recursive Newton-Euler passes are chains
    // over the body tree in which every link reads what the previous one wrote: through the column store that is one L2 round
    // trip (~700 cycles) per link and array, through LDS ~100.  When the environment's share of the scratchpad is large enough
    // (chain_lds), its first lds_perm = 7 nv + nq slots hold cdof[6 nv] and copies of qvel[nv] and qpos[nq] for the whole dynamics
    // evaluation (read by the mass matrix, the bias forces, the joint limits and every constraint row), the constraint rows start behind them, and the slots of the
    // rows -- dead until make_constraints fills them -- carry the chains' transients first (mw_phys.hpp: body frames in
    // kinematics, composite inertias in crb, body velocities / accelerations / forces in smooth_forces).  Same values either way.
    int lds_perm;      // slots in front of the rows (0 without chain_lds)
    // THREAD-private scratchpad slice (slot k at tls[k * tls_stride]) for the narrow phase's polygon clipping, whose dynamically
    // indexed arrays would otherwise live in scratch memory (one L2 round trip per access); it overlays the rows' slots, which are
    // dead while the collision stage runs.  Null when the workgroup's scratchpad is too small (the arrays stay local then).
    MW_LDS T* tls;
    int tls_stride;
    int chain_lds;     // 0 = nothing of the above; 1 = the slots in front of the rows + the composite inertias (what fits a smaller share: 7 nv + nq + 10 nbody slots); 2 = all of it (+ 18 nbody slots)
#if defined(MW_BOUNDS)   // debug build: every column-store access is range-checked; a violation is recorded and redirected to element 0
    unsigned nreal_b, nint_b;
    int* oob;          // context status word: [0] |= ST_OOB, [1] = kind (1 real, 2 int, 3 scratchpad), [2] = index, [3] = limit
    MW_HD unsigned chk(int i, unsigned lim, int kind) const {
        if ((unsigned)i < lim) return (unsigned)i;
        if (oob) { oob[0] |= 16; oob[1] = kind; oob[2] = i; oob[3] = (int)lim; }
        return 0u;
    }
#endif
    // lpb = environments per workgroup of this environment's group; threads t, t + lpb, ... are its sub-lanes
    MW_HD void set_scratchpad(Scratchpad sp, int thread, int lpb, int nv_, int nbody, int nq_, int env_slot) {
        const bool host = sp.host_nsub > 0;
        lds = (MW_LDS T*)sp.base + (host ? 0 : env_slot);
        lds_stride = host ? 1 : lpb;
        sub = host ? 0 : thread / lpb;
        thr = thread;
        nsub = host ? sp.host_nsub : 64 / lpb;
        lds_w = (SR_N + nv_) | 1;          // odd: rows r, r + 1, ... of one field then start in different LDS banks (row-per-lane sweeps; an even width put all 16 rows of a sweep on the same 8 banks)
        const int words = (int)((host ? sp.block_words : sp.block_words / lpb) * 4 / sizeof(T));          // slots of this environment
        chain_lds = sp.chain == 0 ? 0 : (words >= 7 * nv_ + nq_ + 18 * nbody + 2 * lds_w ? 2 : (words >= 7 * nv_ + nq_ + 10 * nbody + 2 * lds_w ? 1 : 0));
        if (sp.chain > 0 && sp.chain < chain_lds) chain_lds = sp.chain;          // (tests / experiments: cap the level)
        lds_perm = chain_lds ? 7 * nv_ + nq_ : 0;
        lds_rows = (words - lds_perm) / lds_w;
        if (sp.max_rows > 0 && lds_rows > sp.max_rows) lds_rows = sp.max_rows;
        const int wave_words = (int)(sp.block_words * 4 / sizeof(T));          // slots of the whole workgroup
        tls_stride = host ? 1 : 64;
        tls = nullptr;
        if (host ? words >= lds_perm + TLS_SLOTS : wave_words >= lds_perm * lpb + TLS_SLOTS * 64)
            tls = (MW_LDS T*)sp.base + (host ? lds_perm : lds_perm * lpb + thread);
    }
    MW_HD void cache_layout(const Layout& L, int nv_) {
        nv = nv_; o_efcJ = L.efcJ; o_efcX = L.efcX; o_con = L.con; o_icon = L.icon; o_iefc = L.iefc; o_icount = L.icount; o_task = L.task;
    }
    MW_HD Env uniform() const {
        Env u = *this;
        u.m = (const Model<T>*)mw_uniform((unsigned long long)m);
        u.stride = mw_uniform(stride);
        u.nv = mw_uniform(nv); u.o_efcJ = mw_uniform(o_efcJ); u.o_efcX = mw_uniform(o_efcX); u.o_con = mw_uniform(o_con);
        u.o_icon = mw_uniform(o_icon); u.o_iefc = mw_uniform(o_iefc); u.o_icount = mw_uniform(o_icount); u.o_task = mw_uniform(o_task);
        u.lds_rows = mw_uniform(lds_rows); u.lds_w = mw_uniform(lds_w); u.lds_stride = mw_uniform(lds_stride); u.nsub = mw_uniform(nsub);
        u.lds_perm = mw_uniform(lds_perm); u.chain_lds = mw_uniform(chain_lds); u.tls_stride = mw_uniform(tls_stride);
        return u;
    }
    MW_HD CModel<T>& model() const { return *(CModel<T>*)(unsigned long long)m; }
    MW_HD CLayout& lay() const { return model().L; }
#if defined(MW_BOUNDS)
    MW_HD GRef<T> R(int i) const { return ((MW_GLOBAL T*)col)[chk(i, nreal_b, 1) * stride]; }
    MW_HD GRef<int> I(int i) const { return ((MW_GLOBAL int*)icol)[chk(i, nint_b, 2) * stride]; }
    MW_HD int S(int row, int f) const { return lds_perm + (int)chk(row, (unsigned)lds_rows, 3) * lds_w + f; }   // scratchpad slot of (row, field)
#else
    MW_HD GRef<T> R(int i) const { return ((MW_GLOBAL T*)col)[(unsigned)i * stride]; }
    MW_HD GRef<int> I(int i) const { return ((MW_GLOBAL int*)icol)[(unsigned)i * stride]; }
    MW_HD int S(int row, int f) const { return lds_perm + row * lds_w + f; }   // scratchpad slot of (row, field); fields >= SR_N: the row's Jacobian
#endif
};

// Cooperative sweeps.  When a batch is too small to fill the chip the runtime puts only `lpb` < 64 environments in a
// workgroup and the 64 / lpb threads with the same (thread % lpb) all belong to ONE environment: its "sub-lanes".
// They execute the lane program redundantly (same values, same stores) except inside the solver's row sweeps,
// which are split over the sub-lanes -- rows / constraint blocks k = sub, sub + nsub, ... -- and whose partial sums
// are combined with a butterfly (xor) exchange, so every sub-lane ends up with bit-identical totals.
//   MW_SUBS(e, sub) { ... slot MW_SLOT(sub) ... }   one parallel section; partial results go into arrays of MW_NSLOT
//   MW_SYNC()                                        makes scratchpad / column-store writes of a section visible
//   sub_sum(e, p) / sub_sum_n<N>(e, p)               butterfly totals
// The host harness has no lanes: it runs the sections for sub = 0 .. nsub-1 in a loop and adds the partials in the
// butterfly's order, which reproduces the device arithmetic exactly for any nsub.
#if defined(__HIP_DEVICE_COMPILE__)
#define MW_SUBS(e, sub) for (int sub = (e).sub, mw_once_ = 1; mw_once_; mw_once_ = 0)
#define MW_SLOT(sub) 0
constexpr int MW_NSLOT = 1;
// The threads that exchange data through memory at an MW_SYNC all belong to ONE wavefront (a workgroup is one wave).  A wave's
// memory instructions are issued and performed in order, so a later load of any lane observes an earlier store of any other lane
// of the same wave without waiting: wavefront-scope release / acquire need no instructions on gfx9 (LLVM AMDGPU memory model).
// What must be prevented is the COMPILER moving memory operations across the exchange point: a wavefront-scope fence (emits
// nothing) + the wave barrier (a scheduling barrier for convergent code).  __syncthreads() instead emits s_waitcnt vmcnt(0)
// lgkmcnt(0) + s_barrier: every exchange point drained all outstanding loads AND store acknowledgements (a full L2 round trip
// after any store), ~30 times per dynamics evaluation.  -DMW_SYNC_STRONG restores it (fault hunting).
#if defined(MW_SYNC_STRONG)
#define MW_SYNC() __syncthreads()
#else
#define MW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); } while (0)
#endif
template <typename T>
__device__ inline T sub_sum(const Env<T>& e, const T* p) {
    T v = p[0];
    for (int off = e.lds_stride; off < 64; off <<= 1) v += __shfl_xor(v, off);
    return v;
}
template <int N, typename T, typename U>
__device__ inline void sub_sum_n(const Env<T>& e, U (*p)[N]) {
    for (int off = e.lds_stride; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < N; k++) p[0][k] += __shfl_xor(p[0][k], off);
    }
}
// the value sub-lane 0 of the environment holds
template <typename T>
__device__ inline int sub_first(const Env<T>& e, int v) { return __shfl(v, e.thr % e.lds_stride); }
// true if the predicate holds in ANY live lane of the wave (wave-uniform result): loop / call conditions built from it keep
// every live lane of the wave inside the loop / the call, so that a non-inlined function is never entered under a partial
// EXEC mask (the lanes with nothing to do pass `active = false` and branch inside the callee) -- see collision()
__device__ inline bool mw_any(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0ull; }
// CANARY for the redundant execution model: the sub-lanes of an environment run the lane program redundantly and must hold
// bit-identical copies of every value that is not explicitly per-sub-lane.  True (in EVERY sub-lane of the environment) if
// some sub-lane's copy of v differs from sub-lane 0's; the stages raise ST_DIVERGED on it (DESIGN.md 5 "compiler sensitivity").
template <typename T>
__device__ inline bool sub_disagree(const Env<T>& e, int v) {
    int bad = v != sub_first(e, v);
    for (int off = e.lds_stride; off < 64; off <<= 1) bad |= __shfl_xor(bad, off);
    return bad != 0;
}
__device__ inline int canary_bits(float x) { return __builtin_bit_cast(int, x); }
__device__ inline int canary_bits(double x) { const unsigned long long u = __builtin_bit_cast(unsigned long long, x); return (int)(u ^ (u >> 32)); }
// exclusive prefix of one int per sub-lane (in sub-lane order) and the total
template <typename T>
__device__ inline int sub_scan(const Env<T>& e, const int* n, int* off) {
    const int mine = n[0], base = e.thr % e.lds_stride;
    int pre = 0, tot = 0;
    for (int s = 0; s < e.nsub; s++) {
        const int v = __shfl(mine, base + s * e.lds_stride);
        if (s < e.sub) pre += v;
        tot += v;
    }
    off[0] = pre;
    return tot;
}
#else
#define MW_SUBS(e, sub) for (int sub = 0; sub < (e).nsub; sub++)
#define MW_SLOT(sub) (sub)
constexpr int MW_NSLOT = 64;
#define MW_SYNC()
template <typename T, typename U>
inline U sub_sum(const Env<T>& e, const U* p) {
    U q[MW_NSLOT], r[MW_NSLOT];
    for (int s = 0; s < e.nsub; s++) q[s] = p[s];
    for (int off = 1; off < e.nsub; off <<= 1) {
        for (int s = 0; s < e.nsub; s++) r[s] = q[s] + q[s ^ off];
        for (int s = 0; s < e.nsub; s++) q[s] = r[s];
    }
    return q[0];
}
template <typename T>
inline int sub_first(const Env<T>&, int v) { return v; }
inline bool mw_any(bool pred) { return pred; }
template <typename T>
inline bool sub_disagree(const Env<T>&, int) { return false; }   // (the host harness computes every value once)
inline int canary_bits(float) { return 0; }
inline int canary_bits(double) { return 0; }
template <typename T>
inline int sub_scan(const Env<T>& e, const int* n, int* off) {
    int tot = 0;
    for (int s = 0; s < e.nsub; s++) { off[s] = tot; tot += n[s]; }
    return tot;
}
template <int N, typename T, typename U>
inline void sub_sum_n(const Env<T>& e, U (*p)[N]) {
    for (int k = 0; k < N; k++) {
        U col[MW_NSLOT];
        for (int s = 0; s < e.nsub; s++) col[s] = p[s][k];
        p[0][k] = sub_sum(e, col);
    }
}
#endif

// ----------------------------------------------------------------------------- small math
template <typename T> MW_HD T mw_sqrt(T x) { return sqrt(x); }
template <typename T> MW_HD T mw_abs(T x) { return x < 0 ? -x : x; }
// approximate single-precision reciprocal (v_rcp_f32, 1 ulp); only where the result is binned, never in the dynamics
MW_HD float mw_rcp_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
template <typename T> MW_HD T mw_min(T a, T b) { return a < b ? a : b; }
template <typename T> MW_HD T mw_max(T a, T b) { return a > b ? a : b; }
template <typename T> MW_HD T mw_clamp(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }

template <typename T> struct V3 { T x, y, z; };
template <typename T> MW_HD V3<T> v3(T x, T y, T z) { return V3<T>{x, y, z}; }
template <typename T> MW_HD V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> MW_HD V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> MW_HD V3<T> operator-(V3<T> a) { return {-a.x, -a.y, -a.z}; }
template <typename T> MW_HD V3<T> operator*(V3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T> MW_HD V3<T> operator*(T s, V3<T> a) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T> MW_HD T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> MW_HD V3<T> cross(V3<T> a, V3<T> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <typename T> MW_HD T norm(V3<T> a) { return mw_sqrt(dot(a, a)); }
template <typename T> MW_HD V3<T> normalized(V3<T> a, T* len = nullptr) {
    T n = norm(a);
    if (len) *len = n;
    if (n < T(1e-15)) { if (len) *len = 0; return {T(1), T(0), T(0)}; }
    T s = T(1) / n;
    return {a.x * s, a.y * s, a.z * s};
}
template <typename T> MW_HD T comp(V3<T> a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

template <typename T> struct Q4 { T w, x, y, z; };
template <typename T> MW_HD Q4<T> qmul(Q4<T> a, Q4<T> b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
template <typename T> MW_HD Q4<T> qnormalized(Q4<T> q) {
    T n = mw_sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    if (n < T(1e-15)) return {T(1), T(0), T(0), T(0)};
    T s = T(1) / n;
    return {q.w * s, q.x * s, q.y * s, q.z * s};
}
template <typename T> MW_HD Q4<T> qconj(Q4<T> q) { return {q.w, -q.x, -q.y, -q.z}; }

// 3x3 rotation, row-major
template <typename T> struct M3 { T m[9]; };
template <typename T> MW_HD M3<T> q2mat(Q4<T> q) {
    T w = q.w, x = q.x, y = q.y, z = q.z;
    M3<T> r;
    r.m[0] = w * w + x * x - y * y - z * z; r.m[1] = 2 * (x * y - w * z); r.m[2] = 2 * (x * z + w * y);
    r.m[3] = 2 * (x * y + w * z); r.m[4] = w * w - x * x + y * y - z * z; r.m[5] = 2 * (y * z - w * x);
    r.m[6] = 2 * (x * z - w * y); r.m[7] = 2 * (y * z + w * x); r.m[8] = w * w - x * x - y * y + z * z;
    return r;
}
template <typename T> MW_HD V3<T> operator*(const M3<T>& a, V3<T> v) {
    return {a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
            a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
template <typename T> MW_HD V3<T> mulT(const M3<T>& a, V3<T> v) {
    return {a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
            a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z};
}
template <typename T> MW_HD V3<T> col(const M3<T>& a, int k) { return {a.m[k], a.m[3 + k], a.m[6 + k]}; }

// column-store load/store helpers
template <typename T> MW_HD V3<T> ld3(const Env<T> e, int i) { return {e.R(i), e.R(i + 1), e.R(i + 2)}; }
template <typename T> MW_HD void st3(const Env<T> e, int i, V3<T> v) { e.R(i) = v.x; e.R(i + 1) = v.y; e.R(i + 2) = v.z; }
template <typename T> MW_HD Q4<T> ld4(const Env<T> e, int i) { return {e.R(i), e.R(i + 1), e.R(i + 2), e.R(i + 3)}; }
template <typename T> MW_HD void st4(const Env<T> e, int i, Q4<T> q) { e.R(i) = q.w; e.R(i + 1) = q.x; e.R(i + 2) = q.y; e.R(i + 3) = q.z; }
template <typename T> MW_HD M3<T> ld9(const Env<T> e, int i) {
    M3<T> r;
    for (int k = 0; k < 9; k++) r.m[k] = e.R(i + k);
    return r;
}
template <typename T> MW_HD void st9(const Env<T> e, int i, const M3<T>& a) {
    for (int k = 0; k < 9; k++) e.R(i + k) = a.m[k];
}
// 3-/4-vector from any pointer (model table in constant memory, local array, ...)
template <typename T> struct elem_of;
template <typename T> struct elem_of<const T*> { typedef T type; };
template <typename T> struct elem_of<T*> { typedef T type; };
#if defined(__HIP_DEVICE_COMPILE__)
template <typename T> struct elem_of<const MW_CONST T*> { typedef T type; };
#endif
template <typename P> MW_HD V3<typename elem_of<P>::type> mv3(P p) { return {p[0], p[1], p[2]}; }
template <typename P> MW_HD Q4<typename elem_of<P>::type> mq4(P p) { return {p[0], p[1], p[2], p[3]}; }

}  // namespace mw
