// mw_tasks.hpp -- per-lane Meta-World task layer: observation, reward/success, reset.
//
// Restates, for one environment per lane, the Python hot path of the reference:
//   SawyerXYZEnv.step            metaworld/sawyer_xyz_env.py:579-642
//   set_xyz_action               :320-336
//   _get_curr_obs_combined_no_goal / _get_obs   :475-527
//   reset / _reset_hand          :664-695
//   reward_utils.tolerance / hamacher_product   metaworld/utils/reward_utils.py:97-144, :222-244
//   per-task evaluate_state / compute_reward / reset_model   metaworld/envs/sawyer_*_v3.py (v2 rewards)
// Task ids are the MT50 one-hot ids (metaworld/env_dict.py:217-270).
#pragma once
#include "mw_collide.hpp"
#include "mw_common.hpp"
#include "mw_phys.hpp"

namespace mw {

// ---- per-env task block (offsets inside Layout::task) ----
enum {
    TK_TASK = 0,      // TaskDesc index
    TK_GOAL = 1,      // goal (rand_vec) index inside the task's goal table
    TK_PATHLEN = 2,   // curr_path_length
    TK_ELAPSED = 3,   // TimeLimit._elapsed_steps
    TK_EPRET = 4,     // RecordEpisodeStatistics return
    TK_EPLEN = 5,     //   "" length
    TK_TARGET = 6,    // _target_pos[3]
    TK_OBJINIT = 9,   // obj_init_pos[3]
    TK_INITTCP = 12,  // init_tcp[3]
    TK_PREVOBS = 15,  // _prev_obs[18]
    TK_RANDVEC = 33,  // rand_vec[6]
    TK_EXTRA = 39,    // task-specific scalars [16]
    TK_SUCCESS = 55,  // last success flag
    TK_PERSIST0 = 56, // 3 reals that survive resets (basketball's drifting goal-site local position)
    TK_EPRET_LO = 59, // low part of the episode return (fp32 contexts sum it as a (hi, lo) float pair = in double)
    TK_END = 60
};
static_assert(TK_END <= TASK_NREAL, "task block too small");

// probe roles common to every task
enum { P_HAND = 0, P_RCLAW, P_LCLAW, P_RPAD, P_LPAD, P_RTCP, P_LTCP, P_OBJ0, P_OBJ1, P_OBJ2, P_OBJ3, P_X0, P_X1, P_X2, P_X3, P_X4, P_COUNT };
// quaternion conventions of obs[7:11] / obs[14:18] (SURVEY.md Appendix A legend)
enum { QUAT_SCIPY = 0, QUAT_MUJOCO = 1, QUAT_ZERO = 2, QUAT_IDENT = 3, QUAT_NONE = 4 };

template <typename T>
struct TaskDesc {
    int kind;              // MT50 id: selects reward / reset code
    int probe[P_COUNT];    // resolved probe ids (-1 = unused)
    int nobj;              // number of (pos, quat) pairs in the observation (1 or 2)
    int quat_mode[2];
    int qadr[4], dadr[4];  // task-specific joint addresses (qpos / dof)
    int geom[4];           // task-specific geom ids
    int reloc[2];          // relocatable body slots
    int partially_observable, max_path_length;
    T hand_init[3], mocap_low[3], mocap_high[3], goal_low[3], goal_high[3];
    T obj_off[2][3];       // constant offsets added to the observed object positions
    T c[16];               // task constants
};

// ------------------------------------------------------------------ reward_utils
template <typename T>
MW_HD T tolerance_lt(T x, T lo, T hi, T margin) {   // sigmoid="long_tail"
    if (lo <= x && x <= hi) return 1;
    if (margin == 0) return 0;
    const T d = (x < lo ? lo - x : x - hi) / margin;
    const T s = d * T(3);   // sqrt(1/0.1 - 1)
    return 1 / (s * s + 1);
}
template <typename T>
MW_HD T tolerance_gauss(T x, T lo, T hi, T margin) {   // sigmoid="gaussian"
    if (lo <= x && x <= hi) return 1;
    if (margin == 0) return 0;
    const T d = (x < lo ? lo - x : x - hi) / margin;
    const T s = T(2.1459660262893472);  // sqrt(-2 ln 0.1)
    return T(exp(double(-0.5) * double(d * s) * double(d * s)));
}
template <typename T>
MW_HD T hamacher(T a, T b) {
    const T den = a + b - a * b;
    return den > 0 ? a * b / den : T(0);
}

// scipy Rotation.from_matrix(m).as_quat() -> (x, y, z, w), not sign-canonicalised
template <typename T>
MW_HD void scipy_quat(const M3<T>& R, T* q) {
    const T m00 = R.m[0], m11 = R.m[4], m22 = R.m[8], tr = m00 + m11 + m22;
    const T dec[4] = {m00, m11, m22, tr};
    int ch = 0;
    for (int k = 1; k < 4; k++) if (dec[k] > dec[ch]) ch = k;
    if (ch != 3) {
        const int i = ch, j = (i + 1) % 3, k = (j + 1) % 3;
        q[i] = 1 - tr + 2 * R.m[3 * i + i];
        q[j] = R.m[3 * j + i] + R.m[3 * i + j];
        q[k] = R.m[3 * k + i] + R.m[3 * i + k];
        q[3] = R.m[3 * k + j] - R.m[3 * j + k];
    } else {
        q[0] = R.m[7] - R.m[5]; q[1] = R.m[2] - R.m[6]; q[2] = R.m[3] - R.m[1]; q[3] = 1 + tr;
    }
    const T n = mw_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int c = 0; c < 4; c++) q[c] /= n;
}

template <typename T> MW_HD GRef<T> TK(const Env<T> e, int k) { return e.R(e.o_task + k); }
template <typename T> MW_HD V3<T> tk3(const Env<T> e, int k) { return ld3(e, e.o_task + k); }
template <typename T> MW_HD void set_tk3(const Env<T> e, int k, V3<T> v) { st3(e, e.o_task + k, v); }

template <typename T>
MW_HD V3<T> tcp_center(const Env<T> e, const TaskDesc<T>& td) {
    return (probe_pos(e, td.probe[P_RTCP]) + probe_pos(e, td.probe[P_LTCP])) * T(0.5);
}

// _get_curr_obs_combined_no_goal (sawyer_xyz_env.py:475-511)
template <typename T>
MW_HD void curr_obs(const Env<T> e, const TaskDesc<T>& td, T* o18) {
    const V3<T> hand = probe_pos(e, td.probe[P_HAND]);
    o18[0] = hand.x; o18[1] = hand.y; o18[2] = hand.z;
    const T gd = norm(probe_pos(e, td.probe[P_RCLAW]) - probe_pos(e, td.probe[P_LCLAW]));
    o18[3] = mw_clamp(gd / T(0.1), T(0), T(1));
    for (int k = 4; k < 18; k++) o18[k] = 0;
    for (int i = 0; i < td.nobj; i++) {
        T* o = o18 + 4 + 7 * i;
        V3<T> p = probe_pos(e, td.probe[P_OBJ0 + 2 * i]);
        if (td.kind == 11) {   // dial-turn: dial centre + 0.05 * [sin q, -cos q, 0]  (envs/sawyer_dial_turn_v3.py:87-98)
            const T q = e.R(e.lay().qpos + td.qadr[0]);
            p = p + v3<T>(T(0.05) * sin(q), T(-0.05) * cos(q), 0);
        }
        o[0] = p.x + td.obj_off[i][0]; o[1] = p.y + td.obj_off[i][1]; o[2] = p.z + td.obj_off[i][2];
        const int qm = td.quat_mode[i];
        if (qm == QUAT_SCIPY) scipy_quat(q2mat(probe_quat(e, td.probe[P_OBJ1 + 2 * i])), o + 3);
        else if (qm == QUAT_MUJOCO) { const Q4<T> q = probe_quat(e, td.probe[P_OBJ1 + 2 * i]); o[3] = q.w; o[4] = q.x; o[5] = q.y; o[6] = q.z; }
        else if (qm == QUAT_IDENT) { o[3] = 1; o[4] = o[5] = o[6] = 0; }
        else { o[3] = o[4] = o[5] = o[6] = 0; }
    }
}

// _get_obs (:513-527): [curr18, prev18, goal3]; updates _prev_obs
template <typename T>
MW_HD void get_obs(const Env<T> e, const TaskDesc<T>& td, T* obs39) {
    curr_obs(e, td, obs39);
    for (int k = 0; k < 18; k++) { obs39[18 + k] = TK(e, TK_PREVOBS + k); TK(e, TK_PREVOBS + k) = obs39[k]; }
    for (int k = 0; k < 3; k++) obs39[36 + k] = td.partially_observable ? T(0) : TK(e, TK_TARGET + k);
}

// observation-space clip (:537-577, :623-628)
template <typename T>
MW_HD void clip_obs(const TaskDesc<T>& td, T* o) {
    const T hlo[3] = {T(-0.525), T(0.348), T(-0.0525)}, hhi[3] = {T(0.525), T(1.025), T(0.7)};
    for (int f = 0; f < 2; f++) {
        for (int k = 0; k < 3; k++) o[18 * f + k] = mw_clamp(o[18 * f + k], hlo[k], hhi[k]);
        o[18 * f + 3] = mw_clamp(o[18 * f + 3], T(-1), T(1));
    }
    for (int k = 0; k < 3; k++)
        o[36 + k] = td.partially_observable ? T(0) : mw_clamp(o[36 + k], td.goal_low[k], td.goal_high[k]);
}

// _reset_hand (:684-695): 50 x { mocap <- hand_init_pos ; do_simulation([-1, 1], 5) }; init_tcp from the lagged FK
template <typename T>
MW_HD void reset_hand(const Env<T> e, const TaskDesc<T>& td, int steps = 50) {
    for (int s = 0; s < steps; s++) {
        st3(e, e.lay().mocap, v3(td.hand_init[0], td.hand_init[1], td.hand_init[2]));
        e.R(e.lay().ctrl) = -1; e.R(e.lay().ctrl + 1) = 1;
        for (int k = 0; k < 5; k++) substep(e);
    }
    set_tk3(e, TK_INITTCP, tcp_center(e, td));
}

// default _set_obj_xyz (:351-361): qpos[9:12] <- pos ; qvel[9:15] <- 0 ; set_state -> mj_forward
template <typename T>
MW_HD void set_obj_xyz(const Env<T> e, V3<T> p) {
    st3(e, e.lay().qpos + 9, p);
    for (int k = 9; k < 15; k++) e.R(e.lay().qvel + k) = 0;
    forward(e);
}

struct Info { float near_object, grasp_success, grasp_reward, in_place_reward, obj_to_target, unscaled_reward; };

MW_HD Info make_info(double near_object, double grasp_success, double grasp_reward, double in_place, double obj_to_target, double unscaled) {
    return Info{float(near_object), float(grasp_success), float(grasp_reward), float(in_place), float(obj_to_target), float(unscaled)};
}

// ---- per-task slots ----
// TaskDesc::geom : 0 main object geom (touching_main_object), 1 leftpad_geom, 2 rightpad_geom
// TaskDesc::c    : 0-2 init_config obj_init_pos, 3-5 class `goal`, 6.. task constants, 15 one-hot id
enum { G_OBJ = 0, G_LPAD = 1, G_RPAD = 2 };

// touching_object (sawyer_xyz_env.py:401-440): both pads have positive summed normal force against the geom.
// Faithful to the reference's `efc_force[contact.efc_address]` including efc_address == -1 reading the LAST row.
// the tasks (MT50 one-hot ids) whose evaluate_state calls touching_object: push 40, pick-place 30, push-back 42, push-wall 41,
// pick-place-wall 28, sweep 47 / sweep-into 46 / soccer 37 / hand-insert 17 (sweepfam_eval), coffee-pull 9 / coffee-push 10,
// shelf-place 45, stick-push 38 / stick-pull 39
MW_HD bool task_touches(int kind) {
    const unsigned long long m = (1ull << 40) | (1ull << 30) | (1ull << 42) | (1ull << 41) | (1ull << 28) | (1ull << 47) | (1ull << 46) |
                                 (1ull << 37) | (1ull << 17) | (1ull << 9) | (1ull << 10) | (1ull << 45) | (1ull << 38) | (1ull << 39);
    return kind >= 0 && kind < 64 && ((m >> kind) & 1ull);
}
template <typename T>
MW_HD bool touching_object(const Env<T> e, const TaskDesc<T>& td, int objgeom) {
    // env_step stopped its final mj_forward after the kinematics and ran the second half for the whole wave iff some lane's task
    // is listed in task_touches().  Arriving here without it means the list is missing this task: the result is still computed,
    // but forward_dynamics (whose narrow phase is a non-inlined 256-VGPR callee) is then entered under a partial EXEC mask --
    // the condition DESIGN.md 5 "register hazard" removes -- so the execution-model flag is raised and every test fails on it.
    if (!e.I(e.lay().icount + IC_DYN_VALID)) { e.I(e.lay().icount + 3) |= ST_DIVERGED; forward_dynamics(e); }
    const int ncon = e.I(e.lay().icount), nefc = e.I(e.lay().icount + 1);
    T fl = 0, fr = 0;
    for (int c = 0; c < ncon; c++) {
        const int g1 = ICON(e, c, 0), g2 = ICON(e, c, 1);
        if (g1 != objgeom && g2 != objgeom) continue;
        const bool l = g1 == td.geom[G_LPAD] || g2 == td.geom[G_LPAD], r = g1 == td.geom[G_RPAD] || g2 == td.geom[G_RPAD];
        if (!l && !r) continue;
        int adr = ICON(e, c, 3);
        if (adr < 0) adr = nefc - 1;
        const T f = adr >= 0 ? EX(e, adr, 5) : T(0);
        if (l) fl += f;
        if (r) fr += f;
    }
    return fl > 0 && fr > 0;
}

// base SawyerXYZEnv._gripper_caging_reward (sawyer_xyz_env.py:721-858); `ref` plays obj_init_pos (stick tasks pass stick_init_pos)
template <typename T>
MW_HD T caging_base(const Env<T> e, const TaskDesc<T>& td, const T* act, V3<T> obj, V3<T> ref, T obj_radius, T pad_success_thresh,
                    T object_reach_radius, T xz_thresh, T desired_effort, bool high_density, bool medium_density) {
    const V3<T> lp = probe_pos(e, td.probe[P_LPAD]), rp = probe_pos(e, td.probe[P_RPAD]);
    const T pad_y[2] = {lp.y, rp.y};
    T cag[2];
    for (int i = 0; i < 2; i++) {
        const T x = mw_abs(pad_y[i] - obj.y), m = mw_abs(mw_abs(pad_y[i] - ref.y) - pad_success_thresh);
        cag[i] = tolerance_lt(x, obj_radius, pad_success_thresh, m);
    }
    const T caging_y = hamacher(cag[0], cag[1]);
    const V3<T> tcp = tcp_center(e, td), itcp = tk3(e, TK_INITTCP);
    const T mxz = mw_sqrt((ref.x - itcp.x) * (ref.x - itcp.x) + (ref.z - itcp.z) * (ref.z - itcp.z)) - xz_thresh;
    const T dxz = mw_sqrt((tcp.x - obj.x) * (tcp.x - obj.x) + (tcp.z - obj.z) * (tcp.z - obj.z));
    const T caging_xz = tolerance_lt(dxz, T(0), xz_thresh, mxz);
    const T closed = mw_min(mw_max(T(0), act[3]), desired_effort) / desired_effort;
    const T caging = hamacher(caging_y, caging_xz);
    const T gripping = caging > T(0.97) ? closed : T(0);
    T cg = hamacher(caging, gripping);
    if (high_density) cg = (cg + caging) / 2;
    if (medium_density) {
        const T tcp_to_obj = norm(obj - tcp), init = norm(ref - itcp);
        const T reach = tolerance_lt(tcp_to_obj, T(0), object_reach_radius, mw_abs(init - object_reach_radius));
        cg = (cg + reach) / 2;
    }
    return cg;
}

// the per-task overrides (pick-place :180-248; push-back/soccer/sweep/sweep-into): the "init" pads are live views,
// i.e. the CURRENT pad positions.  mode 0 = pick-place, 1 = y-gripping family.
template <typename T>
MW_HD T caging_override(const Env<T> e, const TaskDesc<T>& td, const T* act, V3<T> obj, int mode, T obj_radius, T grip_margin, T xz_margin) {
    const T pad_margin = T(0.05);
    const V3<T> lp = probe_pos(e, td.probe[P_LPAD]), rp = probe_pos(e, td.probe[P_RPAD]);
    const T dl = lp.y - obj.y, dr = obj.y - rp.y;
    const T mr = mw_abs(mw_abs(obj.y - rp.y) - pad_margin), ml = mw_abs(mw_abs(obj.y - lp.y) - pad_margin);
    const T rc = tolerance_lt(dr, obj_radius, pad_margin, mr), lc = tolerance_lt(dl, obj_radius, pad_margin, ml);
    const T y_caging = hamacher(lc, rc);
    const V3<T> tcp = tcp_center(e, td), oi = tk3(e, TK_OBJINIT), itcp = tk3(e, TK_INITTCP);
    const T dxz = mw_sqrt((tcp.x - obj.x) * (tcp.x - obj.x) + (tcp.z - obj.z) * (tcp.z - obj.z));
    const T mxz = mw_sqrt((oi.x - itcp.x) * (oi.x - itcp.x) + (oi.z - itcp.z) * (oi.z - itcp.z)) - xz_margin;
    const T xz_caging = tolerance_lt(dxz, T(0), xz_margin, mxz);
    const T caging = hamacher(y_caging, xz_caging);
    if (mode == 0) {
        const T closed = mw_min(mw_max(T(0), act[3]), T(1));
        const T gripping = caging > T(0.97) ? closed : T(0);
        return (hamacher(caging, gripping) + caging) / 2;
    }
    const T rg = tolerance_lt(dr, obj_radius, grip_margin, mr), lg = tolerance_lt(dl, obj_radius, grip_margin, ml);
    const T y_gripping = hamacher(rg, lg);
    return (caging + (caging > T(0.95) ? y_gripping : T(0))) / 2;
}

template <typename T> MW_HD V3<T> obs3(const T* o, int k) { return V3<T>{o[k], o[k + 1], o[k + 2]}; }
template <typename T> MW_HD V3<T> c3(const TaskDesc<T>& td, int k) { return V3<T>{td.c[k], td.c[k + 1], td.c[k + 2]}; }
template <typename T> MW_HD V3<T> scale3(V3<T> a, T x, T y, T z) { return V3<T>{a.x * x, a.y * y, a.z * z}; }

// =========================================================================== per-task code
// Each task: <name>_reset restates reset_model(), <name>_eval restates evaluate_state()+compute_reward() (v2 branch).
struct Out { double reward, success; Info info; };

// ---- reach-v3 (43) / reach-wall-v3 (44): envs/sawyer_reach_v3.py:119-161, sawyer_reach_wall_v3.py ----
template <typename T>
MW_HD void reach_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    set_tk3(e, TK_TARGET, tk3(e, TK_RANDVEC + 3));
    set_tk3(e, TK_OBJINIT, tk3(e, TK_RANDVEC));
    set_obj_xyz(e, tk3(e, TK_RANDVEC));
}
template <typename T>
MW_HD Out reach_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> tcp = tcp_center(e, td), target = tk3(e, TK_TARGET);
    const T d = norm(tcp - target);
    const T in_place = tolerance_lt(d, T(0), T(0.05), norm(v3(td.hand_init[0], td.hand_init[1], td.hand_init[2]) - target));
    if (td.kind == 44)   // reach-wall: near_object / grasp_success / grasp_reward are constant 0
        return Out{double(10 * in_place), double(d <= T(0.05)), make_info(0.0, 0.0, 0.0, in_place, d, 10 * in_place)};
    return Out{double(10 * in_place), double(d <= T(0.05)), make_info(d, 1.0, d, in_place, d, 10 * in_place)};
}

// ---- the push / pick-place family: free object at qpos[9:16], goal from rand_vec[3:6] ----
// reset flavours (all after _reset_hand): where obj z and target z come from differs per task.
template <typename T>
MW_HD void pushpick_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC), rv1 = tk3(e, TK_RANDVEC + 3);
    V3<T> oi = rv0, tg = rv1;
    const T body_z = probe_pos(e, td.probe[P_OBJ0]).z;          // get_body_com("obj")[-1] / geom xpos[-1] after settling
    switch (td.kind) {
    case 40:  // push: z of both = fix_extreme_obj_pos(...)[2] = body z
        oi.z = body_z; tg.z = body_z; break;
    case 41:  // push-wall: adjust_initObjPos z = geom("objGeom").xpos z
    case 42:  // push-back
        oi.z = probe_pos(e, td.probe[P_OBJ1]).z; tg.z = oi.z; break;
    case 30:  // pick-place: init_tcp / pads re-captured (views), obj & target straight from rand_vec
        set_tk3(e, TK_INITTCP, tcp_center(e, td)); break;
    case 28:  // pick-place-wall
    default: break;
    }
    set_tk3(e, TK_TARGET, tg);
    set_tk3(e, TK_OBJINIT, oi);
    set_obj_xyz(e, oi);
}
template <typename T>
MW_HD Out push_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {   // push-v3 (40)
    const V3<T> obj = obs3(obs, 4), target = tk3(e, TK_TARGET), oi = tk3(e, TK_OBJINIT);
    const T opened = obs[3], tcp_to_obj = norm(obj - tcp_center(e, td)), t2o = norm(obj - target);
    const T in_place = tolerance_lt(t2o, T(0), T(0.05), norm(oi - target));
    const T grasped = caging_base(e, td, act, obj, oi, T(0.015), T(0.05), T(0.01), T(0.005), T(1), true, false);
    T reward = 2 * grasped;
    if (tcp_to_obj < T(0.02) && opened > 0) reward += 1 + reward + 5 * in_place;
    if (t2o < T(0.05)) reward = 10;
    const bool gs = touching_object(e, td, td.geom[G_OBJ]) && opened > 0 && obj.z - T(0.02) > oi.z;
    return Out{double(reward), double(t2o <= T(0.05)), make_info(tcp_to_obj <= T(0.03), gs, grasped, in_place, t2o, reward)};
}
template <typename T>
MW_HD Out pick_place_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {   // pick-place-v3 (30)
    const V3<T> obj = obs3(obs, 4), target = tk3(e, TK_TARGET), oi = tk3(e, TK_OBJINIT);
    const T opened = obs[3], o2t = norm(obj - target), tcp_to_obj = norm(obj - tcp_center(e, td));
    const T in_place = tolerance_lt(o2t, T(0), T(0.05), norm(oi - target));
    const T grasped = caging_override(e, td, act, obj, 0, T(0.015), T(0), T(0.005));
    T reward = hamacher(grasped, in_place);
    if (tcp_to_obj < T(0.02) && opened > 0 && obj.z - T(0.01) > oi.z) reward += 1 + 5 * in_place;
    if (o2t < T(0.05)) reward = 10;
    const bool gs = touching_object(e, td, td.geom[G_OBJ]) && opened > 0 && obj.z - T(0.02) > oi.z;
    return Out{double(reward), double(o2t <= T(0.07)), make_info(tcp_to_obj <= T(0.03), gs, grasped, in_place, o2t, reward)};
}
template <typename T>
MW_HD Out push_back_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {   // push-back-v3 (42)
    const V3<T> obj = obs3(obs, 4), target = tk3(e, TK_TARGET), oi = tk3(e, TK_OBJINIT);
    const T opened = obs[3], tcp_to_obj = norm(obj - tcp_center(e, td)), t2o = norm(obj - target), t2oi = norm(oi - target);
    const T in_place = tolerance_lt(t2o, T(0), T(0.05), t2oi);
    const T grasped = caging_override(e, td, act, obj, 1, T(0.007), T(0.007 + 0.003), T(0.01));
    T reward = hamacher(grasped, in_place);
    if (tcp_to_obj < T(0.01) && opened > 0 && opened < T(0.55) && t2oi - t2o > T(0.01)) reward += 1 + 5 * in_place;
    if (t2o < T(0.05)) reward = 10;
    const bool gs = touching_object(e, td, td.geom[G_OBJ]) && opened > 0 && obj.z - T(0.02) > oi.z;
    return Out{double(reward), double(t2o <= T(0.07)), make_info(tcp_to_obj <= T(0.03), gs, grasped, in_place, t2o, reward)};
}
template <typename T>
MW_HD Out wall_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {   // push-wall (41) / pick-place-wall (28)
    const bool pick = td.kind == 28;
    const V3<T> obj = obs3(obs, 4), target = tk3(e, TK_TARGET), oi = tk3(e, TK_OBJINIT);
    const T opened = obs[3], tcp_to_obj = norm(obj - tcp_center(e, td));
    const V3<T> mid = pick ? V3<T>{target.x, T(0.77), T(0.25)} : V3<T>{T(-0.05), T(0.77), obj.z};
    const T sx = pick ? T(1) : T(3), sz = pick ? T(3) : T(1);
    const T o2m = norm(scale3(obj - mid, sx, T(1), sz)), o2mi = norm(scale3(oi - mid, sx, T(1), sz));
    const T o2t = norm(obj - target), o2ti = norm(oi - target);
    const T p1 = tolerance_lt(o2m, T(0), T(0.05), o2mi), p2 = tolerance_lt(o2t, T(0), T(0.05), o2ti);
    const T grasped = caging_base(e, td, act, obj, oi, T(0.015), T(0.05), T(0.01), T(0.005), T(1), !pick, false);
    T reward;
    if (pick) {
        const T ipg = hamacher(grasped, p1);
        reward = ipg;
        if (tcp_to_obj < T(0.02) && opened > 0 && obj.z - T(0.015) > oi.z) {
            reward = ipg + 1 + 4 * p1;
            if (obj.y > T(0.75)) reward = ipg + 1 + 4 + 3 * p2;
        }
    } else {
        reward = 2 * grasped;
        if (tcp_to_obj < T(0.02) && opened > 0) {
            reward = 2 * grasped + 1 + 4 * p1;
            if (obj.y > T(0.75)) reward = 2 * grasped + 1 + 4 + 3 * p2;
        }
    }
    if (o2t < T(0.05)) reward = 10;
    const bool gs = touching_object(e, td, td.geom[G_OBJ]) && opened > 0 && obj.z - T(0.02) > oi.z;
    return Out{double(reward), double(o2t <= T(0.07)), make_info(tcp_to_obj <= T(0.03), gs, grasped, p2, o2t, reward)};
}

// ---- sweep-v3 (47), sweep-into-v3 (46), soccer-v3 (37), hand-insert-v3 (17) ----
template <typename T>
MW_HD void sweepfam_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC), rv1 = tk3(e, TK_RANDVEC + 3);
    V3<T> oi, tg;
    if (td.kind == 47) {            // sweep: target = goal with y <- rand_vec.y ; obj z = init_config z
        oi = V3<T>{rv0.x, rv0.y, td.c[2]}; tg = V3<T>{td.c[3], rv0.y, td.c[5]};
    } else if (td.kind == 46) {     // sweep-into: obj z = settled body z ; target = class goal
        oi = V3<T>{rv0.x, rv0.y, probe_pos(e, td.probe[P_OBJ0]).z}; tg = c3(td, 3);
    } else if (td.kind == 37) {     // soccer: goal_whole body relocated to the target
        oi = V3<T>{rv0.x, rv0.y, td.c[2]}; tg = rv1;
        st3(e, e.lay().reloc + 3 * td.reloc[0], tg);
    } else {                        // hand-insert
        oi = V3<T>{rv0.x, rv0.y, td.c[2]}; tg = rv1;
    }
    set_tk3(e, TK_TARGET, tg);
    set_tk3(e, TK_OBJINIT, oi);
    set_obj_xyz(e, oi);
}
template <typename T>
MW_HD Out sweepfam_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> obj = obs3(obs, 4), oi = tk3(e, TK_OBJINIT);
    V3<T> target = tk3(e, TK_TARGET);
    const T opened = obs[3], tcp_to_obj = norm(obj - tcp_center(e, td));
    const bool touch = touching_object(e, td, td.geom[G_OBJ]);
    if (td.kind == 47 || td.kind == 46) {
        if (td.kind == 46) target.z = obj.z;
        const T o2t = norm(obj - target), in_place = tolerance_lt(o2t, T(0), T(0.05), norm(oi - target));
        const T grasped = td.kind == 47 ? caging_override(e, td, act, obj, 1, T(0.02), T(0.02 + 0.01), T(0.005))
                                        : caging_override(e, td, act, obj, 1, T(0.02), T(0.02 + 0.005), T(0.01));
        T reward = 2 * grasped + 6 * hamacher(grasped, in_place);
        if (o2t < T(0.05)) reward = 10;
        return Out{double(reward), double(o2t <= T(0.05)), make_info(tcp_to_obj <= T(0.03), touch && opened > 0, grasped, in_place, o2t, reward)};
    }
    if (td.kind == 37) {
        const T t2o = norm(scale3(obj - target, T(3), T(1), T(1))), t2oi = norm(scale3(obj - oi, T(3), T(1), T(1)));
        T in_place = tolerance_lt(t2o, T(0), T(0.07), t2oi);
        const T goal_line = target.y - T(0.1);
        if (obj.y > goal_line && mw_abs(obj.x - target.x) > T(0.10))
            in_place = mw_clamp(in_place - 2 * ((obj.y - goal_line) / (1 - goal_line)), T(0), T(1));
        const T grasped = caging_override(e, td, act, obj, 1, T(0.013), T(0.013 + 0.01), T(0.005));
        T reward = 3 * grasped + T(6.5) * in_place;
        if (t2o < T(0.07)) reward = 10;
        const T d = norm(obj - target);
        const bool gs = touch && opened > 0 && obj.z - T(0.02) > oi.z;
        return Out{double(reward), double(d <= T(0.07)), make_info(tcp_to_obj <= T(0.03), gs, grasped, in_place, d, reward)};
    }
    // hand-insert
    const T t2o = norm(obj - target), in_place = tolerance_lt(t2o, T(0), T(0.05), norm(oi - target));
    const T grasped = caging_base(e, td, act, obj, oi, T(0.015), T(0.05), T(0.01), T(0.005), T(1), true, false);
    T reward = hamacher(grasped, in_place);
    if (tcp_to_obj < T(0.02) && opened > 0) reward += 1 + 7 * in_place;
    if (t2o < T(0.05)) reward = 10;
    const bool gs = touch && opened > 0 && obj.z - T(0.02) > oi.z;
    return Out{double(reward), double(t2o <= T(0.05)), make_info(tcp_to_obj <= T(0.03), gs, grasped, in_place, t2o, reward)};
}

// ---- bin-picking-v3 (2): envs/sawyer_bin_picking_v3.py ; TK_EXTRA[0] = _target_to_obj_init latch (-1 = None) ----
template <typename T>
MW_HD void bin_picking_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC);
    const V3<T> oi{rv0.x, rv0.y, probe_pos(e, td.probe[P_OBJ0]).z};
    set_tk3(e, TK_OBJINIT, oi);
    set_obj_xyz(e, oi);
    set_tk3(e, TK_TARGET, probe_pos(e, td.probe[P_X0]));   // get_body_com("bin_goal")
    TK(e, TK_EXTRA) = -1;
}
template <typename T>
MW_HD Out bin_picking_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> hand = obs3(obs, 0), obj = obs3(obs, 4), target = tk3(e, TK_TARGET), oi = tk3(e, TK_OBJINIT);
    const T t2o = norm(obj - target);
    if (TK(e, TK_EXTRA) < 0) TK(e, TK_EXTRA) = t2o;
    const T in_place = tolerance_lt(t2o, T(0), T(0.05), TK(e, TK_EXTRA));
    const T thr = T(0.03);
    const T r0 = mw_sqrt((hand.x - oi.x) * (hand.x - oi.x) + (hand.y - oi.y) * (hand.y - oi.y));
    const T r1 = mw_sqrt((hand.x - target.x) * (hand.x - target.x) + (hand.y - target.y) * (hand.y - target.y));
    const T f0 = r0 > thr ? T(0.02) * T(log(double(r0 - thr))) + T(0.2) : T(0), f1 = r1 > thr ? T(0.02) * T(log(double(r1 - thr))) + T(0.2) : T(0);
    const T floor_ = mw_min(f0, f1);
    const T above = hand.z >= floor_ ? T(1) : tolerance_lt(mw_max(floor_ - hand.z, T(0)), T(0), T(0.01), T(0.05));
    const T grasped = caging_base(e, td, act, obj, oi, T(0.015), T(0.05), T(0.01), T(0.01), T(0.7), true, false);
    T reward = hamacher(grasped, in_place);
    const bool near = norm(obj - hand) < T(0.04), pinched = obs[3] < T(0.43), lifted = obj.z - T(0.02) > oi.z;
    const bool gs = near && lifted && !pinched;
    if (gs) reward += 1 + 5 * hamacher(above, in_place);
    if (t2o < T(0.05)) reward = 10;
    return Out{double(reward), double(t2o <= T(0.05)), make_info(near, gs, grasped, in_place, t2o, reward)};
}

// ---- generic helpers for fixture tasks ----
// per-env model write: model.body(X).pos = v  (reloc slot k)
template <typename T> MW_HD void set_reloc(const Env<T> e, const TaskDesc<T>& td, int k, V3<T> v) { st3(e, e.lay().reloc + 3 * td.reloc[k], v); }
// joint-level _set_obj_xyz variants: qpos[adr] <- q ; qvel[dadr] <- 0 ; set_state -> mj_forward
template <typename T> MW_HD void set_joint(const Env<T> e, int qadr, int dadr, T q) {
    e.R(e.lay().qpos + qadr) = q;
    if (dadr >= 0) e.R(e.lay().qvel + dadr) = 0;
    forward(e);
}

// ---- button-press-topdown (4), -topdown-wall (5), button-press (6), -wall (7) ----
// probes: P_X0 = site hole, P_X1 = site buttonStart ; reloc0 = body box ; TK_EXTRA[0] = _obj_to_target_init
template <typename T>
MW_HD void button_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> oi = tk3(e, TK_RANDVEC);
    set_tk3(e, TK_OBJINIT, oi);
    set_reloc(e, td, 0, oi);
    if (td.kind == 4 || td.kind == 5) forward(e);       // mujoco.mj_forward
    else set_joint(e, 9, 9, T(0));                      // _set_obj_xyz(0): qpos[9], qvel[9]
    const V3<T> target = probe_pos(e, td.probe[P_X0]), bs = probe_pos(e, td.probe[P_X1]);
    set_tk3(e, TK_TARGET, target);
    const int ax = (td.kind == 4 || td.kind == 5) ? 2 : 1;
    TK(e, TK_EXTRA) = mw_abs(comp(target, ax) - comp(bs, ax));
}
template <typename T>
MW_HD Out button_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> obj = obs3(obs, 4), tcp = tcp_center(e, td), target = tk3(e, TK_TARGET);
    const T tcp_to_obj = norm(obj - tcp), tcp_to_obj_init = norm(obj - tk3(e, TK_INITTCP));
    const int ax = (td.kind == 4 || td.kind == 5) ? 2 : 1;
    const T o2t = mw_abs(comp(target, ax) - comp(obj, ax));
    const T pressed = tolerance_lt(o2t, T(0), T(0.005), TK(e, TK_EXTRA));
    T near, reward, thr;
    if (td.kind == 4) {
        near = tolerance_lt(tcp_to_obj, T(0), T(0.01), tcp_to_obj_init);
        reward = 5 * hamacher(1 - obs[3], near);
        if (tcp_to_obj <= T(0.03)) reward += 5 * pressed;
        thr = T(0.024);
    } else if (td.kind == 5) {
        near = tolerance_lt(tcp_to_obj, T(0), T(0.01), tcp_to_obj_init);
        reward = 5 * hamacher(mw_max(obs[3], T(0)), near);
        if (tcp_to_obj <= T(0.03)) reward += 5 * pressed;
        thr = T(0.024);
    } else if (td.kind == 6) {
        near = tolerance_lt(tcp_to_obj, T(0), T(0.05), tcp_to_obj_init);
        reward = 2 * hamacher(mw_max(obs[3], T(0)), near);
        if (tcp_to_obj <= T(0.05)) reward += 8 * pressed;
        thr = T(0.02);
    } else {
        near = tolerance_lt(tcp_to_obj, T(0), T(0.01), tcp_to_obj_init);
        if (tcp_to_obj > T(0.07)) reward = 2 * hamacher((1 - obs[3]) / 2, near);
        else reward = 2 + 2 * (1 + obs[3]) + 4 * pressed * pressed;
        thr = T(0.03);
    }
    return Out{double(reward), double(o2t <= thr), make_info(tcp_to_obj <= T(0.05), obs[3] > 0, near, pressed, o2t, reward)};
}

// ---- coffee-button (8), coffee-pull (9), coffee-push (10): mug free joint FIRST (qpos[0:7]); reloc0 = coffee_machine ----
template <typename T>
MW_HD void coffee_set_mug(const Env<T> e, V3<T> p) {      // qpos[0:3] <- pos ; qvel[9:15] <- 0 (robot dofs: reference quirk)
    st3(e, e.lay().qpos, p);
    for (int k = 9; k < 15; k++) e.R(e.lay().qvel + k) = 0;
    forward(e);
}
template <typename T>
MW_HD void coffee_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC), rv1 = tk3(e, TK_RANDVEC + 3);
    if (td.kind == 8) {
        set_tk3(e, TK_OBJINIT, rv0);
        set_reloc(e, td, 0, rv0);
        coffee_set_mug(e, rv0 + v3<T>(0, T(-0.22), 0));
        set_tk3(e, TK_TARGET, rv0 + v3<T>(0, T(-0.22), T(0.3)) + v3<T>(0, T(0.03), 0));
    } else {
        coffee_set_mug(e, rv0);
        set_tk3(e, TK_OBJINIT, rv0);
        set_reloc(e, td, 0, (td.kind == 9 ? rv0 : rv1) + v3<T>(0, T(0.22), 0));
        set_tk3(e, TK_TARGET, rv1);
    }
}
template <typename T>
MW_HD Out coffee_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> obj = obs3(obs, 4), tcp = tcp_center(e, td), target = tk3(e, TK_TARGET), oi = tk3(e, TK_OBJINIT);
    const T tcp_to_obj = norm(obj - tcp);
    if (td.kind == 8) {
        const T o2t = mw_abs(target.y - obj.y);
        const T near = tolerance_lt(tcp_to_obj, T(0), T(0.05), norm(obj - tk3(e, TK_INITTCP)));
        const T pressed = tolerance_lt(o2t, T(0), T(0.005), T(0.03));
        T reward = 2 * hamacher(mw_max(obs[3], T(0)), near);
        if (tcp_to_obj <= T(0.05)) reward += 8 * pressed;
        return Out{double(reward), double(o2t <= T(0.02)), make_info(tcp_to_obj <= T(0.05), obs[3] > 0, near, pressed, o2t, reward)};
    }
    const T t2o = norm(scale3(obj - target, T(2), T(2), T(1))), t2oi = norm(scale3(oi - target, T(2), T(2), T(1)));
    const T in_place = tolerance_lt(t2o, T(0), T(0.05), t2oi);
    const T grasped = caging_base(e, td, act, obj, oi, T(0.02), T(0.05), T(0.04), T(0.05), T(0.7), false, true);
    T reward = hamacher(grasped, in_place);
    if (tcp_to_obj < T(0.04) && obs[3] > 0) reward += 1 + 5 * in_place;
    if (t2o < T(0.05)) reward = 10;
    const T d = norm(obj - target);
    const bool gs = touching_object(e, td, td.geom[G_OBJ]) && obs[3] > 0;
    return Out{double(reward), double(d <= T(0.07)), make_info(tcp_to_obj <= T(0.03), gs, grasped, in_place, d, reward)};
}

// ---- dial-turn (11): obs pos = B(dial) + 0.05*[sin q, -cos q, 0] ; reloc0 = dial ; qadr0 = knob_Joint_1 ; P_X0 = body dial ----
template <typename T>
MW_HD V3<T> dial_pos(const Env<T> e, const TaskDesc<T>& td) {
    const T q = e.R(e.lay().qpos + td.qadr[0]);
    return probe_pos(e, td.probe[P_X0]) + v3<T>(T(0.05) * sin(q), T(-0.05) * cos(q), 0);
}
template <typename T>
MW_HD void dial_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC);
    set_tk3(e, TK_OBJINIT, rv0);
    set_tk3(e, TK_TARGET, rv0 + v3<T>(0, T(0.03), T(0.03)));
    set_reloc(e, td, 0, rv0);
    st3(e, e.o_task + TK_EXTRA, dial_pos(e, td) + v3<T>(T(0.05), T(0.02), T(0.09)));   // dial_push_position (stale FK, like the reference)
}
template <typename T>
MW_HD Out dial_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> obj = dial_pos(e, td), push = obj + v3<T>(T(0.05), T(0.02), T(0.09)), tcp = tcp_center(e, td), target = tk3(e, TK_TARGET);
    const V3<T> push0 = tk3(e, TK_EXTRA);
    const T t2o = norm(obj - target), t2oi = norm(push0 - target);
    const T in_place = tolerance_lt(t2o, T(0), T(0.07), mw_abs(t2oi - T(0.07)));
    const T tcp_to_obj = norm(push - tcp), tcp_to_obj_init = norm(push0 - tk3(e, TK_INITTCP));
    T reach = tolerance_gauss(tcp_to_obj, T(0), T(0.005), mw_abs(tcp_to_obj_init - T(0.005)));
    reach = hamacher(reach, mw_min(mw_max(T(0), act[3]), T(1)));
    const T reward = 10 * hamacher(reach, in_place);
    return Out{double(reward), double(t2o <= T(0.07)), make_info(tcp_to_obj <= T(0.01), 1.0, reach, in_place, t2o, reward)};
}

// ---- door-close (13), door-open (15) [sawyer_door_pull], door-lock (14), door-unlock (16) [sawyer_door_lock] ----
// reloc0 = door ; qadr0/dadr0 = doorjoint ; P_X0 = body lock_link
template <typename T>
MW_HD void door_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC);
    if (td.kind == 13 || td.kind == 15) {
        set_tk3(e, TK_OBJINIT, rv0);
        set_tk3(e, TK_TARGET, rv0 + (td.kind == 13 ? v3<T>(T(0.2), T(-0.2), 0) : v3<T>(T(-0.3), T(-0.45), 0)));
        set_reloc(e, td, 0, rv0);
        set_joint(e, td.qadr[0], td.dadr[0], td.kind == 13 ? T(-1.5708) : T(0));
    } else if (td.kind == 14) {
        set_reloc(e, td, 0, rv0);
        for (int k = 0; k < 5; k++) substep(e);                 // raw mj_step x frame_skip
        const V3<T> oi = probe_pos(e, td.probe[P_X0]);          // body("lock_link").xpos (view; read at reset for the target)
        set_tk3(e, TK_OBJINIT, oi);
        set_tk3(e, TK_TARGET, oi + v3<T>(0, T(-0.04), T(-0.1)));
    } else {
        set_reloc(e, td, 0, rv0);
        set_joint(e, 9, 9, T(1.5708));
        const V3<T> oi = probe_pos(e, td.probe[P_X0]);
        set_tk3(e, TK_OBJINIT, oi);
        set_tk3(e, TK_TARGET, oi + v3<T>(T(0.1), T(-0.04), 0));
    }
}
template <typename T>
MW_HD Out door_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> obj = obs3(obs, 4), target = tk3(e, TK_TARGET);
    if (td.kind == 13) {
        const V3<T> tcp = tcp_center(e, td);
        const T tcp_to_target = norm(tcp - target), o2t = norm(obj - target);
        const T in_place = tolerance_gauss(o2t, T(0), T(0.05), norm(tk3(e, TK_OBJINIT) - target));
        const T hand_margin = norm(v3(td.hand_init[0], td.hand_init[1], td.hand_init[2]) - obj) + T(0.1);
        const T hand_in_place = tolerance_gauss(tcp_to_target, T(0), T(0.25 * 0.05), hand_margin);
        T reward = 3 * hand_in_place + 6 * in_place;
        if (o2t < T(0.05)) reward = 10;
        return Out{double(reward), double(o2t <= T(0.08)), make_info(0.0, 1.0, 1.0, hand_in_place, o2t, reward)};
    }
    if (td.kind == 15) {
        const T theta = e.R(e.lay().qpos + td.qadr[0]);
        const T grab = (mw_clamp(act[3], T(-1), T(1)) + 1) / 2;
        const V3<T> hand = obs3(obs, 0), door = obj + v3<T>(T(-0.05), 0, 0);
        const T thr = T(0.12), radius = mw_sqrt((hand.x - door.x) * (hand.x - door.x) + (hand.y - door.y) * (hand.y - door.y));
        const T floor_ = radius <= thr ? T(0) : T(0.04) * T(log(double(radius - thr))) + T(0.4);
        const T above = hand.z >= floor_ ? T(1) : tolerance_lt(floor_ - hand.z, T(0), T(0.01), floor_ / 2);
        const T in_place = tolerance_lt(norm(hand - door - v3<T>(T(0.05), T(0.03), T(-0.01))), T(0), thr / 2, T(0.5));
        const T ready = hamacher(above, in_place);
        const T pi = T(3.14159265358979323846);
        const T opened = T(0.2) * T(theta < -pi / 90) + T(0.8) * tolerance_lt(pi / 2 + pi / 6 + theta, T(0), T(0.5), pi / 3);
        T reward = 2 * hamacher(ready, grab) + 8 * opened;
        const bool ok = mw_abs(obs[4] - target.x) <= T(0.08);
        if (ok) reward = 10;
        return Out{double(reward), double(ok), make_info(ready, grab >= T(0.5), grab, opened, 0.0, reward)};
    }
    if (td.kind == 14) {
        const V3<T> tcp = probe_pos(e, td.probe[P_LPAD]);
        const T tcp_to_obj = norm(scale3(obj - tcp, T(0.25), T(1), T(0.5)));
        const T tcp_to_obj_init = tcp_to_obj;              // init_left_pad is a live view of the left pad
        const T o2t = mw_abs(target.z - obj.z);
        const T near = tolerance_lt(tcp_to_obj, T(0), T(0.01), tcp_to_obj_init);
        const T pressed = tolerance_lt(o2t, T(0), T(0.005), T(0.1));
        const T reward = 2 * hamacher(mw_max(obs[3], T(0)), near) + 8 * pressed;
        return Out{double(reward), double(o2t <= T(0.02)), make_info(tcp_to_obj <= T(0.05), obs[3] > 0, near, pressed, o2t, reward)};
    }
    // door-unlock: obj_init_pos is a live view of lock_link.xpos
    const V3<T> gripper = obs3(obs, 0), off = v3<T>(0, T(0.055), T(0.07)), oi_live = probe_pos(e, td.probe[P_X0]);
    const T s2l = norm(scale3(gripper + off - obj, T(0.25), T(1), T(0.5)));
    const T s2li = norm(scale3(tk3(e, TK_INITTCP) + off - oi_live, T(0.25), T(1), T(0.5)));
    const T ready = tolerance_lt(s2l, T(0), T(0.02), s2li);
    const T o2t = mw_abs(target.x - obj.x);
    const T pushed = tolerance_lt(o2t, T(0), T(0.005), T(0.1));
    const T reward = 2 * ready + 8 * pushed;
    return Out{double(reward), double(o2t <= T(0.02)), make_info(s2l <= T(0.05), obs[3] > 0, ready, pushed, o2t, reward)};
}

// ---- drawer-close (18), drawer-open (19): reloc0 = drawer ; obs pos = B(drawer_link) + offset ----
template <typename T>
MW_HD void drawer_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC);
    set_reloc(e, td, 0, rv0);
    if (td.kind == 18) {
        set_tk3(e, TK_TARGET, rv0 + v3<T>(0, T(-0.16), T(0.09)));
        set_joint(e, 9, -1, T(-0.15));                   // _set_obj_xyz(-maxDist): qpos[9] only, qvel untouched
        set_tk3(e, TK_OBJINIT, probe_pos(e, td.probe[P_OBJ0]) + v3<T>(0, T(-0.16), T(0.05)));
    } else {
        set_tk3(e, TK_OBJINIT, rv0);
        set_tk3(e, TK_TARGET, rv0 + v3<T>(0, T(-0.16 - 0.2), T(0.09)));
    }
}
template <typename T>
MW_HD Out drawer_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> obj = obs3(obs, 4), target = tk3(e, TK_TARGET);
    if (td.kind == 18) {
        const V3<T> oi = tk3(e, TK_OBJINIT), tcp = tcp_center(e, td);
        const T t2o = norm(obj - target), t2oi = norm(oi - target);
        const T in_place = tolerance_lt(t2o, T(0), T(0.05), mw_abs(t2oi - T(0.05)));
        const T tcp_to_obj = norm(obj - tcp), tcp_to_obj_init = norm(oi - tk3(e, TK_INITTCP));
        T reach = tolerance_gauss(tcp_to_obj, T(0), T(0.005), mw_abs(tcp_to_obj_init - T(0.005)));
        reach = hamacher(reach, mw_min(mw_max(T(0), act[3]), T(1)));
        T reward = hamacher(reach, in_place);
        if (t2o <= T(0.05 + 0.015)) reward = 1;
        reward *= 10;
        return Out{double(reward), double(t2o <= T(0.05 + 0.015)), make_info(tcp_to_obj <= T(0.01), 1.0, reach, in_place, t2o, reward)};
    }
    const V3<T> gripper = obs3(obs, 0);
    const T handle_error = norm(obj - target);
    const T opening = tolerance_lt(handle_error, T(0), T(0.02), T(0.2));
    const V3<T> handle_init = target + v3<T>(0, T(0.2), 0);
    const T ge = norm(scale3(obj - gripper, T(3), T(3), T(1))), gei = norm(scale3(handle_init - tk3(e, TK_INITTCP), T(3), T(3), T(1)));
    const T caging = tolerance_lt(ge, T(0), T(0.01), gei);
    const T reward = (caging + opening) * 5;
    const T gerr = norm(obj - gripper);
    return Out{double(reward), double(handle_error <= T(0.03)), make_info(gerr <= T(0.03), obs[3] > 0, caging, opening, handle_error, reward)};
}

// ---- faucet-open (20), faucet-close (21): reloc0 = faucetBase ----
template <typename T>
MW_HD void faucet_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC);
    set_tk3(e, TK_OBJINIT, rv0);
    set_reloc(e, td, 0, rv0);
    set_tk3(e, TK_TARGET, rv0 + v3<T>(td.kind == 20 ? T(0.175) : T(-0.175), 0, T(0.125)));
    if (td.kind == 21) forward(e);
}
template <typename T>
MW_HD Out faucet_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    V3<T> obj = obs3(obs, 4);
    if (td.kind == 20) obj = obj + v3<T>(T(-0.04), 0, T(0.03));
    const V3<T> tcp = tcp_center(e, td), target = tk3(e, TK_TARGET), oi = tk3(e, TK_OBJINIT);
    const T t2o = norm(obj - target), t2oi = norm(oi - target);
    const T in_place = tolerance_lt(t2o, T(0), T(0.07), mw_abs(t2oi - T(0.07)));
    const T tcp_to_obj = norm(obj - tcp), tcp_to_obj_init = norm(oi - tk3(e, TK_INITTCP));
    const T reach = tolerance_gauss(tcp_to_obj, T(0), T(0.01), mw_abs(tcp_to_obj_init - T(0.01)));
    T reward = (2 * reach + 3 * in_place) * 2;
    if (t2o <= T(0.07)) reward = 10;
    return Out{double(reward), double(t2o <= T(0.07)), make_info(tcp_to_obj <= T(0.01), 1.0, reach, in_place, t2o, reward)};
}

// ---- handle-press-side (23), handle-press (24), handle-pull-side (25), handle-pull (26): reloc0 = box ----
// probes: P_X0 = goal site (goalPress / goalPull) ; TK_EXTRA[0..2] = _handle_init_pos
template <typename T>
MW_HD void handle_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC);
    set_tk3(e, TK_OBJINIT, rv0);
    set_reloc(e, td, 0, rv0);
    set_joint(e, 9, 9, (td.kind == 23 || td.kind == 24) ? T(-0.001) : T(-0.1));
    set_tk3(e, TK_TARGET, probe_pos(e, td.probe[P_X0]));
    const V3<T> h0 = probe_pos(e, td.probe[P_OBJ0]);
    st3(e, e.o_task + TK_EXTRA, h0);
    if (td.kind == 25) set_tk3(e, TK_OBJINIT, h0);      // handle-pull-side re-captures obj_init_pos; handle-pull keeps rand_vec
}
template <typename T>
MW_HD Out handle_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> target = tk3(e, TK_TARGET), tcp = tcp_center(e, td);
    if (td.kind == 23 || td.kind == 24) {
        const V3<T> obj = probe_pos(e, td.probe[P_OBJ0]), h0 = tk3(e, TK_EXTRA);
        const T t2o = mw_abs(obj.z - target.z), t2oi = mw_abs(h0.z - target.z);
        const T in_place = tolerance_lt(t2o, T(0), T(0.02), mw_abs(t2oi - T(0.02)));
        const T tcp_to_obj = norm(obj - tcp), tcp_to_obj_init = norm(h0 - tk3(e, TK_INITTCP));
        const T reach = tolerance_lt(tcp_to_obj, T(0), T(0.02), mw_abs(tcp_to_obj_init - T(0.02)));
        T reward = hamacher(reach, in_place);
        if (t2o <= T(0.02)) reward = 1;
        reward *= 10;
        return Out{double(reward), double(t2o <= T(0.02)), make_info(tcp_to_obj <= T(0.05), 1.0, reach, in_place, t2o, reward)};
    }
    const V3<T> obj = obs3(obs, 4), oi = tk3(e, TK_OBJINIT);
    T t2o, in_place, grasped;
    bool lifted;
    if (td.kind == 25) {
        t2o = norm(obj - target);
        in_place = tolerance_lt(t2o, T(0), T(0.05), norm(oi - target));
        grasped = caging_base(e, td, act, obj, oi, T(0.032), T(0.06), T(0.01), T(0.01), T(1), true, false);
        lifted = obj.z - T(0.01) > oi.z;
    } else {
        t2o = mw_abs(target.z - obj.z);
        in_place = tolerance_lt(t2o, T(0), T(0.05), mw_abs(target.z - oi.z));
        grasped = caging_base(e, td, act, obj, oi, T(0.022), T(0.05), T(0.01), T(0.01), T(1), true, false);
        lifted = obj.y - T(0.01) > oi.z;               // sic: y vs z (envs/sawyer_handle_pull_v3.py:159)
    }
    T reward = hamacher(grasped, in_place);
    const T tcp_to_obj = norm(obj - tcp);
    if (tcp_to_obj < T(0.035) && obs[3] > 0 && lifted) reward += 1 + 5 * in_place;
    if (t2o < T(0.05)) reward = 10;
    const bool gs = obs[3] > 0 && obj.z - T(0.03) > oi.z;
    return Out{double(reward), double(t2o <= (td.kind == 25 ? T(0.08) : T(0.05))), make_info(tcp_to_obj <= T(0.05), gs, grasped, in_place, t2o, reward)};
}

// ---- lever-pull (27): reloc0 = lever ; qadr0 = LeverAxis ; TK_EXTRA[0..2] = _lever_pos_init ----
template <typename T>
MW_HD void lever_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC);
    set_tk3(e, TK_OBJINIT, rv0);
    set_reloc(e, td, 0, rv0);
    st3(e, e.o_task + TK_EXTRA, rv0 + v3<T>(T(0.12), T(-0.2), T(0.25)));
    set_tk3(e, TK_TARGET, rv0 + v3<T>(T(0.12), 0, T(0.25 + 0.2)));
}
template <typename T>
MW_HD Out lever_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> gripper = obs3(obs, 0), lever = obs3(obs, 4), off = v3<T>(0, T(0.055), T(0.07)), l0 = tk3(e, TK_EXTRA), target = tk3(e, TK_TARGET);
    const T s2l = norm(scale3(gripper + off - lever, T(4), T(1), T(4)));
    const T s2li = norm(scale3(tk3(e, TK_INITTCP) + off - l0, T(4), T(1), T(4)));
    const T ready = tolerance_lt(s2l, T(0), T(0.02), s2li);
    const T pi = T(3.14159265358979323846);
    const T angle = -e.R(e.lay().qpos + td.qadr[0]), err = mw_abs(angle - pi / 2);
    const T engagement = tolerance_lt(err, T(0), pi / 48, pi / 2 - pi / 12);
    const T in_place = tolerance_lt(norm(lever - target), T(0), T(0.04), norm(l0 - target));
    const T reward = 10 * hamacher(ready, in_place);
    return Out{double(reward), double(err <= pi / 24), make_info(s2l < T(0.03), ready > T(0.9), ready, engagement, s2l, reward)};
}

// ---- window-open (48), window-close (49): reloc0 = window ; qadr0 = window_slide ; TK_EXTRA[0..2] = window_handle_pos_init ----
template <typename T>
MW_HD void window_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    if (td.kind == 49) set_tk3(e, TK_INITTCP, tcp_center(e, td));
    const V3<T> rv0 = tk3(e, TK_RANDVEC);
    set_tk3(e, TK_OBJINIT, rv0);
    set_tk3(e, TK_TARGET, td.kind == 48 ? rv0 + v3<T>(T(0.2), 0, 0) : rv0);
    set_reloc(e, td, 0, rv0);
    const V3<T> h = probe_pos(e, td.probe[P_OBJ0]);                      // stale FK, like the reference
    st3(e, e.o_task + TK_EXTRA, td.kind == 48 ? h : h + v3<T>(T(0.2), 0, 0));
    e.R(e.lay().qpos + td.qadr[0]) = td.kind == 48 ? T(0) : T(0.2);          // data.joint("window_slide").qpos = ... (no forward)
}
template <typename T>
MW_HD Out window_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> obj = probe_pos(e, td.probe[P_OBJ0]), tcp = tcp_center(e, td), target = tk3(e, TK_TARGET), h0 = tk3(e, TK_EXTRA);
    const T t2o = mw_abs(obj.x - target.x);
    const T t2oi = td.kind == 48 ? mw_abs(TK(e, TK_OBJINIT) - target.x) : mw_abs(h0.x - target.x);
    const T in_place = tolerance_lt(t2o, T(0), T(0.05), mw_abs(t2oi - T(0.05)));
    const T tcp_to_obj = norm(obj - tcp), tcp_to_obj_init = norm(h0 - tk3(e, TK_INITTCP));
    const T reach = td.kind == 48 ? tolerance_lt(tcp_to_obj, T(0), T(0.02), mw_abs(tcp_to_obj_init - T(0.02)))
                                  : tolerance_gauss(tcp_to_obj, T(0), T(0.02), mw_abs(tcp_to_obj_init - T(0.02)));
    const T reward = 10 * hamacher(reach, in_place);
    return Out{double(reward), double(t2o <= T(0.05)), make_info(tcp_to_obj <= T(0.05), 1.0, reach, in_place, t2o, reward)};
}

// ---- plate-slide (31), -side (32), -back (33), -back-side (34): puck on two slide joints qpos[9:11]; reloc0 = puck_goal ----
template <typename T>
MW_HD void plate_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC), rv1 = tk3(e, TK_RANDVEC + 3);
    set_tk3(e, TK_OBJINIT, rv0);
    set_tk3(e, TK_TARGET, rv1);
    if (td.kind == 31) set_reloc(e, td, 0, rv1);
    if (td.kind == 34) set_reloc(e, td, 0, rv0);
    const T q0 = td.kind == 34 ? T(-0.15) : T(0), q1 = td.kind == 33 ? T(0.15) : T(0);
    e.R(e.lay().qpos + 9) = q0; e.R(e.lay().qpos + 10) = q1;          // _set_obj_xyz: qpos[9:11], qvel untouched
    forward(e);
}
template <typename T>
MW_HD T tolerance_lt_checked(T x, T lo, T hi, T margin) {   // the reference raises ValueError for margin < 0; defined here as margin 0
    return tolerance_lt(x, lo, hi, margin < 0 ? T(0) : margin);
}
template <typename T>
MW_HD Out plate_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> tcp = tcp_center(e, td), obj = obs3(obs, 4), target = tk3(e, TK_TARGET), oi = tk3(e, TK_OBJINIT);
    const T o2t = norm(obj - target), tcp_to_obj = norm(tcp - obj);
    const T m1 = norm(oi - target), m2 = norm(tk3(e, TK_INITTCP) - oi);
    T in_place, grasped, reward;
    if (td.kind == 31) {
        in_place = tolerance_lt(o2t, T(0), T(0.05), m1);
        grasped = tolerance_lt(tcp_to_obj, T(0), T(0.05), m2);
        reward = 8 * hamacher(grasped, in_place);
    } else {
        in_place = tolerance_lt_checked(o2t, T(0), T(0.05), m1 - T(0.05));
        grasped = tolerance_lt_checked(tcp_to_obj, T(0), T(0.05), m2 - T(0.05));
        reward = T(1.5) * grasped;
        if (tcp.z <= T(0.03) && tcp_to_obj < T(0.07)) reward = 2 + 7 * in_place;
    }
    if (o2t < T(0.05)) reward = 10;
    return Out{double(reward), double(o2t <= T(0.07)), make_info(tcp_to_obj <= T(0.03), 0.0, grasped, in_place, o2t, reward)};
}

// ---- assembly (0), disassemble (12) [sawyer_assembly_peg], hammer (22) ----
// probes: P_X0 = site RoundNut (wrench centre) ; reloc0 = peg (assembly/disassemble), box (hammer) ; qadr0 = NailSlideJoint
template <typename T>
MW_HD void wrench_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC), rv1 = tk3(e, TK_RANDVEC + 3);
    set_tk3(e, TK_OBJINIT, rv0);
    if (td.kind == 0) {
        set_tk3(e, TK_TARGET, rv1);
        set_obj_xyz(e, rv0);
        set_reloc(e, td, 0, rv1 - v3<T>(0, 0, T(0.05)));
    } else if (td.kind == 12) {
        set_tk3(e, TK_TARGET, rv0 + v3<T>(0, 0, T(0.15)));
        set_reloc(e, td, 0, rv0 + v3<T>(0, 0, T(0.03)));
        forward(e);
        set_obj_xyz(e, rv0);
    } else {
        set_reloc(e, td, 0, v3<T>(T(0.24), T(0.85), 0));
        set_tk3(e, TK_TARGET, probe_pos(e, td.probe[P_X0]));     // _get_site_pos("goal") from the current FK
        set_obj_xyz(e, rv0);                                     // _set_hammer_xyz == default layout qpos[9:12], qvel[9:15]
    }
}
template <typename T>
MW_HD Out wrench_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> hand = obs3(obs, 0), obj = obs3(obs, 4), target = tk3(e, TK_TARGET);
    V3<T> threshed = obj;
    const T half = td.kind == 22 ? T(0.07) : T(0.01);
    if (mw_abs(obj.x - hand.x) < half) threshed.x = hand.x;
    const T ideal[4] = {td.kind == 22 ? T(1) : T(0.707), 0, 0, td.kind == 22 ? T(0) : T(0.707)};
    T qe = 0;
    for (int k = 0; k < 4; k++) qe += (obs[7 + k] - ideal[k]) * (obs[7 + k] - ideal[k]);
    const T rquat = mw_max(1 - mw_sqrt(qe) / T(0.4), T(0));
    const T grab = caging_base(e, td, act, threshed, tk3(e, TK_OBJINIT), T(0.015), T(0.02), T(0.01), T(0.01), T(1), td.kind != 0, td.kind == 0);
    T in_place;
    bool success;
    if (td.kind == 0) {
        const V3<T> wc = probe_pos(e, td.probe[P_X0]);
        V3<T> pe = target - wc;
        const T radius = mw_sqrt(pe.x * pe.x + pe.y * pe.y);
        success = radius < T(0.02) && pe.z > 0;
        const T thr = success ? T(0.02) : T(0.01);
        T th = 0;
        if (radius > thr) th = T(0.02) * T(log(double(radius - thr))) + T(0.2);
        pe.z = th - wc.z;
        const bool lifted = wc.z > T(0.02) || radius < thr;
        in_place = T(0.1) * T(lifted) + T(0.9) * tolerance_lt(norm(scale3(pe, T(1), T(1), T(3))), T(0), T(0.02), T(0.4));
    } else if (td.kind == 12) {
        const V3<T> wc = probe_pos(e, td.probe[P_X0]);
        in_place = T(0.1) * T(wc.z > T(0.02)) + T(0.9) * tolerance_lt(norm(target + v3<T>(0, 0, T(0.1)) - wc), T(0), T(0.02), T(0.2));
        success = obs[6] > target.z;
    } else {
        const V3<T> head = obj + v3<T>(T(0.16), T(0.06), 0);
        in_place = T(0.1) * T(head.z > T(0.02)) + T(0.9) * tolerance_lt(norm(target - head), T(0), T(0.02), T(0.2));
        success = e.R(e.lay().qpos + td.qadr[0]) > T(0.09);
    }
    T reward = (2 * grab + 6 * in_place) * rquat;
    if (td.kind == 22 ? (success && reward > 5) : success) reward = 10;
    return Out{double(reward), double(success), make_info(rquat, grab >= T(0.5), grab, in_place, 0.0, reward)};
}

// ---- basketball (1): reloc0 = basket_goal ; P_X0 = site goal.  The reference assigns the site's WORLD position to its
// LOCAL model.site_pos at every reset_model call (envs/sawyer_basketball_v3.py:119-122), so the goal site drifts by
// 2*(hoop position) per reset.  TK_PERSIST[0..2] = accumulated model.site("goal").pos (survives resets). ----
enum { TK_PERSIST = TK_PERSIST0 };
template <typename T>
MW_HD void basketball_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC), rv1 = tk3(e, TK_RANDVEC + 3);
    const V3<T> oi{rv0.x, rv0.y, td.c[2]};
    set_tk3(e, TK_OBJINIT, oi);
    set_reloc(e, td, 0, rv1);
    set_obj_xyz(e, oi);
    st3(e, e.o_task + TK_EXTRA, probe_pos(e, td.probe[P_X0]));   // hoop site position for local pos 0 (= B + c)
}
template <typename T>
MW_HD Out basketball_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> obj = obs3(obs, 4), oi = tk3(e, TK_OBJINIT);
    V3<T> target = tk3(e, TK_TARGET);
    target.z = T(0.3);
    const T t2o = norm(scale3(obj - target, T(1), T(1), T(2))), t2oi = norm(scale3(oi - target, T(1), T(1), T(2)));
    const T in_place = tolerance_lt(t2o, T(0), T(0.08), t2oi);
    const T opened = obs[3], tcp_to_obj = norm(obj - tcp_center(e, td));
    T grasped = caging_base(e, td, act, obj, oi, T(0.025), T(0.06), T(0.01), T(0.005), T(1), true, false);
    const bool held = tcp_to_obj < T(0.035) && opened > 0 && obj.z - T(0.01) > oi.z;
    if (held) grasped = 1;
    T reward = hamacher(grasped, in_place);
    if (held) reward += 1 + 5 * in_place;
    if (t2o < T(0.08)) reward = 10;
    const bool gs = opened > 0 && obj.z - T(0.03) > oi.z;
    return Out{double(reward), double(t2o <= T(0.08)), make_info(tcp_to_obj <= T(0.05), gs, grasped, in_place, t2o, reward)};
}

// ---- box-close (3): reloc0 = boxbody ; c[6] = model z of boxbody ----
template <typename T>
MW_HD void box_close_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC), rv1 = tk3(e, TK_RANDVEC + 3);
    const V3<T> oi{rv0.x, rv0.y, td.c[2]};
    set_tk3(e, TK_OBJINIT, oi);
    set_tk3(e, TK_TARGET, rv1);
    set_reloc(e, td, 0, v3<T>(rv1.x, rv1.y, td.c[6]));
    for (int k = 0; k < 5; k++) substep(e);
    set_obj_xyz(e, oi);
}
template <typename T>
MW_HD Out box_close_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> hand = obs3(obs, 0), lid = obs3(obs, 4) + v3<T>(0, 0, T(0.02)), target = tk3(e, TK_TARGET);
    const T grab = mw_clamp((mw_clamp(act[3], T(-1), T(1)) + 1) / 2, T(0), T(1));
    const T ideal[4] = {T(0.707), 0, 0, T(0.707)};
    T qe = 0;
    for (int k = 0; k < 4; k++) qe += (obs[7 + k] - ideal[k]) * (obs[7 + k] - ideal[k]);
    const T rquat = mw_max(1 - mw_sqrt(qe) / T(0.2), T(0));
    const T radius = mw_sqrt((hand.x - lid.x) * (hand.x - lid.x) + (hand.y - lid.y) * (hand.y - lid.y));
    const T floor_ = radius <= T(0.02) ? T(0) : T(0.04) * T(log(double(radius - T(0.02)))) + T(0.4);
    const T above = hand.z >= floor_ ? T(1) : tolerance_lt(floor_ - hand.z, T(0), T(0.01), floor_ / 2);
    const T in_place = tolerance_lt(norm(hand - lid), T(0), T(0.02), T(0.5));
    const T ready = hamacher(above, in_place);
    const T lifted = T(0.2) * T(lid.z > T(0.04)) + T(0.8) * tolerance_lt(norm(scale3(target - lid, T(1), T(1), T(3))), T(0), T(0.05), T(0.25));
    T reward = 2 * hamacher(grab, ready) + 8 * lifted;
    const bool success = norm(obs3(obs, 4) - target) < T(0.08);
    if (success) reward = 10;
    reward *= rquat;
    return Out{double(reward), double(success), make_info(ready, grab >= T(0.5), grab, lifted, 0.0, reward)};
}

// ---- pick-out-of-hole (29), shelf-place (45), peg-insert-side (35), peg-unplug-side (36) ----
template <typename T>
MW_HD void misc_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC), rv1 = tk3(e, TK_RANDVEC + 3);
    if (td.kind == 29) {
        set_tk3(e, TK_OBJINIT, rv0); set_obj_xyz(e, rv0); set_tk3(e, TK_TARGET, rv1);
    } else if (td.kind == 45) {
        const V3<T> oi{rv0.x, rv0.y, probe_pos(e, td.probe[P_OBJ0]).z};     // adjust_initObjPos z = body z
        const V3<T> shelf = rv1 - v3<T>(0, 0, T(0.3));
        set_tk3(e, TK_OBJINIT, oi);
        set_reloc(e, td, 0, shelf);
        forward(e);
        set_tk3(e, TK_TARGET, c3(td, 6) + shelf);                            // model.site("goal").pos + model.body("shelf").pos
        set_obj_xyz(e, oi);
    } else if (td.kind == 35) {
        set_tk3(e, TK_OBJINIT, rv0);
        st3(e, e.o_task + TK_EXTRA, probe_pos(e, td.probe[P_X0]));           // peg_head_pos_init (before the peg is placed)
        set_obj_xyz(e, rv0);
        set_reloc(e, td, 0, rv1);
        set_tk3(e, TK_TARGET, rv1 + v3<T>(T(0.03), 0, T(0.13)));
    } else {
        set_reloc(e, td, 0, rv0);
        const V3<T> plug = rv0 + v3<T>(T(0.044), 0, T(0.131));
        st3(e, e.lay().qpos + 9, plug);
        st4(e, e.lay().qpos + 12, Q4<T>{1, 0, 0, 0});
        for (int k = 9; k < 12; k++) e.R(e.lay().qvel + k) = 0;
        forward(e);
        set_tk3(e, TK_OBJINIT, probe_pos(e, td.probe[P_OBJ0]));
        set_tk3(e, TK_TARGET, plug + v3<T>(T(0.15), 0, 0));
    }
}
template <typename T>
MW_HD T rect_prism_tolerance(V3<T> curr, V3<T> zero, V3<T> one) {
    auto in_range = [](T a, T b, T c) { return c >= b ? (b <= a && a <= c) : (c <= a && a <= b); };
    if (in_range(curr.x, zero.x, one.x) && in_range(curr.y, zero.y, one.y) && in_range(curr.z, zero.z, one.z)) {
        const V3<T> d = one - zero;
        return (curr.x - zero.x) / d.x * ((curr.y - zero.y) / d.y) * ((curr.z - zero.z) / d.z);
    }
    return 1;
}
template <typename T>
MW_HD Out misc_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> tcp = tcp_center(e, td), obj = obs3(obs, 4), target = tk3(e, TK_TARGET), oi = tk3(e, TK_OBJINIT);
    const T opened = obs[3], tcp_to_obj = norm(obj - tcp);
    if (td.kind == 29) {
        const T o2t = norm(obj - target);
        const T radius = mw_sqrt((tcp.x - oi.x) * (tcp.x - oi.x) + (tcp.y - oi.y) * (tcp.y - oi.y));
        const T floor_ = radius <= T(0.03) ? T(0) : T(0.015) * T(log(double(radius - T(0.03)))) + T(0.15);
        const T above = tcp.z >= floor_ ? T(1) : tolerance_lt(mw_max(floor_ - tcp.z, T(0)), T(0), T(0.01), T(0.02));
        const T grasped = caging_base(e, td, act, obj, oi, T(0.015), T(0.02), T(0.01), T(0.03), T(0.1), true, false);
        const T in_place = tolerance_lt(o2t, T(0), T(0.02), norm(oi - target));
        T reward = hamacher(grasped, in_place);
        const bool gs = tcp_to_obj < T(0.04) && obj.z - T(0.02) > oi.z && !(opened < T(0.33));
        if (gs) reward += 1 + 5 * hamacher(in_place, above);
        if (o2t < T(0.05)) reward = 10;
        return Out{double(reward), double(o2t <= T(0.07)), make_info(tcp_to_obj <= T(0.03), gs, grasped, in_place, o2t, reward)};
    }
    if (td.kind == 45) {
        const T o2t = norm(obj - target);
        T in_place = tolerance_lt(o2t, T(0), T(0.05), norm(oi - target));
        const T grasped = caging_base(e, td, act, obj, oi, T(0.02), T(0.05), T(0.01), T(0.01), T(1), false, false);
        T reward = hamacher(grasped, in_place);
        const bool inx = target.x - T(0.15) < obj.x && obj.x < target.x + T(0.15), inz = 0 < obj.z && obj.z < T(0.24);
        if (inz && inx && target.y - T(0.15) < obj.y && obj.y < target.y) {
            const T zs = (T(0.24) - obj.z) / T(0.24), ys = (obj.y - (target.y - T(0.15))) / T(0.15);
            in_place = mw_clamp(in_place - hamacher(ys, zs), T(0), T(1));
        }
        if (inz && inx && obj.y > target.y) in_place = 0;
        if (tcp_to_obj < T(0.025) && opened > 0 && obj.z - T(0.01) > oi.z) reward += 1 + 5 * in_place;
        if (o2t < T(0.05)) reward = 10;
        const bool gs = touching_object(e, td, td.geom[G_OBJ]) && opened > 0 && obj.z - T(0.02) > oi.z;
        return Out{double(reward), double(o2t <= T(0.07)), make_info(tcp_to_obj <= T(0.03), gs, grasped, in_place, o2t, reward)};
    }
    if (td.kind == 35) {
        const V3<T> head = probe_pos(e, td.probe[P_X0]);
        const T o2t = norm(scale3(head - target, T(1), T(2), T(2)));
        T in_place = tolerance_lt(o2t, T(0), T(0.07), norm(scale3(tk3(e, TK_EXTRA) - target, T(1), T(2), T(2))));
        const T cb1 = rect_prism_tolerance(head, probe_pos(e, td.probe[P_X1]), probe_pos(e, td.probe[P_X2]));
        const T cb2 = rect_prism_tolerance(head, probe_pos(e, td.probe[P_X3]), probe_pos(e, td.probe[P_X4]));
        in_place = hamacher(in_place, hamacher(cb2, cb1));
        T grasped = caging_base(e, td, act, obj, oi, T(0.0075), T(0.03), T(0.01), T(0.005), T(1), true, false);
        const bool held = tcp_to_obj < T(0.08) && opened > 0 && obj.z - T(0.01) > oi.z;
        if (held) grasped = 1;
        T reward = hamacher(grasped, in_place);
        if (held) reward += 1 + 5 * in_place;
        if (o2t <= T(0.07)) reward = 10;
        const bool gs = tcp_to_obj < T(0.02) && opened > 0 && obj.z - T(0.01) > oi.z;
        return Out{double(reward), double(o2t <= T(0.07)), make_info(tcp_to_obj <= T(0.03), gs, grasped, in_place, o2t, reward)};
    }
    const T o2t = norm(obj - target);
    const T grasped = caging_base(e, td, act, obj, oi, T(0.025), T(0.05), T(0.01), T(0.005), T(0.8), true, false);
    const T in_place = tolerance_lt(o2t, T(0), T(0.05), norm(oi - target));
    const bool gs = opened > T(0.5) && obj.x - oi.x > T(0.015);
    T reward = 2 * grasped;
    if (gs && tcp_to_obj < T(0.035)) reward = 1 + 2 * grasped + 5 * in_place;
    if (o2t <= T(0.05)) reward = 10;
    return Out{double(reward), double(o2t <= T(0.07)), make_info(tcp_to_obj <= T(0.03), gs, grasped, in_place, o2t, reward)};
}

// ---- stick-push (38), stick-pull (39): stick free joint qpos[9:16], container slides qpos[16:18] ----
// probes: OBJ0 = body stick, OBJ1 = body stick (scipy quat), OBJ2 = site insertion ; P_X0 = body object, P_X1 = site stick_end
// TK_EXTRA[0..2] = stick_init_pos ; c[6] = stick_init z
template <typename T>
MW_HD void stick_reset(const Env<T> e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv0 = tk3(e, TK_RANDVEC), rv1 = tk3(e, TK_RANDVEC + 3);
    const V3<T> si{rv0.x, rv0.y, td.c[6]};
    st3(e, e.o_task + TK_EXTRA, si);
    set_tk3(e, TK_TARGET, v3<T>(rv1.x, rv1.y, td.kind == 38 ? probe_pos(e, td.probe[P_OBJ2]).z : si.z));
    set_obj_xyz(e, si);                                           // _set_stick_xyz
    e.R(e.lay().qpos + 16) = 0; e.R(e.lay().qpos + 17) = td.kind == 38 ? T(0) : T(0.09);
    e.R(e.lay().qvel + 15) = e.R(e.lay().qvel + 15); e.R(e.lay().qvel + 16) = 0;   // qvel[16:18] = 0 (index 17 is out of range for nv = 17)
    forward(e);
    set_tk3(e, TK_OBJINIT, probe_pos(e, td.probe[P_X0]));
}
template <typename T>
MW_HD Out stick_eval(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> tcp = tcp_center(e, td), target = tk3(e, TK_TARGET), si = tk3(e, TK_EXTRA), oi = tk3(e, TK_OBJINIT);
    const T opened = obs[3];
    const bool touch = touching_object(e, td, td.geom[G_OBJ]);
    if (td.kind == 38) {
        const V3<T> stick = obs3(obs, 4) + v3<T>(T(0.015), 0, 0), container = obs3(obs, 11);
        const T tcp_to_stick = norm(stick - tcp), s2t = norm(stick - target), c2t = norm(container - target);
        const T sip = tolerance_lt_checked(s2t, T(0), T(0.12), norm(si - target) - T(0.12));
        const T cip = tolerance_lt_checked(c2t, T(0), T(0.12), norm(oi - target) - T(0.12));
        T grasped = caging_base(e, td, act, stick, si, T(0.04), T(0.05), T(0.01), T(0.01), T(1), true, false);
        T reward = grasped;
        if (tcp_to_stick < T(0.02) && opened > 0 && stick.z - T(0.01) > si.z) {
            grasped = 1;
            reward = 2 + 5 * sip + 3 * cip;
            if (c2t <= T(0.12)) reward = 10;
        }
        const bool gs = touch && opened > 0 && obs[6] - T(0.01) > si.z;
        const bool ok = norm(container - target) <= T(0.12);
        return Out{double(reward), double(gs && ok), make_info(tcp_to_stick <= T(0.03), gs, grasped, sip, c2t, reward)};
    }
    const V3<T> stick = obs3(obs, 4), handle = obs3(obs, 11), end = probe_pos(e, td.probe[P_X1]);
    const V3<T> container = handle + v3<T>(T(0.05), 0, 0), cinit = oi + v3<T>(T(0.05), 0, 0);
    const T tcp_to_stick = norm(stick - tcp), h2t = norm(handle - target);
    const T s2c = norm(scale3(stick - container, T(1), T(1), T(2)));
    const T sip = tolerance_lt(s2c, T(0), T(0.05), norm(scale3(si - cinit, T(1), T(1), T(2))));
    const T sip2 = tolerance_lt(norm(stick - target), T(0), T(0.05), norm(si - target));
    const T cip = tolerance_lt(norm(container - target), T(0), T(0.05), norm(oi - target));
    T grasped = caging_base(e, td, act, stick, oi, T(0.014), T(0.05), T(0.01), T(0.01), T(1), true, false);
    const bool gsr = tcp_to_stick < T(0.02) && opened > 0 && stick.z - T(0.01) > si.z;
    if (gsr) grasped = 1;
    const bool inserted = end.x >= handle.x && mw_abs(end.y - handle.y) <= T(0.04) && mw_abs(end.z - handle.z) <= T(0.06);
    const T ipg = hamacher(grasped, sip);
    T reward = ipg;
    if (gsr) {
        reward = 1 + ipg + 5 * sip;
        if (inserted) {
            reward = 1 + ipg + 5 + 2 * sip2 + cip;
            if (h2t <= T(0.12)) reward = 10;
        }
    }
    const bool gs = touch && opened > 0 && stick.z - T(0.02) > oi.z;
    return Out{double(reward), double(h2t <= T(0.12) && inserted), make_info(tcp_to_stick <= T(0.03), gs, grasped, sip, h2t, reward)};
}

// model writes that the FIRST reset_model pass leaves behind (the physics of that pass is discarded by mj_resetData,
// its `model.body(X).pos = ...` writes are not): apply them before the replayed second pass.
template <typename T>
MW_HD void task_model_writes(const Env<T> e, const TaskDesc<T>& td) {
    const V3<T> rv0 = tk3(e, TK_RANDVEC), rv1 = tk3(e, TK_RANDVEC + 3);
    switch (td.kind) {
    case 37: set_reloc(e, td, 0, rv1); break;
    case 4: case 5: case 6: case 7: case 8: case 11: case 13: case 15: case 14: case 16:
    case 18: case 19: case 20: case 21: case 23: case 24: case 25: case 26: case 27: case 48: case 49: set_reloc(e, td, 0, rv0); break;
    case 9: set_reloc(e, td, 0, rv0 + v3<T>(0, T(0.22), 0)); break;
    case 31: case 1: set_reloc(e, td, 0, rv1); break;
    case 34: set_reloc(e, td, 0, rv0); break;
    case 0: set_reloc(e, td, 0, rv1 - v3<T>(0, 0, T(0.05))); break;
    case 12: set_reloc(e, td, 0, rv0 + v3<T>(0, 0, T(0.03))); break;
    case 22: set_reloc(e, td, 0, v3<T>(T(0.24), T(0.85), 0)); break;
    case 3: set_reloc(e, td, 0, v3<T>(rv1.x, rv1.y, td.c[6])); break;
    case 45: set_reloc(e, td, 0, rv1 - v3<T>(0, 0, T(0.3))); break;
    case 35: set_reloc(e, td, 0, rv1); break;
    case 36: set_reloc(e, td, 0, rv0); break;
    case 10: set_reloc(e, td, 0, rv1 + v3<T>(0, T(0.22), 0)); break;
    default: break;
    }
}

template <typename T>
MW_HD void task_reset_model(const Env<T> e, const TaskDesc<T>& td) {
    switch (td.kind) {
    case 43: case 44: reach_reset(e, td); break;
    case 40: case 41: case 42: case 30: case 28: pushpick_reset(e, td); break;
    case 47: case 46: case 37: case 17: sweepfam_reset(e, td); break;
    case 2: bin_picking_reset(e, td); break;
    case 4: case 5: case 6: case 7: button_reset(e, td); break;
    case 8: case 9: case 10: coffee_reset(e, td); break;
    case 11: dial_reset(e, td); break;
    case 13: case 14: case 15: case 16: door_reset(e, td); break;
    case 18: case 19: drawer_reset(e, td); break;
    case 20: case 21: faucet_reset(e, td); break;
    case 23: case 24: case 25: case 26: handle_reset(e, td); break;
    case 27: lever_reset(e, td); break;
    case 48: case 49: window_reset(e, td); break;
    case 31: case 32: case 33: case 34: plate_reset(e, td); break;
    case 0: case 12: case 22: wrench_reset(e, td); break;
    case 1: basketball_reset(e, td); break;
    case 3: box_close_reset(e, td); break;
    case 29: case 45: case 35: case 36: misc_reset(e, td); break;
    case 38: case 39: stick_reset(e, td); break;
    default: reset_hand(e, td); break;
    }
}
template <typename T>
MW_HD void task_evaluate(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act, T* reward, T* success, Info* info) {
    Out o{0, 0, Info{0, 0, 0, 0, 0, 0}};
    switch (td.kind) {
    case 43: case 44: o = reach_eval(e, td, obs, act); break;
    case 40: o = push_eval(e, td, obs, act); break;
    case 30: o = pick_place_eval(e, td, obs, act); break;
    case 42: o = push_back_eval(e, td, obs, act); break;
    case 41: case 28: o = wall_eval(e, td, obs, act); break;
    case 47: case 46: case 37: case 17: o = sweepfam_eval(e, td, obs, act); break;
    case 2: o = bin_picking_eval(e, td, obs, act); break;
    case 4: case 5: case 6: case 7: o = button_eval(e, td, obs, act); break;
    case 8: case 9: case 10: o = coffee_eval(e, td, obs, act); break;
    case 11: o = dial_eval(e, td, obs, act); break;
    case 13: case 14: case 15: case 16: o = door_eval(e, td, obs, act); break;
    case 18: case 19: o = drawer_eval(e, td, obs, act); break;
    case 20: case 21: o = faucet_eval(e, td, obs, act); break;
    case 23: case 24: case 25: case 26: o = handle_eval(e, td, obs, act); break;
    case 27: o = lever_eval(e, td, obs, act); break;
    case 48: case 49: o = window_eval(e, td, obs, act); break;
    case 31: case 32: case 33: case 34: o = plate_eval(e, td, obs, act); break;
    case 0: case 12: case 22: o = wrench_eval(e, td, obs, act); break;
    case 1: o = basketball_eval(e, td, obs, act); break;
    case 3: o = box_close_eval(e, td, obs, act); break;
    case 29: case 45: case 35: case 36: o = misc_eval(e, td, obs, act); break;
    case 38: case 39: o = stick_eval(e, td, obs, act); break;
    default: break;
    }
    *reward = T(o.reward); *success = T(o.success); *info = o.info;
}

// state that survives a reset (only basketball's accumulated goal-site position): called after reset_model / snapshot load
template <typename T>
MW_HD void task_after_reset(const Env<T> e, const TaskDesc<T>& td, V3<T> persist_before, T* obs39) {
    if (td.kind != 1) return;
    const V3<T> P = tk3(e, TK_EXTRA), s = persist_before + P * T(2);
    st3(e, e.o_task + TK_PERSIST, s);
    set_tk3(e, TK_TARGET, P + s);
    // the reset observation is built from the FK that precedes the last site write: it shows s (= 2P + s_before), the
    // episode itself sees P + s
    if (!td.partially_observable) { obs39[36] = s.x; obs39[37] = s.y; obs39[38] = s.z; }
}

}  // namespace mw
#include "mw_tasks_v1.hpp"   // reward_function_version = "v1": selected per context at run time (mw_config.reward_version, World::reward_v1)
namespace mw {

// =========================================================================== env-level reset / step
// SawyerXYZEnv.reset (:664-682) second pass semantics: mj_resetData -> reset_model -> obs with prev := curr.
// (The first reset_model pass only leaves model writes behind; they are functions of rand_vec and are
//  re-applied by the second pass, see DESIGN.md "reset".)
template <typename T>
MW_HD void env_reset(const Env<T> e, const TaskDesc<T>& td, T* obs39, bool reward_v1) {
    reset_data(e);
    TK(e, TK_PATHLEN) = 0; TK(e, TK_ELAPSED) = 0; TK(e, TK_EPRET) = 0; TK(e, TK_EPRET_LO) = 0; TK(e, TK_EPLEN) = 0; TK(e, TK_SUCCESS) = 0;
    for (int k = 0; k < 16; k++) TK(e, TK_EXTRA + k) = 0;
    const V3<T> persist = tk3(e, TK_PERSIST0);
    task_model_writes(e, td);
    task_reset_model(e, td);
    if (reward_v1) task_reset_v1(e, td);          // (the per-env quantities the v1 branches keep on `self`: spare reals of the task block)
    get_obs(e, td, obs39);
    for (int k = 0; k < 18; k++) { obs39[18 + k] = obs39[k]; TK(e, TK_PREVOBS + k) = obs39[k]; }
    task_after_reset(e, td, persist, obs39);
}

// SawyerXYZEnv.step (:579-642) up to (obs, reward, success, info); wrappers are applied by the caller
// The final mj_forward is run LAZILY: the observation and every reward read only frames (kinematics); the collision ->
// constraint -> solver half runs when (and only when) the task's reward asks for contact forces (touching_object does it on
// demand), or always with `full_forward` (mw_config.full_forward: engine-level comparisons that read ncon / nefc / efc_force
// after a step).  Same values either way: the two halves do not feed back into each other, and the next step's first
// mj_step recomputes everything from (qpos, qvel, mocap, ctrl).  tests/test_lazy_forward.py holds the equivalence.
template <typename T>
MW_HD void env_step(const Env<T> e, const TaskDesc<T>& td, const T* act, T* obs39, T* reward, T* success, Info* info, bool full_forward, bool reward_v1) {
    // set_xyz_action: mocap += clip(a,-1,1)*0.01, clipped to the mocap box
    // (the reference multiplies the float32 action by action_scale in float32: numpy keeps float32 * python-float in float32)
    for (int k = 0; k < 3; k++) {
        const float a = fminf(fmaxf(float(act[k]), -1.0f), 1.0f);
        const float delta = a * 0.01f;
        e.R(e.lay().mocap + k) = mw_clamp(e.R(e.lay().mocap + k) + T(delta), td.mocap_low[k], td.mocap_high[k]);
    }
    e.R(e.lay().ctrl) = act[3]; e.R(e.lay().ctrl + 1) = -act[3];
    for (int k = 0; k < 5; k++) substep(e);
    TK(e, TK_PATHLEN) += 1;
    kinematics(e);
    e.I(e.lay().icount + IC_DYN_VALID) = 0;
    // WAVE-UNIFORM decision (ballot): a workgroup holds the environments of one SCENE, which several tasks can share (coffee-button
    // next to coffee-pull / -push, ...); the lanes whose reward does not read contact forces run the second half along with the
    // ones whose reward does (same wave time: the wave executes it anyway; no persistent state is touched), so that its
    // non-inlined stages are entered by every live lane of the wave or by none.
    if (mw_any(full_forward || task_touches(td.kind))) forward_dynamics(e);
    MW_TICK(t_o0)
    get_obs(e, td, obs39);
    clip_obs(td, obs39);
    MW_TICK(t_o1)
    if (reward_v1) task_evaluate_v1(e, td, obs39, act, reward, success, info);          // (wave-uniform: one flag per context)
    else task_evaluate(e, td, obs39, act, reward, success, info);
#if defined(MW_STEP_FINE) && defined(MW_SOLVER_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    MW_TICK(t_o2)          // step-level timers (-DMW_STEP_FINE): slot 2 = get_obs + clip, slot 3 = task_evaluate
    e.I(e.lay().icount + 4 + 2) += (int)((t_o1 - t_o0) >> 4); e.I(e.lay().icount + 4 + 3) += (int)((t_o2 - t_o1) >> 4);
#endif
}

}  // namespace mw
