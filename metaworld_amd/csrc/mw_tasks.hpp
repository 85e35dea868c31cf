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
    TK_END = 56
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

template <typename T> MW_HD T& TK(const Env<T>& e, int k) { return e.R(e.L.task + k); }
template <typename T> MW_HD V3<T> tk3(const Env<T>& e, int k) { return ld3(e, e.L.task + k); }
template <typename T> MW_HD void set_tk3(const Env<T>& e, int k, V3<T> v) { st3(e, e.L.task + k, v); }

template <typename T>
MW_HD V3<T> tcp_center(const Env<T>& e, const TaskDesc<T>& td) {
    return (probe_pos(e, td.probe[P_RTCP]) + probe_pos(e, td.probe[P_LTCP])) * T(0.5);
}

// _get_curr_obs_combined_no_goal (sawyer_xyz_env.py:475-511)
template <typename T>
MW_HD void curr_obs(const Env<T>& e, const TaskDesc<T>& td, T* o18) {
    const V3<T> hand = probe_pos(e, td.probe[P_HAND]);
    o18[0] = hand.x; o18[1] = hand.y; o18[2] = hand.z;
    const T gd = norm(probe_pos(e, td.probe[P_RCLAW]) - probe_pos(e, td.probe[P_LCLAW]));
    o18[3] = mw_clamp(gd / T(0.1), T(0), T(1));
    for (int k = 4; k < 18; k++) o18[k] = 0;
    for (int i = 0; i < td.nobj; i++) {
        T* o = o18 + 4 + 7 * i;
        const V3<T> p = probe_pos(e, td.probe[P_OBJ0 + 2 * i]);
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
MW_HD void get_obs(const Env<T>& e, const TaskDesc<T>& td, T* obs39) {
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
MW_HD void reset_hand(const Env<T>& e, const TaskDesc<T>& td, int steps = 50) {
    for (int s = 0; s < steps; s++) {
        st3(e, e.L.mocap, v3(td.hand_init[0], td.hand_init[1], td.hand_init[2]));
        e.R(e.L.ctrl) = -1; e.R(e.L.ctrl + 1) = 1;
        for (int k = 0; k < 5; k++) substep(e);
    }
    set_tk3(e, TK_INITTCP, tcp_center(e, td));
}

// default _set_obj_xyz (:351-361): qpos[9:12] <- pos ; qvel[9:15] <- 0 ; set_state -> mj_forward
template <typename T>
MW_HD void set_obj_xyz(const Env<T>& e, V3<T> p) {
    st3(e, e.L.qpos + 9, p);
    for (int k = 9; k < 15; k++) e.R(e.L.qvel + k) = 0;
    forward(e);
}

struct Info { float near_object, grasp_success, grasp_reward, in_place_reward, obj_to_target, unscaled_reward; };

// =========================================================================== per-task code
// ---- reach-v3 (id 43) / reach-wall-v3 (id 44): metaworld/envs/sawyer_reach_v3.py:119-161, sawyer_reach_wall_v3.py
template <typename T>
MW_HD void reach_reset(const Env<T>& e, const TaskDesc<T>& td) {
    reset_hand(e, td);
    const V3<T> rv_obj = tk3(e, TK_RANDVEC), rv_goal = tk3(e, TK_RANDVEC + 3);
    set_tk3(e, TK_TARGET, rv_goal);
    set_tk3(e, TK_OBJINIT, rv_obj);
    set_obj_xyz(e, rv_obj);
}
template <typename T>
MW_HD void reach_eval(const Env<T>& e, const TaskDesc<T>& td, const T* obs, const T* act, T* reward, T* success, Info* info) {
    const V3<T> tcp = tcp_center(e, td), target = tk3(e, TK_TARGET);
    const T d = norm(tcp - target);
    const T margin = norm(v3(td.hand_init[0], td.hand_init[1], td.hand_init[2]) - target);
    const T in_place = tolerance_lt(d, T(0), T(0.05), margin);
    *reward = 10 * in_place;
    *success = d <= T(0.05) ? T(1) : T(0);
    info->near_object = float(d); info->grasp_success = 1.f; info->grasp_reward = float(d);
    info->in_place_reward = float(in_place); info->obj_to_target = float(d); info->unscaled_reward = float(*reward);
}

template <typename T>
MW_HD bool task_supported(int kind) { return kind == 43 || kind == 44; }

template <typename T>
MW_HD void task_reset_model(const Env<T>& e, const TaskDesc<T>& td) {
    switch (td.kind) {
    case 43: case 44: reach_reset(e, td); break;
    default: break;
    }
}
template <typename T>
MW_HD void task_evaluate(const Env<T>& e, const TaskDesc<T>& td, const T* obs, const T* act, T* reward, T* success, Info* info) {
    switch (td.kind) {
    case 43: case 44: reach_eval(e, td, obs, act, reward, success, info); break;
    default: *reward = 0; *success = 0; *info = Info{0, 0, 0, 0, 0, 0}; break;
    }
}

// =========================================================================== env-level reset / step
// SawyerXYZEnv.reset (:664-682) second pass semantics: mj_resetData -> reset_model -> obs with prev := curr.
// (The first reset_model pass only leaves model writes behind; they are functions of rand_vec and are
//  re-applied by the second pass, see DESIGN.md "reset".)
template <typename T>
MW_HD void env_reset(const Env<T>& e, const TaskDesc<T>& td, T* obs39) {
    reset_data(e);
    TK(e, TK_PATHLEN) = 0; TK(e, TK_ELAPSED) = 0; TK(e, TK_EPRET) = 0; TK(e, TK_EPLEN) = 0; TK(e, TK_SUCCESS) = 0;
    for (int k = 0; k < 16; k++) TK(e, TK_EXTRA + k) = 0;
    task_reset_model(e, td);
    get_obs(e, td, obs39);
    for (int k = 0; k < 18; k++) { obs39[18 + k] = obs39[k]; TK(e, TK_PREVOBS + k) = obs39[k]; }
}

// SawyerXYZEnv.step (:579-642) up to (obs, reward, success, info); wrappers are applied by the caller
template <typename T>
MW_HD void env_step(const Env<T>& e, const TaskDesc<T>& td, const T* act, T* obs39, T* reward, T* success, Info* info) {
    // set_xyz_action: mocap += clip(a,-1,1)*0.01, clipped to the mocap box
    for (int k = 0; k < 3; k++) {
        const T a = mw_clamp(act[k], T(-1), T(1));
        e.R(e.L.mocap + k) = mw_clamp(e.R(e.L.mocap + k) + a * T(0.01), td.mocap_low[k], td.mocap_high[k]);
    }
    e.R(e.L.ctrl) = act[3]; e.R(e.L.ctrl + 1) = -act[3];
    for (int k = 0; k < 5; k++) substep(e);
    TK(e, TK_PATHLEN) += 1;
    forward(e);
    get_obs(e, td, obs39);
    clip_obs(td, obs39);
    task_evaluate(e, td, obs39, act, reward, success, info);
}

}  // namespace mw
