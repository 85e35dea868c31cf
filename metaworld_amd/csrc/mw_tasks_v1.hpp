// mw_tasks_v1.hpp -- the reference's `reward_function_version="v1"` branches (the `else:` of every
// metaworld/envs/sawyer_*_v3.py::compute_reward, with the success / info composition of its evaluate_state), one environment per
// lane.  Selected per CONTEXT at run time (mw_config.reward_version = 1 -> World::reward_v1; rounds 2-4 compiled it into a second
// library, libmwgpu_v1.so): a v1 context evaluates v1 rewards for every task.  Physics, observations, resets and wrappers are shared.  The per-env quantities the v1 branches keep on `self` (maxDist, maxPullDist, heightTarget, pickCompleted, ...) live
// in the spare reals of the task block (TK_V1 ...), are set by task_reset_v1 right after reset_model and so travel with the reset
// snapshots.  Tasks without a restatement yet fall through to the v2 evaluation; metaworld_amd/tasks.py::V1_TASKS lists the ported
// ones and the Python boundary refuses the others.
#pragma once

namespace mw {

enum {
    TK_V1 = TK_END,      // first spare real of the task block
    V1_MAXA = TK_V1,     // maxDist / maxReachDist / maxPullDist / maxPushDist
    V1_MAXB,             // maxPlacingDist / maxPlaceDist / maxHammerDist
    V1_OBJH,             // objHeight / hammerHeight / stickHeight / obj_height
    V1_HTARGET,          // heightTarget
    V1_PICKED,           // pickCompleted
    V1_PLACED,           // placeCompleted
    V1_REACHED,          // reachCompleted
    V1_END
};
static_assert(V1_END <= TASK_NREAL, "task block too small for the v1 state");

template <typename T>
MW_HD V3<T> finger_com(const Env<T> e, const TaskDesc<T>& td) {      // (rightEndEffector + leftEndEffector) / 2
    return (probe_pos(e, td.probe[P_RTCP]) + probe_pos(e, td.probe[P_LTCP])) * T(0.5);
}
// c1 * (exp(-d^2 / c2) + exp(-d^2 / c3)) with the constants every v1 branch uses
template <typename T>
MW_HD T v1_bumps(T d) { return T(1000) * (exp(-(d * d) / T(0.01)) + exp(-(d * d) / T(0.001))); }

// ---- reach-v3 (43), reach-wall-v3 (44): sawyer_reach_v3.py:163-181, sawyer_reach_wall_v3.py ----
template <typename T>
MW_HD Out reach_eval_v1(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> goal = tk3(e, TK_TARGET);
    const T reachDist = norm(finger_com(e, td) - goal);
    T reachRew = T(1000) * (TK(e, V1_MAXA) - reachDist) + v1_bumps(reachDist);
    reachRew = mw_max(reachRew, T(0));
    // evaluate_state: (reward, reach_dist, in_place) -> success = reach_dist <= 0.05
    if (td.kind == 44)
        return Out{double(reachRew), double(reachDist <= T(0.05)), make_info(0.0, 0.0, 0.0, 0.0, reachDist, reachRew)};
    return Out{double(reachRew), double(reachDist <= T(0.05)), make_info(reachDist, 1.0, reachDist, 0.0, reachDist, reachRew)};
}

// ---- "reach, then move the fixture" family: 27 tasks share one v1 shape -------------------------------------------------------
//   reward = -reachDist + (reachDist < 0.05 ? [max(., 0)] (1000 (maxDist - dist) + c1 (exp(-dist^2/c2) + exp(-dist^2/c3))) : 0)
// with reachDist = |obj - finger| (finger = the two end-effector sites' mean, or the left one) and dist = how far the fixture is
// from its goal along one axis / in the xy plane / in space.  The tuple returned is (reward, 0, 0, dist, 0, 0) except where
// noted; evaluate_state then builds success / info from it exactly as it does for v2 (a zero `tcp_to_obj` makes
// near_object = 1, a zero `tcp_open` makes grasp_success = 0, ...).  Per task: button-press-topdown(-wall) :180-200, button-press
// (-wall) :170-195, coffee-button, drawer-close / -open, dial-turn, door-close / -lock / -open / -unlock, faucet-open / -close,
// handle-press(-side), handle-pull, lever-pull, plate-slide(-side, -back, -back-side), window-open / -close.
enum { M_AX0 = 0, M_AX1, M_AX2, M_XY, M_XYZ };
struct V1Fixture { int kind, left_finger, metric, clamp; double thr; int near_object, grasp_success, form; };
// form 4 = (reward, 0, 0, |obj - goal| in space, 0, 0) although the reward uses the xy distance (coffee-push)
// form: 0 = (reward, 0, 0, dist, 0, 0); 1 = door-open (reward, 0, 0, 0) with its own success; 2 = door-close (reward, dist, 0);
//       3 = lever-pull (reward, 0, 0, dist, 0): obj_to_target is its 2nd element = 0
MW_HD V1Fixture v1_fixture(int kind) {
    switch (kind) {
    case 4: case 5: return {kind, 0, M_AX2, 1, 0.024, 1, 0, 0};
    case 6: return {kind, 1, M_AX1, 1, 0.02, 1, 0, 0};
    case 7: return {kind, 1, M_AX1, 1, 0.03, 1, 0, 0};
    case 8: return {kind, 1, M_AX1, 1, 0.02, 1, 0, 0};
    case 10: return {kind, 0, M_XY, 1, 0.07, 1, 0, 4};                 // coffee-push
    case 37: case 41: case 42: return {kind, 0, M_XY, 1, 0.07, 1, 0, 0};   // soccer, push-wall, push-back
    case 40: return {kind, 0, M_XY, 1, 0.05, 1, 0, 0};                 // push
    case 11: return {kind, 0, M_AX1, 1, 0.07, 1, 1, 0};
    case 13: return {kind, 0, M_XY, 1, 0.08, 0, 1, 2};
    case 14: case 16: return {kind, 0, M_XYZ, 1, 0.02, 1, 0, 0};
    case 15: return {kind, 0, M_XY, 1, 0.08, 0, 0, 1};
    case 18: return {kind, 0, M_AX1, 1, 0.05 + 0.015, 1, 1, 0};
    case 19: return {kind, 0, M_AX1, 1, 0.03, 1, 0, 0};
    case 20: case 21: return {kind, 0, M_XYZ, 1, 0.07, 1, 1, 0};
    case 23: case 24: return {kind, 1, M_AX2, 1, 0.02, 1, 1, 0};
    case 25: return {kind, 1, M_AX2, 1, 0.08, 1, 0, 0};
    case 26: return {kind, 1, M_AX2, 1, 0.05, 1, 0, 0};
    case 27: return {kind, 0, M_XYZ, 1, 3.14159265358979323846 / 24, 1, 0, 3};
    case 31: case 32: case 33: case 34: return {kind, 0, M_XY, 1, 0.07, 1, 0, 0};
    case 48: case 49: return {kind, 0, M_AX0, 0, 0.05, 1, 1, 0};
    default: return {-1, 0, 0, 0, 0, 0, 0, 0};
    }
}
template <typename T>
MW_HD T v1_metric(int metric, V3<T> a, V3<T> b) {
    switch (metric) {
    case M_AX0: return mw_abs(a.x - b.x);
    case M_AX1: return mw_abs(a.y - b.y);
    case M_AX2: return mw_abs(a.z - b.z);
    case M_XY: return mw_sqrt((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y));
    default: return norm(a - b);
    }
}
template <typename T>
MW_HD Out fixture_eval_v1(const Env<T> e, const TaskDesc<T>& td, const V1Fixture f, const T* obs) {
    const V3<T> obj = obs3(obs, 4), goal = tk3(e, TK_TARGET);
    const V3<T> finger = f.left_finger ? probe_pos(e, td.probe[P_LTCP]) : finger_com(e, td);
    const T dist = v1_metric(f.metric, obj, goal), reachDist = norm(obj - finger);
    T rew2 = 0;
    if (reachDist < T(0.05)) {
        rew2 = T(1000) * (TK(e, V1_MAXA) - dist) + v1_bumps(dist);
        if (f.clamp) rew2 = mw_max(rew2, T(0));
    }
    TK(e, V1_REACHED) = reachDist < T(0.05) ? T(1) : T(0);
    const T reward = -reachDist + rew2;
    if (f.form == 1)      // door-open: success from the handle's x alone (sawyer_door_v3.py evaluate_state)
        return Out{double(reward), double(mw_abs(obs[4] - goal.x) <= T(0.08)), make_info(0.0, 0.0, 0.0, 0.0, 0.0, reward)};
    if (f.form == 2)      // door-close: info = {obj_to_target: dist, in_place: 0, near_object: 0, grasp_success: 1, grasp_reward: 1}
        return Out{double(reward), double(dist <= T(f.thr)), make_info(0.0, 1.0, 1.0, 0.0, dist, reward)};
    if (f.form == 4) {
        const T d3 = norm(obj - goal);
        return Out{double(reward), double(d3 <= T(f.thr)), make_info(double(f.near_object), double(f.grasp_success), 0.0, 0.0, d3, reward)};
    }
    if (f.form == 3)      // lever-pull: lever_error := dist, everything else of the tuple 0
        return Out{double(reward), double(dist <= T(f.thr)), make_info(1.0, 0.0, 0.0, 0.0, 0.0, reward)};
    return Out{double(reward), double(dist <= T(f.thr)), make_info(double(f.near_object), double(f.grasp_success), 0.0, 0.0, dist, reward)};
}
// ---- coffee-pull (9): sawyer_coffee_pull_v3.py v1 branch ----
template <typename T>
MW_HD Out coffee_pull_eval_v1(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> obj = obs3(obs, 4), goal = tk3(e, TK_TARGET), finger = finger_com(e, td);
    const T reachDist = norm(finger - obj), pullDist = v1_metric(M_XY, obj, goal);
    const T itz = TK(e, TK_INITTCP + 2);
    const T reachDistxy = mw_sqrt(obj.x * obj.x + obj.y * obj.y + itz * itz);      // |(obj.x, obj.y, init_tcp.z)| (sic: not a difference)
    T reachRew;
    if (reachDistxy < T(0.05)) {
        reachRew = -reachDist + T(0.1);
        if (reachDist < T(0.05)) {          // max(action[-1], 0) / 50 in the action's float32
            const float a = (float)act[3];
            reachRew += T((a > 0.0f ? a : 0.0f) / 50.0f);
        }
    } else reachRew = -reachDistxy;
    T pullRew = 0;
    if (reachDist < T(0.05)) pullRew = mw_max(T(1000) * (TK(e, V1_MAXA) - pullDist) + v1_bumps(pullDist), T(0));
    const T reward = reachRew + pullRew, d3 = norm(obj - goal);
    return Out{double(reward), double(d3 <= T(0.07)), make_info(1.0, 0.0, 0.0, 0.0, d3, reward)};
}
// ---- sweep (47), sweep-into (46): the object fallen off the table zeroes the terms AFTER reachCompleted was taken ----
template <typename T>
MW_HD Out sweep_eval_v1(const Env<T> e, const TaskDesc<T>& td, const T* obs) {
    const V3<T> obj = obs3(obs, 4), goal = tk3(e, TK_TARGET), oi = tk3(e, TK_OBJINIT);
    T reachDist = norm(obj - finger_com(e, td)), pushDist = v1_metric(M_XY, obj, goal), reachRew = -reachDist;
    const bool reached = reachDist < T(0.05);
    TK(e, V1_REACHED) = reached ? T(1) : T(0);
    if (obj.z < oi.z - T(0.05) && (td.kind == 47 || (T(0.4) < obj.y && obj.y < T(1.0)))) { reachRew = 0; reachDist = 0; pushDist = 0; }
    T pushRew = 0;
    if (reached) pushRew = mw_max(T(1000) * (TK(e, V1_MAXA) - pushDist) + v1_bumps(pushDist), T(0));
    const T reward = reachRew + pushRew;
    return Out{double(reward), double(pushDist <= T(0.05)), make_info(1.0, 0.0, 0.0, 0.0, pushDist, reward)};
}
// ---- hand-insert (17) ----
template <typename T>
MW_HD Out hand_insert_eval_v1(const Env<T> e, const TaskDesc<T>& td, const T* obs) {
    const V3<T> goal = tk3(e, TK_TARGET), finger = finger_com(e, td);
    const T reachDist = v1_metric(M_XY, finger, goal), dz = mw_abs(finger.z - goal.z);
    T near = 0;
    if (reachDist < T(0.05)) near = T(1000) * (TK(e, V1_MAXA) - dz) + v1_bumps(dz);
    near = mw_max(near, T(0));
    const T reward = -reachDist + near, d = norm(tk3(e, TK_OBJINIT) - goal);
    return Out{double(reward), double(d <= T(0.05)), make_info(1.0, 0.0, 0.0, 0.0, d, reward)};
}

// ---- pick-and-place family: reach (with a z penalty away from the object), pick (height bonus), place ---------------------
// pick-place (30), pick-place-wall (28), shelf-place (45), pick-out-of-hole (29), basketball (1), box-close (3),
// peg-insert-side (35).  Common core (e.g. sawyer_pick_place_v3.py v1 branch); per task: the z-penalty factor, the pick bonus
// of pick-out-of-hole, the distance the place term uses (peg-insert: the peg HEAD until it is within 5 cm), and what
// evaluate_state makes of the returned tuple.
template <typename T>
MW_HD T pick_core_v1(const Env<T> e, const TaskDesc<T>& td, V3<T> obj, V3<T> goal, const T* act, T zfac, bool hole, T placeDist) {
    const V3<T> finger = finger_com(e, td);
    const T heightTarget = TK(e, V1_HTARGET), objHeight = TK(e, V1_OBJH);
    const T reachDist = norm(obj - finger), placingDist = norm(obj - goal);
    const T reachDistxy = v1_metric(M_XY, obj, finger), zRew = mw_abs(finger.z - TK(e, TK_INITTCP + 2));
    T reachRew = reachDistxy < T(0.05) ? -reachDist : -reachDistxy - zfac * zRew;
    if (reachDist < T(0.05)) {          // incentive to close the fingers: max(action[-1], 0) / 50 in the action's float32
        const float a = (float)act[3];
        reachRew = -reachDist + T((a > 0.0f ? a : 0.0f) / 50.0f);
    }
    const bool picked = obj.z >= heightTarget - T(0.01);
    TK(e, V1_PICKED) = picked ? T(1) : T(0);
    const bool dropped = obj.z < objHeight + T(0.005) && placingDist > T(0.02) && reachDist > T(0.02);
    T pickRew = 0;
    if (picked && !dropped) pickRew = T(100) * (hole ? heightTarget - objHeight + T(0.02) : heightTarget);
    else if (reachDist < T(0.1) && obj.z > objHeight + T(0.005))
        pickRew = T(100) * (hole ? mw_min(heightTarget, obj.z) - objHeight + T(0.02) : mw_min(heightTarget, obj.z));
    T placeRew = 0;
    if (picked && reachDist < T(0.1) && !dropped) placeRew = mw_max(T(1000) * (TK(e, V1_MAXB) - placeDist) + v1_bumps(placeDist), T(0));
    return reachRew + pickRew + placeRew;
}
template <typename T>
MW_HD Out pick_eval_v1(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> obj = obs3(obs, 4), goal = tk3(e, TK_TARGET);
    const T placingDist = norm(obj - goal);
    if (td.kind == 3) {          // box-close: (reward, 0, 0, 0, success), success = |obj - target| < 0.08
        const T reward = pick_core_v1(e, td, obj, goal, act, T(2), false, placingDist);
        return Out{double(reward), double(placingDist < T(0.08)), make_info(0.0, 0.0, 0.0, 0.0, 0.0, reward)};
    }
    if (td.kind == 35) {         // peg-insert-side: the place term follows the peg head until it is within 5 cm of the goal
        const T head = norm(probe_pos(e, td.probe[P_X0]) - goal);
        const T reward = pick_core_v1(e, td, obj, goal, act, T(1), false, head <= T(0.05) ? placingDist : head);
        return Out{double(reward), double(placingDist <= T(0.07)), make_info(1.0, 0.0, 0.0, 0.0, placingDist, reward)};
    }
    const T reward = pick_core_v1(e, td, obj, goal, act, T(2), td.kind == 29, placingDist);
    const T thr = td.kind == 1 ? T(0.08) : T(0.07);
    return Out{double(reward), double(placingDist <= thr), make_info(1.0, 0.0, 0.0, 0.0, placingDist, reward)};
}
// ---- bin-picking (2): sawyer_bin_picking_v3.py v1 branch (placing measured in the xy plane; a placed object pays -200 a[3]) ----
template <typename T>
MW_HD Out bin_picking_eval_v1(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> obj = obs3(obs, 4), goal = tk3(e, TK_TARGET), finger = finger_com(e, td);
    const T heightTarget = TK(e, V1_HTARGET), objHeight = TK(e, V1_OBJH);
    const T reachDist = norm(obj - finger), placingDist = v1_metric(M_XY, obj, goal);
    const T reachDistxy = v1_metric(M_XY, obj, finger), zRew = mw_abs(finger.z - TK(e, TK_INITTCP + 2));
    T reachRew = reachDistxy < T(0.06) ? -reachDist : -reachDistxy - zRew;
    const float a = (float)act[3];
    if (reachDist < T(0.05)) reachRew = -reachDist + T((a > 0.0f ? a : 0.0f) / 50.0f);
    const bool picked = obj.z >= heightTarget - T(0.01);
    const bool dropped = obj.z < objHeight + T(0.005) && placingDist > T(0.02) && reachDist > T(0.02);
    const bool inbox = mw_abs(obj.x - goal.x) < T(0.05) && mw_abs(obj.y - goal.y) < T(0.05);
    const bool placed = inbox && obj.z < objHeight + T(0.05);
    TK(e, V1_PICKED) = picked ? T(1) : T(0); TK(e, V1_PLACED) = placed ? T(1) : T(0);
    T pickRew = 0;
    if (placed || (picked && !dropped)) pickRew = T(100) * heightTarget;
    else if (reachDist < T(0.1) && obj.z > objHeight + T(0.005)) pickRew = T(100) * mw_min(heightTarget, obj.z);
    T placeRew = mw_max(T(1000) * (TK(e, V1_MAXB) - placingDist) + v1_bumps(placingDist), T(0));
    const T grip = T(-200.0f * a);          // -200 * action[-1] in the action's float32
    T reward;
    if (placed) reward = grip + placeRew;
    else {
        if (picked && reachDist < T(0.1) && !dropped) placeRew = inbox ? grip + placeRew : placeRew;
        else placeRew = 0;
        reward = reachRew + pickRew + placeRew;
    }
    return Out{double(reward), double(placingDist <= T(0.05)), make_info(0.0, 0.0, 0.0, 0.0, placingDist, reward)};
}
// ---- peg-unplug-side (36) ----
template <typename T>
MW_HD Out peg_unplug_eval_v1(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> obj = obs3(obs, 4), goal = tk3(e, TK_TARGET), finger = finger_com(e, td);
    const T reachDist = norm(obj - finger), placingDist = v1_metric(M_XY, obj, goal);
    const T reachDistxy = v1_metric(M_XY, obj, finger), zRew = mw_abs(finger.z - td.hand_init[2]);
    T reachRew = reachDistxy < T(0.05) ? -reachDist : -reachDistxy - 2 * zRew;
    if (reachDist < T(0.05)) {
        const float a = (float)act[3];
        reachRew = -reachDist + T((a > 0.0f ? a : 0.0f) / 50.0f);
    }
    TK(e, V1_REACHED) = reachDist < T(0.05) ? T(1) : T(0);
    T placeRew = 0;
    if (reachDist < T(0.05)) placeRew = mw_max(T(1000) * (TK(e, V1_MAXB) - placingDist) + v1_bumps(placingDist), T(0));
    const T reward = reachRew + placeRew;
    return Out{double(reward), double(placingDist <= T(0.07)), make_info(1.0, 0.0, 0.0, 0.0, placingDist, reward)};
}

// ---- assembly (0): sawyer_assembly_peg_v3.py v1 branch ----
template <typename T>
MW_HD Out assembly_eval_v1(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> grasp = obs3(obs, 4), obj = probe_pos(e, td.probe[P_OBJ1]), goal = tk3(e, TK_TARGET), finger = finger_com(e, td);
    const T heightTarget = TK(e, V1_HTARGET), objHeight = TK(e, V1_OBJH);
    const T reachDist = norm(grasp - finger), placingDist = v1_metric(M_XY, obj, goal), placingDistFinal = mw_abs(obj.z - objHeight);
    const T reachDistxy = v1_metric(M_XY, grasp, finger), zRew = mw_abs(finger.z - TK(e, TK_INITTCP + 2));
    T reachRew = reachDistxy < T(0.04) ? -reachDist : -reachDistxy - zRew;
    if (reachDist < T(0.04)) {
        const float a = (float)act[3];
        reachRew = -reachDist + T((a > 0.0f ? a : 0.0f) / 50.0f);
    }
    const bool picked = obj.z >= heightTarget - T(0.01) && reachDist < T(0.03);
    const bool dropped = obj.z < objHeight + T(0.005) && placingDist > T(0.02) && reachDist > T(0.02);
    const bool placed = mw_abs(obj.x - goal.x) < T(0.03) && mw_abs(obj.y - goal.y) < T(0.03);
    TK(e, V1_PICKED) = picked ? T(1) : T(0); TK(e, V1_PLACED) = placed ? T(1) : T(0);
    T pickRew = 0;
    if (placed || (picked && !dropped)) pickRew = T(100) * heightTarget;
    else if (reachDist < T(0.04) && obj.z > objHeight + T(0.005)) pickRew = T(100) * mw_min(heightTarget, obj.z);
    T placeRew = T(1000) * (TK(e, V1_MAXB) - placingDist) + v1_bumps(placingDist);
    if (placed) {
        const T d = placingDistFinal;
        placeRew += T(2000) * (heightTarget - d) + T(2000) * (exp(-(d * d) / T(0.003)) + exp(-(d * d) / T(0.0003)));
    }
    placeRew = mw_max(placeRew, T(0));
    if (!(placed || (picked && reachDist < T(0.04) && !dropped))) placeRew = 0;
    const T reward = reachRew + pickRew + placeRew;
    const bool success = placed && placingDistFinal <= T(0.04);
    return Out{double(reward), double(success), make_info(0.0, 0.0, 0.0, 0.0, 0.0, reward)};
}
// ---- disassemble (12) ----
template <typename T>
MW_HD Out disassemble_eval_v1(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> obj = obs3(obs, 4), goal = tk3(e, TK_TARGET), finger = finger_com(e, td);
    const T heightTarget = TK(e, V1_HTARGET), objHeight = TK(e, V1_OBJH);
    T reachDist = norm(obj - finger), placingDist = norm(obj - goal);
    const T reachDistxy = v1_metric(M_XY, obj, finger), zDist = mw_abs(finger.z - TK(e, TK_INITTCP + 2));
    T reachRew = reachDistxy < T(0.04) ? -reachDist : -reachDistxy - 2 * zDist;
    if (reachDist < T(0.04)) {
        const float a = (float)act[3];
        reachRew = -reachDist + T((a > 0.0f ? a : 0.0f) / 50.0f);
    }
    const bool picked = obj.z >= heightTarget - T(0.01) && reachDist < T(0.04);
    TK(e, V1_PICKED) = picked ? T(1) : T(0);
    const bool dropped = obj.z < objHeight + T(0.005) && placingDist > T(0.02) && reachDist > T(0.02);
    T pickRew = 0;
    if (picked && !dropped) pickRew = T(100) * heightTarget;
    else if (reachDist < T(0.04) && obj.z > objHeight + T(0.005)) pickRew = T(100) * mw_min(heightTarget, obj.z);
    T placeRew = mw_max(T(1000) * (TK(e, V1_MAXB) - placingDist) + v1_bumps(placingDist), T(0));
    if (!(picked && reachDist < T(0.03) && !dropped)) placeRew = 0;
    const V3<T> peg = ld3(e, e.lay().reloc + 3 * td.reloc[0]), nut = probe_pos(e, td.probe[P_OBJ1]);      // model.body("peg").pos, get_body_com("RoundNut")
    if (mw_abs(nut.x - peg.x) > T(0.05) || mw_abs(nut.y - peg.y) > T(0.05)) { reachRew = 0; pickRew = heightTarget * T(100); }
    const T reward = reachRew + pickRew + placeRew;
    return Out{double(reward), double(obs[6] > goal.z), make_info(0.0, 0.0, 0.0, 0.0, 0.0, reward)};
}

// ---- hammer (22): sawyer_hammer_v3.py v1 branch; P_X1 = geom HammerHead, P_X2 = site nailHead (extra_v1 probes) ----
template <typename T>
MW_HD Out hammer_eval_v1(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> hammer = obs3(obs, 4), head = probe_pos(e, td.probe[P_X1]), nail = probe_pos(e, td.probe[P_X2]);
    const V3<T> finger = finger_com(e, td), goal = tk3(e, TK_TARGET);
    const T heightTarget = TK(e, V1_HTARGET), hammerHeight = TK(e, V1_OBJH);
    const T hammerDist = norm(nail - head), screwDist = mw_abs(nail.y - goal.y), reachDist = norm(hammer - finger);
    T reachRew = -reachDist;
    if (reachDist < T(0.05)) {
        const float a = (float)act[3];
        reachRew = -reachDist + T((a > 0.0f ? a : 0.0f) / 50.0f);
    }
    const bool picked = hammer.z >= heightTarget - T(0.01);
    TK(e, V1_PICKED) = picked ? T(1) : T(0);
    const bool dropped = hammer.z < hammerHeight + T(0.005) && hammerDist > T(0.02) && reachDist > T(0.02);
    T pickRew = 0;
    if (picked && !dropped) pickRew = T(100) * heightTarget;
    else if (reachDist < T(0.1) && hammer.z > hammerHeight + T(0.005)) pickRew = T(100) * mw_min(heightTarget, hammer.z);
    T hammerRew = 0;
    if (picked && reachDist < T(0.1) && !dropped) {
        const T d = hammerDist + screwDist;
        hammerRew = mw_max(T(1000) * (TK(e, V1_MAXB) - hammerDist - screwDist) + v1_bumps(d), T(0));
    }
    const T reward = reachRew + pickRew + hammerRew;
    const bool success = e.R(e.lay().qpos + td.qadr[0]) > T(0.09);        // NailSlideJoint
    return Out{double(reward), double(success), make_info(0.0, 0.0, 0.0, 0.0, 0.0, reward)};
}
// ---- stick-push (38), stick-pull (39): `objPos = obs[6:9]` (sic: the stick's z and the first two components of its
// quaternion) is what the reference computes with ----
template <typename T>
MW_HD Out stick_eval_v1(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act) {
    const V3<T> stick = obs3(obs, 4), objp = obs3(obs, 6), goal = tk3(e, TK_TARGET), finger = finger_com(e, td);
    const T heightTarget = TK(e, V1_HTARGET), stickHeight = TK(e, V1_OBJH);
    const T moveDist = mw_sqrt((objp.x - goal.x) * (objp.x - goal.x) + (objp.y - goal.y) * (objp.y - goal.y));
    const T placeDist = td.kind == 39 ? norm(stick - objp) : norm(objp - stick), reachDist = norm(stick - finger);
    T reachRew = -reachDist;
    if (reachDist < T(0.05)) {
        const float a = (float)act[3];
        reachRew = -reachDist + T((a > 0.0f ? a : 0.0f) / 50.0f);
    }
    const bool picked = stick.z >= heightTarget - T(0.01);
    TK(e, V1_PICKED) = picked ? T(1) : T(0);
    const bool dropped = stick.z < stickHeight + T(0.005) && moveDist > T(0.02) && reachDist > T(0.02);
    T pickRew = 0;
    if (picked && !dropped) pickRew = T(100) * heightTarget;
    else if (reachDist < T(0.1) && stick.z > stickHeight + T(0.005)) pickRew = T(100) * mw_min(heightTarget, stick.z);
    T moveRew = 0;
    if (picked && reachDist < T(0.1) && !dropped) {
        moveRew = T(1000) * (TK(e, V1_MAXB) - placeDist) + v1_bumps(placeDist);
        if (placeDist < T(0.05)) {
            const T c5 = td.kind == 39 ? T(0.01) : T(0.001), c6 = td.kind == 39 ? T(0.001) : T(0.0001);
            moveRew += T(1000) * (TK(e, V1_MAXA) - moveDist) + T(2000) * (exp(-(moveDist * moveDist) / c5) + exp(-(moveDist * moveDist) / c6));
        }
        moveRew = mw_max(moveRew, T(0));
    }
    const T reward = reachRew + pickRew + moveRew;
    double success = 0.0;          // stick-push: `grasp_success and success` with grasp_success = 0.0 -> 0.0
    if (td.kind == 39) {
        const V3<T> handle = obs3(obs, 11), end = probe_pos(e, td.probe[P_X1]);
        const bool inserted = end.x >= handle.x && mw_abs(end.y - handle.y) <= T(0.04) && mw_abs(end.z - handle.z) <= T(0.06);
        success = double(norm(handle - goal) <= T(0.12) && inserted);
    }
    return Out{double(reward), success, make_info(1.0, 0.0, 0.0, 0.0, moveDist, reward)};
}

// objHeight / heightTarget / maxPlacingDist of the family (end of reset_model)
template <typename T>
MW_HD void pick_reset_v1(const Env<T> e, const TaskDesc<T>& td, T objHeight, T lift, V3<T> target) {
    const V3<T> oi = tk3(e, TK_OBJINIT);
    const T ht = objHeight + lift;
    TK(e, V1_OBJH) = objHeight; TK(e, V1_HTARGET) = ht;
    TK(e, V1_MAXB) = norm(v3(oi.x, oi.y, ht) - target) + ht;
}

// the `max...` attribute the family's reset_model leaves on self
template <typename T>
MW_HD T fixture_max_v1(const Env<T> e, const TaskDesc<T>& td) {
    const V3<T> t = tk3(e, TK_TARGET), oi = tk3(e, TK_OBJINIT);
    switch (td.kind) {
    case 4: case 5: case 7: return mw_abs(probe_pos(e, td.probe[P_X1]).z - t.z);      // buttonStart (button-press-wall: z, sic)
    case 6: return mw_abs(probe_pos(e, td.probe[P_X1]).y - t.y);
    case 8: return mw_abs(probe_pos(e, td.probe[P_OBJ0]).y - t.y);                    // coffee-button: site buttonStart is the object
    case 11: return mw_abs(t.y - oi.y);
    case 13: case 15: case 33: case 34: return v1_metric(M_XY, probe_pos(e, td.probe[P_OBJ0]), t);   // geom handle / puck xpos[:2]
    case 14: case 16: case 20: case 21: case 27: return norm(t - oi);
    case 18: return T(0.15);
    case 19: case 48: case 49: return T(0.2);
    case 23: case 24: return mw_abs(probe_pos(e, td.probe[P_OBJ0]).z - t.z);          // site handleStart (world)
    case 25: return mw_abs(probe_pos(e, td.probe[P_X1]).z - t.z);                     // handle-pull-side: site handleStart (an extra_v1 probe)
    case 26: return mw_abs(td.c[6] - t.z);                                            // handle-pull: model.site("handleStart").pos[-1] (LOCAL, sic)
    case 31: case 32: case 10: case 37: case 40: case 41: case 42: return v1_metric(M_XY, oi, t);   // obj_init_pos[:2] vs target[:2]
    default: return T(0);
    }
}

template <typename T>
MW_HD void task_reset_v1(const Env<T> e, const TaskDesc<T>& td) {
    for (int k = TK_V1; k < V1_END; k++) TK(e, k) = 0;
    const V3<T> target = tk3(e, TK_TARGET), init_tcp = tk3(e, TK_INITTCP);
    switch (td.kind) {
    case 43: case 44: TK(e, V1_MAXA) = norm(init_tcp - target); break;      // maxReachDist
    case 9: case 46: case 47: TK(e, V1_MAXA) = v1_metric(M_XY, tk3(e, TK_OBJINIT), target); break;   // maxPullDist / maxPushDist
    case 17: TK(e, V1_MAXA) = mw_abs(td.hand_init[2] - target.z); break;   // maxReachDist = |hand_init_pos[-1] - target[-1]|
    case 30: case 28: case 45: pick_reset_v1(e, td, probe_pos(e, td.probe[P_OBJ1]).z, T(0.04), target); break;   // geom objGeom z
    case 29: pick_reset_v1(e, td, probe_pos(e, td.probe[P_X0]).z, T(0.11), target); break;     // pick-out-of-hole: geom objGeom (extra_v1)
    case 3: pick_reset_v1(e, td, probe_pos(e, td.probe[P_X0]).z, T(0.12), target); break;      // box-close: geom BoxHandleGeom (extra_v1)
    case 35: pick_reset_v1(e, td, e.R(e.lay().qpos + 11), T(0.11), target); break;            // peg-insert-side: get_body_com("peg")[2] = the free joint's z
    case 2: {   // bin-picking: objHeight = data.body("obj").xpos[2], maxPlacingDist = |obj_init[:2] - target[:2]| + heightTarget
        const T oh = probe_pos(e, td.probe[P_OBJ0]).z, ht = oh + T(0.1);
        TK(e, V1_OBJH) = oh; TK(e, V1_HTARGET) = ht; TK(e, V1_MAXB) = v1_metric(M_XY, tk3(e, TK_OBJINIT), target) + ht;
        break;
    }
    case 36: TK(e, V1_MAXB) = norm(target - tk3(e, TK_OBJINIT)); break;      // peg-unplug-side: maxPlacingDist
    case 22: {   // hammer: maxHammerDist = |(hi.x, hi.y, heightTarget) - obj_init| + heightTarget + |obj_init.y - target.y| (hammer_init_pos IS obj_init_pos)
        const V3<T> oi = tk3(e, TK_OBJINIT);
        const T hh = probe_pos(e, td.probe[P_OBJ0]).z, ht = hh + T(0.09);
        TK(e, V1_OBJH) = hh; TK(e, V1_HTARGET) = ht;
        TK(e, V1_MAXB) = norm(v3(oi.x, oi.y, ht) - oi) + ht + mw_abs(oi.y - target.y);
        break;
    }
    case 38: case 39: {   // stick tasks: stickHeight = body stick z; maxPlaceDist vs stick_init_pos; maxPushDist / maxPullDist in the xy plane
        const V3<T> oi = tk3(e, TK_OBJINIT), si = tk3(e, TK_EXTRA);
        const T sh = probe_pos(e, td.probe[P_OBJ0]).z, ht = sh + T(0.04);
        TK(e, V1_OBJH) = sh; TK(e, V1_HTARGET) = ht;
        TK(e, V1_MAXB) = norm(v3(oi.x, oi.y, ht) - si) + ht;
        TK(e, V1_MAXA) = v1_metric(M_XY, oi, target);
        break;
    }
    case 0: pick_reset_v1(e, td, probe_pos(e, td.probe[P_OBJ0]).z, T(0.1), target); break;       // assembly: obj_height = site RoundNut-8 z
    case 12: pick_reset_v1(e, td, probe_pos(e, td.probe[P_OBJ1]).z, T(0.05), target); break;     // disassemble: objHeight = body RoundNut z
    case 1: {   // basketball: at that point of reset_model `_target_pos` is the goal site's xpos of the LAST forward (task_after_reset's s)
        const V3<T> P = tk3(e, TK_EXTRA), sgoal = tk3(e, TK_PERSIST0) + P * T(2);
        pick_reset_v1(e, td, probe_pos(e, td.probe[P_X1]).z, T(0.3), sgoal);
        break;
    }
    default:
        if (v1_fixture(td.kind).kind >= 0) TK(e, V1_MAXA) = fixture_max_v1(e, td);
        break;
    }
}

template <typename T>
MW_HD void task_evaluate_v1(const Env<T> e, const TaskDesc<T>& td, const T* obs, const T* act, T* reward, T* success, Info* info) {
    Out o{0, 0, Info{0, 0, 0, 0, 0, 0}};
    switch (td.kind) {
    case 43: case 44: o = reach_eval_v1(e, td, obs, act); break;
    case 9: o = coffee_pull_eval_v1(e, td, obs, act); break;
    case 46: case 47: o = sweep_eval_v1(e, td, obs); break;
    case 17: o = hand_insert_eval_v1(e, td, obs); break;
    case 2: o = bin_picking_eval_v1(e, td, obs, act); break;
    case 36: o = peg_unplug_eval_v1(e, td, obs, act); break;
    case 0: o = assembly_eval_v1(e, td, obs, act); break;
    case 12: o = disassemble_eval_v1(e, td, obs, act); break;
    case 22: o = hammer_eval_v1(e, td, obs, act); break;
    case 38: case 39: o = stick_eval_v1(e, td, obs, act); break;
    case 1: case 3: case 28: case 29: case 30: case 35: case 45: o = pick_eval_v1(e, td, obs, act); break;
    default: {
        const V1Fixture f = v1_fixture(td.kind);
        if (f.kind >= 0) { o = fixture_eval_v1(e, td, f, obs); break; }
        task_evaluate(e, td, obs, act, reward, success, info); return;
    }
    }
    *reward = T(o.reward); *success = T(o.success); *info = o.info;
}

}  // namespace mw
