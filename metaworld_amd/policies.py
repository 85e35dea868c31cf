"""Batched scripted policies (numpy, [N, 39] observations -> float32 [N, 4] actions) for the MT10 task set.

The reference's scripted policies (metaworld/policies/sawyer_*_v3_policy.py) are per-environment Python: a cascade of
geometric conditions picks a waypoint, the action is `p * (waypoint - hand)` plus a gripper effort, stored as float32
(metaworld/policies/action.py:22-24).  The same cascades are written here as first-match selects over the batch, so a
whole vector env can be driven closed loop without the reference installed (e.g. on the GPU box; BASELINE config 5
style success-rate runs).  tests/test_batched_policies.py checks them against the reference policies, action for
action, on the closed-loop traces of tests/golden/policy_*.npz.  Each function cites the reference file it restates.
"""
from __future__ import annotations

import numpy as np


def _norm(v):
    return np.sqrt((v * v).sum(axis=1))


def _first(conds, vals):
    """rows: vals[i] where conds[i] is the first true condition, else vals[-1]"""
    out = np.array(vals[-1], dtype=np.float64, copy=True)
    for c, v in zip(reversed(conds), reversed(vals[:-1])):
        out = np.where(c[:, None] if out.ndim == 2 else c, v, out)
    return out


def _off(p, dx, dy, dz):
    return p + np.array([dx, dy, dz])


def _act(hand, to, p, grab):
    a = np.zeros((len(hand), 4), dtype=np.float32)
    a[:, :3] = p * (to - hand)
    a[:, 3] = grab
    return a


def _parts(obs):
    obs = np.asarray(obs, dtype=np.float64)
    return obs[:, 0:3], obs[:, 3], obs[:, 4:7], obs[:, 36:39]


def reach(obs):                      # sawyer_reach_v3_policy.py
    h, g, o, goal = _parts(obs)
    return _act(h, goal, 5.0, 0.0)


def push(obs):                       # sawyer_push_v3_policy.py
    h, g, o, goal = _parts(obs)
    puck = _off(o, -0.005, 0, 0)
    far_xy = _norm(h[:, :2] - puck[:, :2]) > 0.02
    to = _first([far_xy, np.abs(h[:, 2] - puck[:, 2]) > 0.04], [_off(puck, 0, 0, 0.2), _off(puck, 0, 0, 0.03), goal])
    grab = np.where((_norm(h[:, :2] - o[:, :2]) > 0.02) | (np.abs(h[:, 2] - o[:, 2]) > 0.10), 0.0, 0.6)
    return _act(h, to, 10.0, grab)


def pick_place(obs):                 # sawyer_pick_place_v3_policy.py
    h, g, o, goal = _parts(obs)
    puck = _off(o, -0.005, 0, 0)
    to = _first([_norm(h[:, :2] - puck[:, :2]) > 0.02, (np.abs(h[:, 2] - puck[:, 2]) > 0.05) & (puck[:, 2] < 0.04), g > 0.73],
                [_off(puck, 0, 0, 0.1), _off(puck, 0, 0, 0.03), h, goal])
    grab = np.where(_norm(h - o) < 0.07, 1.0, 0.0)
    return _act(h, to, 10.0, grab)


def door_open(obs):                  # sawyer_door_open_v3_policy.py (the reference shifts door x by -0.05 in place)
    h, g, o, goal = _parts(obs)
    door = _off(o, -0.05, 0, 0)
    to = _first([_norm(h[:, :2] - door[:, :2]) > 0.12, np.abs(h[:, 2] - door[:, 2]) > 0.04],
                [_off(door, 0.06, 0.02, 0.2), _off(door, 0.06, 0.02, 0.0), door])
    return _act(h, to, 25.0, 1.0)


def drawer_open(obs):                # sawyer_drawer_open_v3_policy.py (gain 4 while approaching, 50 while pulling)
    h, g, o, goal = _parts(obs)
    d = _off(o, 0, 0, -0.02)
    c1, c2 = _norm(h[:, :2] - d[:, :2]) > 0.06, np.abs(h[:, 2] - d[:, 2]) > 0.04
    to = _first([c1, c2], [_off(d, 0, 0, 0.3), d, _off(d, 0, -0.06, 0)])
    p = np.where(c1 | c2, 4.0, 50.0)[:, None]
    return _act(h, to, p, -1.0)


def drawer_close(obs):               # sawyer_drawer_close_v3_policy.py
    h, g, o, goal = _parts(obs)
    d = _off(o, 0, 0, -0.02)
    behind = h[:, 1] > d[:, 1]
    up = np.stack([h[:, 0], h[:, 1], d[:, 2] + 0.5], axis=1)
    to = _first([behind & (h[:, 2] < d[:, 2] + 0.23), behind, np.abs(h[:, 2] - d[:, 2]) > 0.04],
                [up, _off(d, 0, -0.075, 0.23), _off(d, 0, -0.075, 0), d])
    return _act(h, to, 25.0, 1.0)


def button_press_topdown(obs):       # sawyer_button_press_topdown_v3_policy.py
    h, g, o, goal = _parts(obs)
    to = _first([_norm(h[:, :2] - o[:, :2]) > 0.04], [_off(o, 0, 0, 0.1), o])
    return _act(h, to, 25.0, 1.0)


def peg_insert_side(obs):            # sawyer_peg_insertion_side_v3_policy.py
    h, g, o, goal = _parts(obs)
    hole = np.stack([np.full(len(h), -0.35), goal[:, 1], np.full(len(h), 0.16)], axis=1)
    to = _first([_norm(h[:, :2] - o[:, :2]) > 0.04, np.abs(h[:, 2] - o[:, 2]) > 0.025, _norm(o[:, 1:] - hole[:, 1:]) > 0.03],
                [_off(o, 0, 0, 0.3), o, _off(hole, 0.4, 0, 0), hole])
    grab = np.where((_norm(h[:, :2] - o[:, :2]) > 0.04) | (np.abs(h[:, 2] - o[:, 2]) > 0.15), -1.0, 0.6)
    return _act(h, to, 25.0, grab)


def window_open(obs):                # sawyer_window_open_v3_policy.py
    h, g, o, goal = _parts(obs)
    w = _off(o, -0.03, -0.03, -0.08)
    to = _first([_norm(h[:, :2] - w[:, :2]) > 0.04, np.abs(h[:, 2] - w[:, 2]) > 0.02], [_off(w, 0, 0, 0.3), w, _off(w, 0.1, 0, 0)])
    return _act(h, to, 25.0, 1.0)


def window_close(obs):               # sawyer_window_close_v3_policy.py
    h, g, o, goal = _parts(obs)
    w = _off(o, 0.03, -0.03, -0.08)
    to = _first([_norm(h[:, :2] - w[:, :2]) > 0.04, np.abs(h[:, 2] - w[:, 2]) > 0.02], [_off(w, 0, 0, 0.25), w, _off(w, -0.1, 0, 0)])
    return _act(h, to, 25.0, 1.0)


POLICIES = {"reach-v3": reach, "push-v3": push, "pick-place-v3": pick_place, "door-open-v3": door_open,
            "drawer-open-v3": drawer_open, "drawer-close-v3": drawer_close, "button-press-topdown-v3": button_press_topdown,
            "peg-insert-side-v3": peg_insert_side, "window-open-v3": window_open, "window-close-v3": window_close}


def batched_actions(task_names, obs):
    """float32 [N, 4] actions (clipped to [-1, 1] like the env would) for a vector env whose env i runs `task_names[i]`."""
    obs = np.asarray(obs)
    names = np.asarray(task_names)
    out = np.zeros((len(obs), 4), dtype=np.float32)
    for n in np.unique(names):
        idx = np.flatnonzero(names == n)
        out[idx] = POLICIES[n](obs[idx, :39])
    return np.clip(out, -1, 1)
