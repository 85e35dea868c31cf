"""Batched scripted policies (numpy, [N, 39] observations -> float32 [N, 4] actions) for all 50 v3 tasks.

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


def _xyz(x, y, z, n):
    """[n, 3] from per-env arrays or scalars"""
    return np.stack([np.broadcast_to(np.asarray(v, dtype=np.float64), (n,)) for v in (x, y, z)], axis=1)


def assembly(obs):                   # sawyer_assembly_v3_policy.py
    h, g, o, goal = _parts(obs)
    w, peg = _off(o, -0.02, 0, 0), _off(goal, 0.12, 0, 0.14)
    to = _first([_norm(h[:, :2] - w[:, :2]) > 0.02, _norm(h[:, :2] - peg[:, :2]) <= 0.02, np.abs(h[:, 2] - w[:, 2]) > 0.05,
                 np.abs(h[:, 2] - peg[:, 2]) > 0.04],
                [_off(w, 0, 0, 0.1), _off(peg, 0, 0, -0.2), _off(w, 0, 0, 0.03), _xyz(h[:, 0], h[:, 1], peg[:, 2], len(h)), peg])
    grab = np.where((_norm(h[:, :2] - w[:, :2]) > 0.02) | (np.abs(h[:, 2] - w[:, 2]) > 0.12), 0.0, 0.6)
    return _act(h, to, 10.0, grab)


def basketball(obs):                 # sawyer_basketball_v3_policy.py
    h, g, o, goal = _parts(obs)
    ball, hoop = _off(o, 0, 0, 0.01), _xyz(goal[:, 0], 0.875, 0.35, len(h))
    to = _first([_norm(h[:, :2] - ball[:, :2]) > 0.04, np.abs(h[:, 2] - ball[:, 2]) > 0.025, np.abs(ball[:, 2] - hoop[:, 2]) > 0.025],
                [_off(ball, 0, 0, 0.3), ball, _xyz(h[:, 0], h[:, 1], hoop[:, 2], len(h)), hoop])
    grab = np.where((_norm(h[:, :2] - o[:, :2]) > 0.04) | (np.abs(h[:, 2] - o[:, 2]) > 0.15), -1.0, 0.6)
    return _act(h, to, 25.0, grab)


def bin_picking(obs):                # sawyer_bin_picking_v3_policy.py
    h, g, o, goal = _parts(obs)
    cube = _off(o, 0, 0, 0.03)
    cube[:, 1] = np.maximum(0.675, np.minimum(cube[:, 1], 0.725))
    bin_ = _xyz(0.12, 0.7, 0.02, len(h))
    far_bin = _norm(h[:, :2] - bin_[:, :2]) > 0.02
    to = _first([_norm(h[:, :2] - cube[:, :2]) > 0.02, np.abs(h[:, 2] - cube[:, 2]) > 0.01, far_bin & (h[:, 2] < 0.15), far_bin],
                [_off(cube, 0, 0, 0.15), cube, _off(h, 0, 0, 0.1), _xyz(0.12, 0.7, 0.18, len(h)), bin_])
    grab = np.where((_norm(h[:, :2] - cube[:, :2]) > 0.02) | (np.abs(h[:, 2] - cube[:, 2]) > 0.02), -1.0, 0.6)
    return _act(h, to, 25.0, grab)


def box_close(obs):                  # sawyer_box_close_v3_policy.py
    h, g, o, goal = _parts(obs)
    lid, box = _off(o, 0, 0, 0.02), _xyz(goal[:, 0], goal[:, 1], 0.15, len(h))
    to = _first([_norm(h[:, :2] - lid[:, :2]) > 0.01, np.abs(h[:, 2] - lid[:, 2]) > 0.05, np.abs(h[:, 2] - box[:, 2]) > 0.04],
                [_xyz(lid[:, 0], lid[:, 1], 0.2, len(h)), lid, _xyz(h[:, 0], h[:, 1], box[:, 2], len(h)), box])
    grab = np.where((_norm(h[:, :2] - lid[:, :2]) > 0.01) | (np.abs(h[:, 2] - lid[:, 2]) > 0.13), 0.5, 1.0)
    return _act(h, to, 25.0, grab)


def button_press_topdown_wall(obs):  # sawyer_button_press_topdown_wall_v3_policy.py
    h, g, o, goal = _parts(obs)
    b = _off(o, 0, -0.06, 0)
    to = _first([_norm(h[:, :2] - b[:, :2]) > 0.04], [_off(b, 0, 0, 0.1), b])
    return _act(h, to, 25.0, -1.0)


def button_press(obs):               # sawyer_button_press_v3_policy.py (np.isclose(atol=0.02) also has its default rtol 1e-5)
    h, g, o, goal = _parts(obs)
    b = _off(o, 0, 0, -0.07)
    close = (np.abs(h[:, 0] - b[:, 0]) <= 0.02 + 1e-5 * np.abs(b[:, 0])) & (np.abs(h[:, 2] - b[:, 2]) <= 0.02 + 1e-5 * np.abs(b[:, 2]))
    to = _first([~close], [_xyz(b[:, 0], h[:, 1] - 0.1, b[:, 2], len(h)), _off(b, 0, 0.02, 0)])
    return _act(h, to, 25.0, 0.0)


def button_press_wall(obs):          # sawyer_button_press_wall_v3_policy.py
    h, g, o, goal = _parts(obs)
    b = _off(o, 0, 0, 0.04)
    c1, c2, c3 = np.abs(h[:, 0] - b[:, 0]) > 0.02, b[:, 1] - h[:, 1] > 0.09, np.abs(h[:, 2] - b[:, 2]) > 0.02
    to = _first([c1, c2, c3], [_xyz(b[:, 0], h[:, 1], 0.3, len(h)), _xyz(b[:, 0], b[:, 1], 0.3, len(h)), _off(b, 0, -0.05, 0), _off(b, 0, -0.02, 0)])
    return _act(h, to, 15.0, np.where(c1 | c2 | c3, 1.0, -1.0))


def coffee_button(obs):              # sawyer_coffee_button_v3_policy.py
    h, g, o, goal = _parts(obs)
    b = _off(o, 0, 0, -0.07)
    to = _first([_norm(h[:, [0, 2]] - b[:, [0, 2]]) > 0.02], [_xyz(b[:, 0], h[:, 1], b[:, 2], len(h)), _off(b, 0, 0.2, 0)])
    return _act(h, to, 10.0, -1.0)



def coffee_pull(obs):                # sawyer_coffee_pull_v3_policy.py
    h, g, o, goal = _parts(obs)
    mug, mug_g = _off(o, -0.005, 0, 0.05), _off(o, 0.01, 0, 0.05)
    to = _first([_norm(h[:, :2] - mug[:, :2]) > 0.06, np.abs(h[:, 2] - mug[:, 2]) > 0.02], [_off(mug, 0, 0, 0.15), mug, goal])
    grab = np.where((_norm(h[:, :2] - mug_g[:, :2]) > 0.06) | (np.abs(h[:, 2] - mug_g[:, 2]) > 0.1), -1.0, 0.7)
    return _act(h, to, 10.0, grab)


def coffee_push(obs):                # sawyer_coffee_push_v3_policy.py
    h, g, o, goal = _parts(obs)
    mug = _off(o, 0.01, 0, 0.05)
    to = _first([_norm(h[:, :2] - mug[:, :2]) > 0.06, np.abs(h[:, 2] - mug[:, 2]) > 0.02],
                [_off(mug, 0, 0, 0.2), mug, _xyz(goal[:, 0], goal[:, 1], 0.1, len(h))])
    grab = np.where((_norm(h[:, :2] - mug[:, :2]) > 0.06) | (np.abs(h[:, 2] - mug[:, 2]) > 0.1), -1.0, 0.5)
    return _act(h, to, 10.0, grab)


def dial_turn(obs):                  # sawyer_dial_turn_v3_policy.py
    h, g, o, goal = _parts(obs)
    d = _off(o, 0.05, 0.02, 0.09)
    to = _first([_norm(h[:, :2] - d[:, :2]) > 0.02, np.abs(h[:, 2] - d[:, 2]) > 0.02],
                [_xyz(d[:, 0], d[:, 1], 0.2, len(h)), d, _off(d, -0.05, 0.005, 0)])
    return _act(h, to, 10.0, 1.0)


def disassemble(obs):                # sawyer_disassemble_v3_policy.py
    h, g, o, goal = _parts(obs)
    w = _off(o, -0.02, 0, 0.01)
    to = _first([_norm(h[:, :2] - w[:, :2]) > 0.02, np.abs(h[:, 2] - w[:, 2]) > 0.03], [_off(w, 0, 0, 0.1), w, _off(h, 0, 0, 0.1)])
    grab = np.where((_norm(h[:, :2] - w[:, :2]) > 0.02) | (np.abs(h[:, 2] - w[:, 2]) > 0.07), 0.0, 0.8)
    return _act(h, to, 10.0, grab)


def door_close(obs):                 # sawyer_door_close_v3_policy.py (the reference shifts the door position in place)
    h, g, o, goal = _parts(obs)
    d = _off(o, 0.05, 0.12, 0.1)
    right = h[:, 0] > d[:, 0]
    to = _first([right & (h[:, 2] < d[:, 2] + 0.2), right, np.abs(h[:, 2] - d[:, 2]) > 0.04],
                [_xyz(h[:, 0], h[:, 1], d[:, 2] + 0.25, len(h)), _xyz(d[:, 0] - 0.02, d[:, 1], h[:, 2], len(h)), _off(d, -0.02, 0, 0), goal])
    return _act(h, to, 25.0, 1.0)


def door_lock(obs):                  # sawyer_door_lock_v3_policy.py
    h, g, o, goal = _parts(obs)
    k = _off(o, -0.02, -0.02, 0)
    far = _norm(h[:, :2] - k[:, :2]) > 0.02
    to = _first([far & (h[:, 2] < 0.25), far, np.abs(h[:, 2] - k[:, 2]) > 0.02],
                [_off(h, 0, -0.1, 0.1), _off(k, 0, 0, 0.3), k, _off(k, -0.1, 0, -0.1)])
    return _act(h, to, 25.0, -1.0)


def door_unlock(obs):                # sawyer_door_unlock_v3_policy.py
    h, g, o, goal = _parts(obs)
    k = _off(o, -0.04, -0.02, -0.03)
    far = _norm(h[:, :2] - k[:, :2]) > 0.02
    to = _first([far & (h[:, 2] > 0.15), far], [_off(h, 0, -0.1, -0.1), k, _off(k, 0.1, 0, 0.01)])
    return _act(h, to, 25.0, 1.0)


def _faucet(obs, dx, push_x):        # sawyer_faucet_close_v3_policy.py / sawyer_faucet_open_v3_policy.py
    h, g, o, goal = _parts(obs)
    f = _off(o, dx, 0, 0.03)
    to = _first([_norm(h[:, :2] - f[:, :2]) > 0.04, np.abs(h[:, 2] - f[:, 2]) > 0.04], [_off(f, 0, 0, 0.1), f, _off(f, push_x, 0.05, 0)])
    return _act(h, to, 25.0, 1.0)


def faucet_close(obs):
    return _faucet(obs, 0.04, -0.1)


def faucet_open(obs):
    return _faucet(obs, -0.04, 0.1)


def hammer(obs):                     # sawyer_hammer_v3_policy.py
    h, g, o, goal = _parts(obs)
    p = _off(o, -0.04, 0, -0.01)
    tgt = np.array([0.24, 0.71, 0.11]) + np.array([-0.19, 0.0, 0.05])
    tg = np.tile(tgt, (len(h), 1))
    to = _first([_norm(h[:, :2] - p[:, :2]) > 0.04, (np.abs(h[:, 2] - p[:, 2]) > 0.05) & (p[:, 2] < 0.03),
                 _norm(h[:, [0, 2]] - tg[:, [0, 2]]) > 0.02],
                [_off(p, 0, 0, 0.1), _off(p, 0, 0, 0.03), _xyz(tgt[0], h[:, 1], tgt[2], len(h)), tg])
    grab = np.where((_norm(h[:, :2] - p[:, :2]) > 0.04) | (np.abs(h[:, 2] - p[:, 2]) > 0.1), 0.0, 0.8)
    return _act(h, to, 10.0, grab)


def hand_insert(obs):                # sawyer_hand_insert_v3_policy.py
    h, g, o, goal = _parts(obs)
    to = _first([_norm(h[:, :2] - o[:, :2]) > 0.02, np.abs(h[:, 2] - o[:, 2]) > 0.05, _norm(h[:, :2] - goal[:, :2]) > 0.04],
                [_off(o, 0, 0, 0.1), _off(o, 0, 0, 0.03), _xyz(goal[:, 0], goal[:, 1], h[:, 2], len(h)), goal])
    grab = np.where((_norm(h[:, :2] - o[:, :2]) > 0.02) | (np.abs(h[:, 2] - o[:, 2]) > 0.1), 0.0, 0.65)
    return _act(h, to, 10.0, grab)



def handle_press_side(obs):          # sawyer_handle_press_side_v3_policy.py
    h, g, o, goal = _parts(obs)
    to = _first([_norm(h[:, :2] - o[:, :2]) > 0.02], [_off(o, 0, 0, 0.2), _off(o, 0, 0, -0.5)])
    return _act(h, to, 25.0, 1.0)


def handle_press(obs):               # sawyer_handle_press_v3_policy.py
    h, g, o, goal = _parts(obs)
    b = _off(o, 0, -0.02, 0)
    to = _first([_norm(h[:, :2] - b[:, :2]) > 0.02], [_off(b, 0, 0, 0.2), _off(b, 0, 0, -0.5)])
    return _act(h, to, 25.0, -1.0)


def handle_pull_side(obs):           # sawyer_handle_pull_side_v3_policy.py
    h, g, o, goal = _parts(obs)
    to = _first([_norm(h[:, :2] - o[:, :2]) > 0.04, np.abs(h[:, 2] - o[:, 2]) > 0.03], [_off(o, 0, 0, 0.1), o, _off(o, 0, 0, 1.0)])
    grab = np.where((_norm(h[:, :2] - o[:, :2]) > 0.04) | (np.abs(h[:, 2] - o[:, 2]) > 0.04), 0.0, 0.6)
    return _act(h, to, 25.0, grab)


def handle_pull(obs):                # sawyer_handle_pull_v3_policy.py (its middle branch returns the SCALAR z: broadcast to x, y, z)
    h, g, o, goal = _parts(obs)
    k = _off(o, 0, -0.04, 0)
    to = _first([_norm(h[:, :2] - k[:, :2]) > 0.02, np.abs(h[:, 2] - k[:, 2]) > 0.02], [k, _xyz(k[:, 2], k[:, 2], k[:, 2], len(h)), _off(k, 0, 0, 0.1)])
    return _act(h, to, 25.0, 1.0)


def lever_pull(obs):                 # sawyer_lever_pull_v3_policy.py
    h, g, o, goal = _parts(obs)
    v = _off(o, 0, -0.055, 0)
    to = _first([_norm(h[:, :2] - v[:, :2]) > 0.02, np.abs(h[:, 2] - v[:, 2]) > 0.02], [_off(v, 0, 0, -0.1), v, _off(v, 0, 0.08, 0.02)])
    return _act(h, to, 25.0, 1.0)


def peg_unplug_side(obs):            # sawyer_peg_unplug_side_v3_policy.py
    h, g, o, goal = _parts(obs)
    peg = _off(o, -0.02, 0, 0.035)
    to = _first([_norm(h[:, :2] - peg[:, :2]) > 0.04, np.abs(h[:, 2] - 0.15) > 0.02],
                [_off(peg, 0, 0, 0.2), _xyz(peg[:, 0], peg[:, 1], 0.15, len(h)), _off(h, 0.01, 0, 0)])
    grab = np.where((_norm(h[:, :2] - peg[:, :2]) > 0.04) | (np.abs(h[:, 2] - peg[:, 2]) > 0.15), -1.0, 0.1)
    return _act(h, to, 25.0, grab)


def pick_out_of_hole(obs):           # sawyer_pick_out_of_hole_v3_policy.py
    h, g, o, goal = _parts(obs)
    puck = _off(o, 0, 0, 0.02)
    to = _first([_norm(h[:, :2] - puck[:, :2]) > 0.02, np.abs(h[:, 2] - puck[:, 2]) > 0.01, np.abs(h[:, 2] - goal[:, 2]) > 0.04],
                [_off(puck, 0, 0, 0.15), puck, _xyz(h[:, 0], h[:, 1], goal[:, 2], len(h)), goal])
    grab = np.where((_norm(h[:, :2] - puck[:, :2]) > 0.02) | (np.abs(h[:, 2] - puck[:, 2]) > 0.15), 0.0, 0.1)
    return _act(h, to, 25.0, grab)


def pick_place_wall(obs):            # sawyer_pick_place_wall_v3_policy.py
    h, g, o, goal = _parts(obs)
    puck = _off(o, -0.005, 0, 0)
    box = (-0.15 <= h[:, 0]) & (h[:, 0] <= 0.35) & (0.60 <= h[:, 1]) & (h[:, 1] <= 0.80)
    to = _first([_norm(h[:, :2] - puck[:, :2]) > 0.015, (np.abs(h[:, 2] - puck[:, 2]) > 0.04) & (puck[:, 2] < 0.03),
                 box & (h[:, 2] < 0.25), box & (h[:, 2] < 0.35), np.abs(h[:, 2] - goal[:, 2]) > 0.01],
                [_off(puck, 0, 0, 0.1), _off(puck, 0, 0, 0.03), _off(h, 0, 0, 1), _xyz(goal[:, 0], goal[:, 1], h[:, 2], len(h)),
                 _xyz(h[:, 0], h[:, 1], goal[:, 2], len(h)), goal])
    grab = np.where((_norm(h[:, :2] - o[:, :2]) > 0.015) | (np.abs(h[:, 2] - o[:, 2]) > 0.1), 0.0, 0.9)
    return _act(h, to, 10.0, grab)


def plate_slide_back_side(obs):      # sawyer_plate_slide_back_side_v3_policy.py
    h, g, o, goal = _parts(obs)
    puck = _off(o, 0.023, 0, 0.025)
    to = _first([_norm(h[:, :2] - puck[:, :2]) > 0.01, np.abs(h[:, 2] - puck[:, 2]) > 0.04],
                [_off(puck, 0, 0, 0.07), puck, _xyz(h[:, 0] + 0.1, 0.6, h[:, 2], len(h))])
    return _act(h, to, 10.0, 1.0)


def plate_slide_back(obs):           # sawyer_plate_slide_back_v3_policy.py
    h, g, o, goal = _parts(obs)
    puck = _off(o, 0, -0.065, 0.025)
    to = _first([_norm(h[:, :2] - puck[:, :2]) > 0.01, np.abs(h[:, 2] - puck[:, 2]) > 0.04, h[:, 1] > 0.7, h[:, 1] > 0.6],
                [_off(puck, 0, 0, 0.1), puck, _off(h, 0, -0.1, 0), _xyz(0.15, 0.55, h[:, 2], len(h)), _xyz(h[:, 0] - 0.1, 0.55, h[:, 2], len(h))])
    return _act(h, to, 10.0, -1.0)


def plate_slide_side(obs):           # sawyer_plate_slide_side_v3_policy.py
    h, g, o, goal = _parts(obs)
    puck = _off(o, 0.07, 0, -0.005)
    to = _first([_norm(h[:, :2] - puck[:, :2]) > 0.04, np.abs(h[:, 2] - puck[:, 2]) > 0.04, h[:, 0] > -0.2],
                [_off(puck, 0, 0, 0.1), puck, _xyz(h[:, 0] - 0.1, 0.6, h[:, 2], len(h)), _off(puck, -0.1, 0, 0)])
    return _act(h, to, 25.0, 1.0)



def plate_slide(obs):                # sawyer_plate_slide_v3_policy.py
    h, g, o, goal = _parts(obs)
    puck = _off(o, 0, -0.055, 0.03)
    to = _first([~(_norm(h[:, :2] - puck[:, :2]) <= 0.03), np.abs(h[:, 2] - puck[:, 2]) > 0.04],
                [_off(puck, 0, 0, 0.1), puck, _xyz(goal[:, 0], 0.9, puck[:, 2], len(h))])
    return _act(h, to, 10.0, -1.0)


def push_back(obs):                  # sawyer_push_back_v3_policy.py
    h, g, o, goal = _parts(obs)
    to = _first([_norm(h[:, :2] - o[:, :2]) > 0.04, np.abs(h[:, 2] - o[:, 2]) > 0.055],
                [_off(o, 0, 0, 0.3), o, goal + _xyz(0.0, 0.0, h[:, 2], len(h))])
    grab = np.where((_norm(h[:, :2] - o[:, :2]) > 0.04) | (np.abs(h[:, 2] - o[:, 2]) > 0.05), 0.0, 0.9)
    return _act(h, to, 10.0, grab)


def push_wall(obs):                  # sawyer_push_wall_v3_policy.py
    h, g, o, goal = _parts(obs)
    q = _off(o, -0.005, 0, 0)
    x, y = q[:, 0], q[:, 1]
    to = _first([_norm(h[:, :2] - q[:, :2]) > 0.02, np.abs(h[:, 2] - q[:, 2]) > 0.04,
                 (-0.1 <= x) & (x <= 0.3) & (0.65 <= y) & (y <= 0.75),
                 (((-0.15 < x) & (x < 0.05)) | ((0.15 < x) & (x < 0.35))) & (0.695 <= y) & (y <= 0.755)],
                [_off(q, 0, 0, 0.2), _off(q, 0, 0, 0.03), _off(h, -1, 0, 0), _off(h, 0, 1, 0), goal])
    grab = np.where((_norm(h[:, :2] - o[:, :2]) > 0.02) | (np.abs(h[:, 2] - o[:, 2]) > 0.1), 0.0, 0.6)
    return _act(h, to, 10.0, grab)


def reach_wall(obs):                 # sawyer_reach_wall_v3_policy.py
    h, g, o, goal = _parts(obs)
    low = (-0.1 <= h[:, 0]) & (h[:, 0] <= 0.3) & (0.60 <= h[:, 1]) & (h[:, 1] <= 0.80) & (h[:, 2] < 0.25)
    return _act(h, _first([low], [_off(goal, 0, 0, 1.0), goal]), 5.0, 0.0)


def shelf_place(obs):                # sawyer_shelf_place_v3_policy.py
    h, g, o, goal = _parts(obs)
    blk = _off(o, -0.005, 0, 0.015)
    to = _first([_norm(h[:, :2] - blk[:, :2]) > 0.04, np.abs(h[:, 2] - blk[:, 2]) > 0.04, np.abs(h[:, 0] - goal[:, 0]) > 0.02, h[:, 2] < 0.30],
                [_off(blk, 0, 0, 0.3), blk, _xyz(goal[:, 0], h[:, 1], 0.3, len(h)), _off(h, 0, 0, 0.30), _off(h, 0, 0.05, 0)])
    grab = np.where((_norm(h[:, :2] - o[:, :2]) > 0.04) | (np.abs(h[:, 2] - o[:, 2]) > 0.15), -1.0, 0.7)
    return _act(h, to, 25.0, grab)


def soccer(obs):                     # sawyer_soccer_v3_policy.py
    h, g, o, goal = _parts(obs)
    ball = _off(o, 0, 0, 0.03)
    z = np.where(_norm(h[:, :2] - ball[:, :2]) < 0.02, 0.1, 0.03)
    dx = ball[:, 0] - goal[:, 0]
    push = _first([dx < -0.05, dx > 0.05], [_off(ball, -0.03, 0, 0), _off(ball, 0.03, 0, 0), _off(ball, 0, -0.03, 0)])
    push[:, 2] = z
    to = _first([_norm(h - push) > 0.01], [push, ball])
    return _act(h, to, 25.0, 1.0)


def _stick(obs, sx, thermos_off, goal_off, p):   # sawyer_stick_pull_v3_policy.py / sawyer_stick_push_v3_policy.py
    obs = np.asarray(obs, dtype=np.float64)
    h, g, o, goal = _parts(obs)
    st = _off(o, sx, 0, 0.03)
    th = obs[:, 11:14] + np.array(thermos_off)
    gl = goal + np.array(goal_off)
    away = np.abs(st[:, 0] - th[:, 0]) > 0.04
    to = _first([away & (_norm(h[:, :2] - st[:, :2]) > 0.02), away & (np.abs(h[:, 2] - st[:, 2]) > 0.02),
                 away & (np.abs(st[:, 1] - th[:, 1]) > 0.02), away & (np.abs(st[:, 2] - th[:, 2]) > 0.02), away],
                [_off(st, 0, 0, 0.1), st, _xyz(st[:, 0], th[:, 1], st[:, 2], len(h)), _xyz(st[:, 0], th[:, 1], th[:, 2], len(h)), th, gl])
    grab = np.where((_norm(h[:, :2] - st[:, :2]) > 0.02) | (np.abs(h[:, 2] - st[:, 2]) > 0.1), -1.0, 0.7)
    return _act(h, to, p, grab)


def stick_pull(obs):
    return _stick(obs, -0.015, [-0.015, 0.0, 0.03], [-0.05, 0.0, 0.0], 25.0)


def stick_push(obs):
    return _stick(obs, 0.015, [0.0, 0.0, 0.0], [0.0, 0.0, 0.132], 10.0)


def sweep_into(obs):                 # sawyer_sweep_into_v3_policy.py
    h, g, o, goal = _parts(obs)
    cube = _off(o, -0.005, 0, 0.01)
    to = _first([_norm(h[:, :2] - cube[:, :2]) > 0.04, np.abs(h[:, 2] - cube[:, 2]) > 0.04], [_off(cube, 0, 0, 0.3), cube, goal])
    grab = np.where((_norm(h[:, :2] - o[:, :2]) > 0.04) | (np.abs(h[:, 2] - o[:, 2]) > 0.15), -1.0, 0.7)
    return _act(h, to, 25.0, grab)


def sweep(obs):                      # sawyer_sweep_v3_policy.py
    h, g, o, goal = _parts(obs)
    cube = _off(o, 0, 0, 0.015)
    near = h[:, 0] < 0.2
    to = _first([near & (_norm(h[:, :2] - cube[:, :2]) > 0.04), near & (np.abs(h[:, 2] - cube[:, 2]) > 0.04)],
                [_off(cube, 0, 0, 0.3), cube, _off(goal, 0, 0, 0.1)])
    grab = np.where((_norm(h[:, :2] - o[:, :2]) > 0.04) | (np.abs(h[:, 2] - o[:, 2]) > 0.15), -1.0, np.where(o[:, 0] < 0.4, 0.7, -1.0))
    return _act(h, to, 25.0, grab)


POLICIES = {"reach-v3": reach, "push-v3": push, "pick-place-v3": pick_place, "door-open-v3": door_open,
            "drawer-open-v3": drawer_open, "drawer-close-v3": drawer_close, "button-press-topdown-v3": button_press_topdown,
            "peg-insert-side-v3": peg_insert_side, "window-open-v3": window_open, "window-close-v3": window_close,
            "assembly-v3": assembly, "basketball-v3": basketball, "bin-picking-v3": bin_picking, "box-close-v3": box_close,
            "button-press-topdown-wall-v3": button_press_topdown_wall, "button-press-v3": button_press,
            "button-press-wall-v3": button_press_wall, "coffee-button-v3": coffee_button,
            "coffee-pull-v3": coffee_pull, "coffee-push-v3": coffee_push, "dial-turn-v3": dial_turn, "disassemble-v3": disassemble,
            "door-close-v3": door_close, "door-lock-v3": door_lock, "door-unlock-v3": door_unlock, "faucet-close-v3": faucet_close,
            "faucet-open-v3": faucet_open, "hammer-v3": hammer, "hand-insert-v3": hand_insert,
            "handle-press-side-v3": handle_press_side, "handle-press-v3": handle_press, "handle-pull-side-v3": handle_pull_side,
            "handle-pull-v3": handle_pull, "lever-pull-v3": lever_pull, "peg-unplug-side-v3": peg_unplug_side,
            "pick-out-of-hole-v3": pick_out_of_hole, "pick-place-wall-v3": pick_place_wall,
            "plate-slide-back-side-v3": plate_slide_back_side, "plate-slide-back-v3": plate_slide_back, "plate-slide-side-v3": plate_slide_side,
            "plate-slide-v3": plate_slide, "push-back-v3": push_back, "push-wall-v3": push_wall, "reach-wall-v3": reach_wall,
            "shelf-place-v3": shelf_place, "soccer-v3": soccer, "stick-pull-v3": stick_pull, "stick-push-v3": stick_push,
            "sweep-into-v3": sweep_into, "sweep-v3": sweep}


def batched_actions(task_names, obs):
    """float32 [N, 4] actions (clipped to [-1, 1] like the env would) for a vector env whose env i runs `task_names[i]`."""
    obs = np.asarray(obs)
    names = np.asarray(task_names)
    out = np.zeros((len(obs), 4), dtype=np.float32)
    for n in np.unique(names):
        idx = np.flatnonzero(names == n)
        out[idx] = POLICIES[n](obs[idx, :39])
    return np.clip(out, -1, 1)
