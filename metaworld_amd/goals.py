"""Physics-free goal tables: the `rand_vec`s the reference's benchmark builders draw.

`metaworld._make_tasks` (metaworld/__init__.py:114-179) seeds numpy's legacy global RNG once, then for every task of the
benchmark (in `env_dict` order) instantiates the env and calls `env.reset()` `_N_GOALS` = 50 times; each reset runs
`reset_model` twice (sawyer_xyz_env.py:664-682) and every `reset_model` draws `_get_state_rand_vec()`
(`np.random.uniform(low, high, size)`, sawyer_xyz_env.py:713-720) -- 21 tasks redraw while object and goal are too
close (each task's `while` loop, e.g. envs/sawyer_pick_place_v3.py:147-151).  Nothing in that stream depends on the
physics, so the tables for ANY seed can be reproduced with the same MT19937 stream; the committed tables dumped from
the reference itself (metaworld_amd/data/goals_seed42.npz, tools/gen_goal_tables.py) pin this restatement
(tests/test_goal_tables.py).
"""
from __future__ import annotations

import numpy as np

N_GOALS = 50

# redraw while ||v[0:2] - v[3:5]|| < threshold (object xy vs goal xy of the 6-vector)
REJECT_PAIR = {
    "assembly-v3": 0.1, "basketball-v3": 0.15, "box-close-v3": 0.25, "coffee-pull-v3": 0.15, "coffee-push-v3": 0.15,
    "disassemble-v3": 0.1, "hand-insert-v3": 0.15, "peg-insert-side-v3": 0.1, "pick-out-of-hole-v3": 0.15,
    "pick-place-v3": 0.15, "pick-place-wall-v3": 0.15, "push-back-v3": 0.15, "push-v3": 0.15, "push-wall-v3": 0.15,
    "reach-v3": 0.15, "reach-wall-v3": 0.15, "shelf-place-v3": 0.1, "soccer-v3": 0.15, "stick-pull-v3": 0.1,
    "stick-push-v3": 0.1,
}
# redraw while ||v[0:2] - fixed goal xy|| < threshold (envs/sawyer_sweep_into_goal_v3.py:108-110)
REJECT_FIXED = {"sweep-into-v3": 0.15}


def _draw(rs, low, high, task, goal):
    v = rs.uniform(low, high, size=low.size)
    if task in REJECT_PAIR:
        while np.linalg.norm(v[:2] - v[3:5]) < REJECT_PAIR[task]:
            v = rs.uniform(low, high, size=low.size)
    elif task in REJECT_FIXED:
        while np.linalg.norm(v[:2] - goal[:2]) < REJECT_FIXED[task]:
            v = rs.uniform(low, high, size=low.size)
    return v


def make_tables(task_names, seed, consts, resets_per_goal=1, draws_per_reset=2):
    """{task: float64[50][6]} for the tasks of one benchmark, in the reference's construction order."""
    rs = np.random.RandomState(seed)          # == np.random.seed(seed) + the global legacy functions
    out = {}
    for name in task_names:
        c = consts[name]
        low, high = np.array(c["reset_low"], dtype=np.float64), np.array(c["reset_high"], dtype=np.float64)
        goal = np.array(c.get("goal", [0, 0, 0]), dtype=np.float64)
        rows = []
        for _ in range(N_GOALS):
            for _ in range(resets_per_goal):
                for _ in range(draws_per_reset):
                    v = _draw(rs, low, high, name, goal)
            rows.append(np.concatenate([v, np.zeros(3)]) if v.size == 3 else v)
        out[name] = np.array(rows)
    return out
