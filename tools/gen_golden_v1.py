#!/usr/bin/env python
"""Golden transitions for the v1 reward functions: the REFERENCE's own Python with reward_function_version="v1" on the oracle engine
(oracle/refshim.py), like tools/gen_golden.py.  Per task two kinds of episodes -- (a) noisy scripted policy + random actions, 60
steps each, (b) two clean scripted-policy episodes of 200 steps (they reach the press / pull / pick / place branches) -- of which every
`--every`-th transition is kept: the state before the step (qpos, qvel, mocap, qacc_warmstart, previous observation), the action,
and the reference's (reward, success, info).  The v1 branches keep no memory between steps (pickCompleted & co are recomputed
from the current observation), so single transitions test them completely.  Needs /root/reference.
usage: tools/gen_golden_v1.py [task ...]   -> tests/golden/v1_<task>_seed42.npz"""
import argparse
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
from tools.gen_golden import run_task  # noqa: E402



def transitions(G, every, first_episode_only):
    out = {k: [] for k in ("goal", "qpos", "qvel", "mocap", "warm", "prev18", "action", "reward", "success", "info")}
    E, TT = G["actions"].shape[:2]
    for e in range(1 if first_episode_only else E):
        for t in range(0, TT, every):
            src = (G["reset_qpos"][e], G["reset_qvel"][e], G["reset_mocap"][e], G["reset_warm"][e], G["reset_obs"][e][:18]) if t == 0 else \
                (G["qpos"][e, t - 1], G["qvel"][e, t - 1], G["mocap"][e, t - 1], G["warm"][e, t - 1], G["obs"][e, t - 1][:18])
            for k, v in zip(("qpos", "qvel", "mocap", "warm", "prev18"), src):
                out[k].append(v)
            out["goal"].append(G["goal_idx"][e]); out["action"].append(G["actions"][e, t]); out["reward"].append(G["reward"][e, t])
            out["success"].append(G["success"][e, t]); out["info"].append(G["info"][e, t])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tasks", nargs="*")
    ap.add_argument("--every", type=int, default=4)
    args = ap.parse_args()
    from oracle import refshim
    refshim.install()
    from metaworld_amd import tasks as T
    for name in (args.tasks or T.ALL_V3):
        basket = name == "basketball-v3"          # only the first episode of a fresh env is history-free (the drifting goal site)
        parts = [transitions(run_task(name, 42, 2, 60, "mixed", np.random.default_rng(42), "v1"), args.every, basket)]
        if True:          # (b): clean scripted-policy episodes reach the press / pull / pick / place branches
            parts.append(transitions(run_task(name, 42, 2, 200, "policy", np.random.default_rng(43), "v1"), args.every, basket))
        res = {k: np.array(sum((p[k] for p in parts), [])) for k in parts[0]}
        path = os.path.join(ROOT, "tests", "golden", f"v1_{name}_seed42.npz")
        np.savez_compressed(path, **res)
        print(f"{name:30s} {len(res['reward']):4d} transitions, reward range [{res['reward'].min():.1f}, {res['reward'].max():.1f}], "
              f"{int(res['success'].sum())} successes ({os.path.getsize(path) / 1024:.0f} KiB)", flush=True)


if __name__ == "__main__":
    main()
