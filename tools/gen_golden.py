#!/usr/bin/env python
"""Generate golden (seed, action) -> (obs, reward, success, info, state) traces by running the
REFERENCE's own Python (metaworld/*.py under /root/reference) on the oracle engine (oracle/refshim.py).

Only runs where /root/reference exists (the build container).  The resulting small .npz files are
committed under tests/golden/ and are what the GPU parity tests replay.  Per task it records, for
`--episodes` goals of MT1(task, seed):
  rand_vecs[50,6]          the benchmark's goal table (metaworld/__init__.py:114-179)
  reset_obs[E,39], reset_qpos, reset_qvel, reset_mocap, reset_warm   state right after reset()
  actions[E,T,4]           random or scripted actions
  obs[E,T,39], reward[E,T], success[E,T], info[E,T,6], truncate[E,T]
  qpos/qvel/mocap/warm[E,T,...]   state after every step (for one-step-from-synchronised-state parity)
"""
from __future__ import annotations

import argparse
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

INFO_KEYS = ["near_object", "grasp_success", "grasp_reward", "in_place_reward", "obj_to_target", "unscaled_reward"]


def run_task(name, seed, episodes, steps, mode, rng, reward_version="v2", first_goal=0):
    import metaworld
    from metaworld.policies import ENV_POLICY_MAP
    mt1 = metaworld.MT1(name, seed=seed)
    env = mt1.train_classes[name](reward_function_version=reward_version)
    env.seed(seed)
    policy = ENV_POLICY_MAP[name]()
    import pickle
    rand_vecs = np.array([pickle.loads(t.data)["rand_vec"] for t in mt1.train_tasks], dtype=np.float64)
    if rand_vecs.shape[1] == 3:
        rand_vecs = np.concatenate([rand_vecs, np.zeros((len(rand_vecs), 3))], axis=1)
    out = {k: [] for k in ("reset_obs", "reset_qpos", "reset_qvel", "reset_mocap", "reset_warm", "actions", "obs", "reward",
                           "success", "info", "truncate", "qpos", "qvel", "mocap", "warm", "goal_idx")}
    d = env.data
    for ep in range(episodes):
        gi = (first_goal + ep) % len(mt1.train_tasks)
        env.set_task(mt1.train_tasks[gi])
        obs, _ = env.reset()
        out["goal_idx"].append(gi)
        out["reset_obs"].append(obs.copy()); out["reset_qpos"].append(d.qpos.copy()); out["reset_qvel"].append(d.qvel.copy())
        out["reset_mocap"].append(d.mocap_pos[0].copy()); out["reset_warm"].append(d.qacc_warmstart.copy())
        ep_rec = {k: [] for k in ("actions", "obs", "reward", "success", "info", "truncate", "qpos", "qvel", "mocap", "warm")}
        for t in range(steps):
            if mode == "random" or (mode == "mixed" and ep % 2 == 1):
                a = rng.uniform(-1, 1, 4).astype(np.float32)
            else:
                a = np.clip(policy.get_action(obs.copy()), -1, 1).astype(np.float32)
                if mode == "mixed":
                    a = np.clip(a + rng.normal(0, 0.1, 4), -1, 1).astype(np.float32)
            obs, r, term, trunc, info = env.step(a)
            ep_rec["actions"].append(a); ep_rec["obs"].append(obs.copy()); ep_rec["reward"].append(float(r))
            ep_rec["success"].append(float(info["success"])); ep_rec["truncate"].append(bool(trunc))
            ep_rec["info"].append([float(info[k]) for k in INFO_KEYS])
            ep_rec["qpos"].append(d.qpos.copy()); ep_rec["qvel"].append(d.qvel.copy())
            ep_rec["mocap"].append(d.mocap_pos[0].copy()); ep_rec["warm"].append(d.qacc_warmstart.copy())
        for k, v in ep_rec.items():
            out[k].append(np.array(v))
    res = {k: np.array(v) for k, v in out.items()}
    res["rand_vecs"] = rand_vecs
    res["seed"] = np.array(seed)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tasks", nargs="*", default=["reach-v3"])
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--episodes", type=int, default=4)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--mode", default="mixed", choices=["random", "policy", "mixed"])
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--first-goal", type=int, default=0, help="goal index of the first episode (episode k uses first_goal + k)")
    ap.add_argument("--tag", default="", help="file name infix: trace_<tag>_<task>_seed<seed>.npz (e.g. policy200: --mode policy --steps 200 --episodes 1 --first-goal 7)")
    ap.add_argument("--reward-version", default="v2", choices=["v1", "v2"], help="v1: files are named trace_v1_<task>_seed<seed>.npz")
    args = ap.parse_args()
    from oracle import refshim
    refshim.install()
    os.makedirs(args.out, exist_ok=True)
    for name in args.tasks:
        rng = np.random.default_rng(args.seed)
        res = run_task(name, args.seed, args.episodes, args.steps, args.mode, rng, args.reward_version, args.first_goal)
        tag = ("v1_" if args.reward_version == "v1" else "") + (args.tag + "_" if args.tag else "")
        path = os.path.join(args.out, f"trace_{tag}{name}_seed{args.seed}.npz")
        np.savez_compressed(path, **res)
        print(name, "->", path, f"({os.path.getsize(path) / 1024:.0f} KiB)", "success steps:", int(res["success"].sum()))


if __name__ == "__main__":
    main()
