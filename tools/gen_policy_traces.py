#!/usr/bin/env python
"""Closed-loop episodes of the REFERENCE scripted policies (metaworld/policies) on the REFERENCE env classes running on
the oracle engine (oracle/refshim.py): the manipulation regimes (grasp, lift, insert, ...) that random actions never reach.
Per task one episode (goal 0 of MT1(task, seed)), until 10 steps after the first success or `--steps`:
  actions[T,4] f32, obs18[T,18] (hand, gripper, object poses), reward[T], success[T], reset_obs[39]
-> tests/golden/policy_<task>_seed<seed>.npz; replayed OPEN LOOP by tests/test_policy_traces.py.  Needs /root/reference."""
import argparse
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tasks", nargs="*")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--steps", type=int, default=250)
    args = ap.parse_args()
    from oracle import refshim
    refshim.install()
    import metaworld
    from metaworld.env_dict import ALL_V3_ENVIRONMENTS
    from metaworld.policies import ENV_POLICY_MAP
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name in (args.tasks or list(ALL_V3_ENVIRONMENTS.keys())):
        mt1 = metaworld.MT1(name, seed=args.seed)
        env = mt1.train_classes[name]()
        env.set_task(mt1.train_tasks[0])
        policy = ENV_POLICY_MAP[name]()
        obs, _ = env.reset()
        rec = dict(actions=[], obs18=[], reward=[], success=[])
        first = None
        for t in range(args.steps):
            a = np.clip(policy.get_action(obs.copy()), -1, 1).astype(np.float32)
            obs, r, term, trunc, info = env.step(a)
            rec["actions"].append(a); rec["obs18"].append(obs[:18].copy()); rec["reward"].append(float(r))
            rec["success"].append(float(info["success"]))
            if first is None and info["success"]:
                first = t
            if first is not None and t >= first + 10:
                break
        res = {k: np.array(v) for k, v in rec.items()}
        res["goal_idx"] = np.array([0]); res["seed"] = np.array(args.seed)
        path = os.path.join(out_dir, f"policy_{name}_seed{args.seed}.npz")
        np.savez_compressed(path, **res)
        print(f"{name:30s} steps {len(res['reward']):3d} first success {first}  ({os.path.getsize(path) / 1024:.0f} KiB)", flush=True)


if __name__ == "__main__":
    main()
