#!/usr/bin/env python
"""BASELINE config 5 as a parity fixture: ML45-train @ 2048 envs with scripted policies, success per env from the REFERENCE.

For every env of `MetaWorldGpuVectorEnv("ML45-train", num_envs=2048, seed=42)` -- env j of a task runs goal j mod (number of train goals) of
the task's ML45 table -- the unmodified reference env class + reference scripted policy (metaworld/policies) run ONE closed-loop
500-step episode on the oracle engine (oracle/refshim.py) with that goal made visible (tests/metaworld/test_evaluation.py:70-82),
and record whether info["success"] was ever 1 (what metaworld/evaluation.py counts).  tests/test_gpu_cfg5.py replays the same
(task, goal) assignment on the GPU with the device policies and compares per-task success counts.
Needs /root/reference.  usage: tools/gen_cfg5_fixture.py [procs=8]  -> tests/golden/cfg5_ml45_train_2048_seed42.npz"""
import os
import pickle
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
N, SEED, BENCH = 2048, 42, "ML45-train"


def assignment():
    """(task name, goal index, rand_vec) of every env: physics-free (metaworld_amd.goals / tasks), same rule as the GPU test"""
    from metaworld_amd import tasks as T
    names = T.benchmark_task_names(BENCH, None)
    per, rem = divmod(N, len(names))
    out = []
    for i, name in enumerate(names):
        tab = T.goal_table(BENCH, name, SEED)
        for j in range(per + (1 if i < rem else 0)):
            out.append((name, j % len(tab), np.array(tab[j % len(tab)])))
    return out


def work(chunk):
    from oracle import refshim
    refshim.install()
    from metaworld.env_dict import ALL_V3_ENVIRONMENTS
    from metaworld.policies import ENV_POLICY_MAP
    from metaworld.types import Task
    envs, res = {}, []
    for (e, name, g, rv) in chunk:
        if name not in envs:
            envs[name] = (ALL_V3_ENVIRONMENTS[name](), ENV_POLICY_MAP[name]())
        env, policy = envs[name]
        n_rv = len(env._random_reset_space.low)
        env.set_task(Task(env_name=name, data=pickle.dumps(dict(rand_vec=rv[:n_rv], env_cls=ALL_V3_ENVIRONMENTS[name], partially_observable=False))))
        obs, _ = env.reset()
        ok, first = 0, -1
        for t in range(500):
            obs, r, te, tr, info = env.step(policy.get_action(obs))
            if int(info["success"]) == 1:
                ok, first = 1, t
                break          # success "ever" is decided; the episode's remaining steps cannot change it
        res.append((e, ok, first))
    return res


def main():
    import multiprocessing as mp
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    A = assignment()
    # unique (task, goal) pairs only: envs sharing both are the same episode
    uniq = {}
    for e, (name, g, rv) in enumerate(A):
        uniq.setdefault((name, g), (e, name, g, rv))
    jobs = sorted(uniq.values(), key=lambda x: x[1])
    chunks = [jobs[k::procs] for k in range(procs)]
    with mp.get_context("spawn").Pool(procs) as pool:
        got = {}
        for part in pool.imap_unordered(work, chunks):
            for e, ok, first in part:
                got[A[e][0], A[e][1]] = (ok, first)
    succ = np.array([got[name, g][0] for name, g, _ in A], dtype=np.uint8)
    first = np.array([got[name, g][1] for name, g, _ in A], dtype=np.int32)
    path = os.path.join(ROOT, "tests", "golden", "cfg5_ml45_train_2048_seed42.npz")
    np.savez_compressed(path, task=np.array([a[0] for a in A]), goal=np.array([a[1] for a in A], dtype=np.int32), success=succ, first_success=first)
    names = sorted(set(a[0] for a in A), key=[a[0] for a in A].index)
    for n in names:
        m = np.array([a[0] == n for a in A])
        print(f"{n:30s} {int(succ[m].sum()):3d}/{int(m.sum()):3d}")
    print(f"{len(jobs)} distinct episodes, mean success {succ.mean():.4f} -> {path}")


if __name__ == "__main__":
    main()
