#!/usr/bin/env python
"""Closed-loop success rates on the GPU with the batched scripted policies (metaworld_amd/policies.py): MT10 (BASELINE
config 3 size by default) or any benchmark split (4th argument, e.g. ML45-train = config 5), every env runs whole 500-step episodes of its task with a goal drawn by the
task-sampling stream; success = the reference's `info["success"]` reached at any step of the episode (what
metaworld/evaluation.py counts with terminate_on_success).  Reports per-task success and the end-to-end rate including
the host-side policy and the PCIe round trip of every step.  usage: tools/policy_eval_gpu.py [num_envs] [episodes] [precision] [benchmark]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd import policies as P
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
episodes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
bench = sys.argv[4] if len(sys.argv) > 4 else "MT10"       # e.g. ML45-train: goal made visible like tests/metaworld/test_evaluation.py:70-82
env = MetaWorldGpuVectorEnv(bench, num_envs=n, seed=42, use_one_hot=bench.startswith("MT"), precision=prec, partially_observable=False)
names = np.array(env.env_task_names)
obs, _ = env.reset()
succ = np.zeros(n, dtype=bool); wins = {t: 0 for t in env.task_list}; tot = {t: 0 for t in env.task_list}
t0 = time.perf_counter(); steps = 0
for ep in range(episodes):
    succ[:] = False
    for t in range(500):
        obs, rew, term, trunc, info = env.step(P.batched_actions(names, obs))
        succ |= info["success"].astype(bool)
        steps += 1
    for tname in env.task_list:
        m = names == tname
        wins[tname] += int(succ[m].sum()); tot[tname] += int(m.sum())
dt = time.perf_counter() - t0
for tname in env.task_list:
    print(f"{tname:28s} success {wins[tname]:5d}/{tot[tname]:5d} = {wins[tname] / tot[tname]:.3f}")
print(f"mean success {np.mean([wins[t] / tot[t] for t in env.task_list]):.3f}   {n} envs x {steps} steps in {dt:.1f} s = {n * steps / dt / 1e3:.1f} k env-steps/s "
      f"(host policy + PCIe in/out every step, {prec})")
