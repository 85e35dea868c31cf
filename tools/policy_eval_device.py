#!/usr/bin/env python
"""BASELINE config 5 with the scripted policies ON THE DEVICE (mw_policy_rollout): every env of a benchmark split runs
`episodes` whole 500-step episodes of its task, one goal per episode (env j of a task starts at goal j and walks the table),
policy kernel + step kernel back to back with no host round trip; success = `info["success"]` ever 1 within the episode, the
reference's gate is 0.8 per task (tests/metaworld/test_scripted_policies.py:35).
usage: tools/policy_eval_device.py [num_envs] [episodes] [precision] [benchmark] [lockstep]
(lockstep = 1: every env of a task gets the same goal, like the reference's identically seeded task-selection streams do)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd import tasks as T
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
episodes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
bench = sys.argv[4] if len(sys.argv) > 4 else "ML45-train"
env = MetaWorldGpuVectorEnv(bench, num_envs=n, seed=42, precision=prec, partially_observable=False, max_episode_steps=500)
names = np.array(env.env_task_names)
pid = np.array([T.ALL_V3.index(t) for t in names], dtype=np.int32)
rank_in_task = np.concatenate([np.arange((names == t).sum()) for t in env.task_list])          # envs are task-major contiguous
lockstep = len(sys.argv) > 5 and sys.argv[5] == "1"
sched = ((0 if lockstep else rank_in_task[None, :]) + np.arange(episodes + 1)[:, None] + np.zeros((1, n), dtype=int)) % 50
t0 = time.perf_counter()
ep, su, ms = env.ctx.policy_rollout(pid, sched, 500 * episodes)
dt = time.perf_counter() - t0
assert (ep == episodes).all()
rates = {}
for t in env.task_list:
    m = names == t
    rates[t] = su[m].sum() / ep[m].sum()
    print(f"{t:28s} success {int(su[m].sum()):5d}/{int(ep[m].sum()):5d} = {rates[t]:.3f}")
steps = 500 * episodes
print(f"mean success {np.mean(list(rates.values())):.3f}   tasks >= 0.8: {sum(r >= 0.8 for r in rates.values())}/{len(rates)}")
print(f"{n} envs x {steps} steps: kernels {ms / steps:.3f} ms/step = {n * steps / ms:.1f} k env-steps/s on the device "
      f"(policy + step kernels, {prec}); wall {dt:.1f} s incl. reset and upload")
