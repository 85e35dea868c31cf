#!/usr/bin/env python
"""ms per VectorEnv.step for each task alone (n envs, random actions, actions resident), measured over a window that
starts `warm` steps into the episode (the step gets slower as the flailing arm makes contacts): finds the groups that
set the MT50 kernel's critical path.  usage: per_task_timing.py [n=82] [steps=100] [precision=fp64] [warm=150] [task ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd import tasks as T  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 82
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
prec = sys.argv[3] if len(sys.argv) > 3 else "fp64"
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 150
names = sys.argv[5:] or T.ALL_V3
rows = []
for name in names:
    env = MetaWorldGpuVectorEnv("MT1", name, num_envs=n, seed=0, precision=prec)
    env.reset()
    env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (64, n, 4)).astype(np.float32))
    if warm:
        env.ctx.step_resident(warm)
    ms = env.ctx.step_resident(steps) / steps
    ic = np.array([env.ctx.read_int(e, "icount") for e in range(0, n, max(1, n // 8))])
    nv = env.ctx._check(env.ctx.lib.column_size(env.ctx.ptr, 0, b"qvel"))
    rows.append((ms, name, nv, ic[:, 0].max(), ic[:, 1].max(), ic[:, 2].max(), ic[:, 3].max()))
    env.close()
for r in sorted(rows, reverse=True):
    print(f"{r[1]:30s} {r[0]:8.2f} ms/step  nv {r[2]:2d} ncon<= {r[3]:3d} nefc<= {r[4]:3d} niter<= {r[5]:2d} flags {r[6]}", flush=True)
