#!/usr/bin/env python
"""ms per VectorEnv.step for each task alone (n envs, fp32, random actions, actions resident): finds the groups that
set the MT50 kernel's critical path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd import tasks as T  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 82
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
rows = []
for name in T.ALL_V3:
    env = MetaWorldGpuVectorEnv("MT1", name, num_envs=n, seed=0, precision=prec)
    env.reset()
    env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (32, n, 4)).astype(np.float32))
    env.ctx.step_resident(3)
    ms = env.ctx.step_resident(steps) / steps
    ic = np.array([env.ctx.read_int(e, "icount") for e in range(0, n, max(1, n // 8))])
    rows.append((ms, name, ic[:, 0].max(), ic[:, 1].max(), ic[:, 2].max(), ic[:, 3].max()))
    env.close()
for r in sorted(rows, reverse=True):
    print(f"{r[1]:30s} {r[0]:8.2f} ms/step  ncon<= {r[2]:3d} nefc<= {r[3]:3d} niter<= {r[4]:2d} flags {r[5]}")
