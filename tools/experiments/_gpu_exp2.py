#!/usr/bin/env python
"""How the MT50 step time scales with the number of concurrent task groups (82 envs per task)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv

order = [l.split()[0] for l in open(os.path.join(ROOT, "tools/data/per_task_order.txt")) if "ms/step" in l]
for sel in (order[:1], order[:2], order[:5], order[:10], order[:25], order[25:], order):
    n = 82 * len(sel)
    env = MetaWorldGpuVectorEnv("MT50", num_envs=n, seed=0, precision="fp32", task_names=sel)
    env.reset()
    env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (32, n, 4)).astype(np.float32))
    env.ctx.step_resident(3)
    ms = env.ctx.step_resident(20) / 20
    print(f"{len(sel):3d} tasks ({sel[0]} .. {sel[-1]}), {n} envs: {ms:7.2f} ms/step", flush=True)
    env.close()
