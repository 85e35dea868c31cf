#!/usr/bin/env python
"""Per-step kernel time over an episode (random actions): MT50 @4096 and the heaviest tasks alone @82."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv

def profile(env, n, steps=260):
    env.reset()
    env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (64, n, 4)).astype(np.float32))
    out = []
    for s in range(0, steps, 20):
        out.append(env.ctx.step_resident(20) / 20)
    return out

env = MetaWorldGpuVectorEnv("MT50", num_envs=4096, seed=0, precision="fp32")
print("MT50@4096      ", " ".join(f"{x:6.1f}" for x in profile(env, 4096)), flush=True)
env.close()
for name in sys.argv[1:]:
    env = MetaWorldGpuVectorEnv("MT1", name, num_envs=82, seed=0, precision="fp32")
    print(f"{name:15s}", " ".join(f"{x:6.1f}" for x in profile(env, 82)), flush=True)
    env.close()
