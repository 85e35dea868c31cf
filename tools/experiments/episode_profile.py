#!/usr/bin/env python
"""ms per MT50 step over several episodes (random actions, SAME_STEP auto-reset every 500 steps)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
env = MetaWorldGpuVectorEnv("MT50", num_envs=4096, seed=42, use_one_hot=True, precision=sys.argv[1] if len(sys.argv) > 1 else "fp32")
env.reset()
env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (64, 4096, 4)).astype(np.float32))
ms = [env.ctx.step_resident(50) / 50 for _ in range(30)]
print(" ".join(f"{x:5.1f}" for x in ms))
print("mean over steps 500-1500: %.2f ms -> %.0f env-steps/s" % (np.mean(ms[10:30]), 4096 / np.mean(ms[10:30]) * 1e3))
