import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
name = sys.argv[1]
env = MetaWorldGpuVectorEnv("MT1", name, num_envs=82, seed=0, precision="fp32")
env.reset()
env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (64, 82, 4)).astype(np.float32))
print(name, "lpb", os.environ.get("MW_LANES_PER_BLOCK"), " ".join(f"{env.ctx.step_resident(40) / 40:6.1f}" for _ in range(6)), flush=True)
