import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
env = MetaWorldGpuVectorEnv("MT50", num_envs=200, seed=1, use_one_hot=True, precision="fp32", max_episode_steps=10)
obs, _ = env.reset()
print("reset finite", np.isfinite(obs).all())
rng = np.random.default_rng(0)
for t in range(10):
    obs, rew, term, trunc, infos = env.step(rng.uniform(-1, 1, (200, 4)).astype(np.float32))
    bad = np.nonzero(~np.isfinite(obs).all(1) | ~np.isfinite(rew))[0]
    print("step", t, "bad envs", [(int(b), env.env_task_names[b]) for b in bad][:10], flush=True)
    if len(bad): break
