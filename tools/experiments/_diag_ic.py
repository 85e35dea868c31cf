import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
n = 4096
env = MetaWorldGpuVectorEnv("MT50", num_envs=n, seed=1, use_one_hot=True, precision="fp32")  # noqa
env.reset()
env.ctx.upload_actions(np.random.default_rng(1).uniform(-1, 1, (97, n, 4)).astype(np.float32))
seen = set()
for chunk in range(12):
    env.ctx.step_resident(50)
    ic = np.array([env.ctx.read_int(e, "icount", 24) for e in range(n)])
    bad = np.flatnonzero((ic[:, 20] > 1000) | (ic[:, 20] < 0))
    new = [b for b in bad if b not in seen]
    print("steps", 50 * (chunk + 1), "bad envs", len(bad), "new", len(new), "status", env.ctx.status(), flush=True)
    for b in new[:6]:
        print("   env", b, env.env_task_names[b], "icount", ic[b].tolist(), flush=True)
    seen |= set(bad.tolist())
    if len(seen) > 40: break
