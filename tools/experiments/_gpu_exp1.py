import sys, time
sys.path.insert(0,'.')
import numpy as np
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
def run(label, **kw):
    env = MetaWorldGpuVectorEnv(**kw)
    env.reset()
    n=env.num_envs
    env.ctx.upload_actions(np.random.default_rng(0).uniform(-1,1,(32,n,4)).astype(np.float32))
    env.ctx.step_resident(3)
    ms = env.ctx.step_resident(15)/15
    print(f"{label:40s} n={n:5d} {ms:8.2f} ms/step  {n/ms*1e3:10.0f} steps/s", flush=True)
    env.close()
run("MT50 50 envs", benchmark="MT50", num_envs=50, seed=0, use_one_hot=True)
run("MT50 4096 envs", benchmark="MT50", num_envs=4096, seed=0, use_one_hot=True)
run("MT10 820 envs", benchmark="MT10", num_envs=820, seed=0, use_one_hot=True)
run("box-close 4096", benchmark="MT1", env_name="box-close-v3", num_envs=4096, seed=0)
run("box-close 64", benchmark="MT1", env_name="box-close-v3", num_envs=64, seed=0)
run("reach 4096", benchmark="MT1", env_name="reach-v3", num_envs=4096, seed=0)
run("reach 16384", benchmark="MT1", env_name="reach-v3", num_envs=16384, seed=0)
run("reach 65536", benchmark="MT1", env_name="reach-v3", num_envs=65536, seed=0, maxcon=16, maxefc=64)
