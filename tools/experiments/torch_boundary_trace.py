#!/usr/bin/env python
"""30 steps of MetaWorldTorchVectorEnv at the bench workload, for a rocprofv3 kernel trace of the boundary's per-step timeline"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from metaworld_amd.torch_env import MetaWorldTorchVectorEnv
env = MetaWorldTorchVectorEnv("MT50", num_envs=4096, seed=42, use_one_hot=True, precision="fp64")
env.reset()
env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (64, 4096, 4)).astype(np.float32))
env.ctx.set_episode_phase((np.arange(4096, dtype=np.int64) * 7919 % 500).astype(np.int32))
env.step_resident(500)
acts = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (8, 4096, 4)).astype(np.float32)).to(env.device)
for t in range(5):
    env.step(acts[t % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(30):
    o, r, te, tr, info = env.step(acts[t % 8])
float(r.sum().item())
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / 30 * 1e3)
