#!/usr/bin/env python
"""env-steps/s through the full VectorEnv.step boundary (host numpy actions in, obs/reward/flags/infos out per step)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
n = 4096
env = MetaWorldGpuVectorEnv("MT50", num_envs=n, seed=42, use_one_hot=True, precision="fp32")
env.reset()
acts = np.random.default_rng(0).uniform(-1, 1, (64, n, 4)).astype(np.float32)
for t in range(20):
    env.step(acts[t % 64])
t0 = time.perf_counter()
for t in range(300):
    obs, rew, term, trunc, info = env.step(acts[t % 64])
dt = time.perf_counter() - t0
print(f"VectorEnv.step (host in/out): {dt / 300 * 1e3:.2f} ms/step  {n * 300 / dt / 1e3:.1f} k env-steps/s")
t0 = time.perf_counter()
for t in range(300):
    env.ctx.step(acts[t % 64], env._next_goal)
dt = time.perf_counter() - t0
print(f"mw_step C ABI only (host in/out): {dt / 300 * 1e3:.2f} ms/step  {n * 300 / dt / 1e3:.1f} k env-steps/s")
env.close()

# device-resident boundary: actions drawn on the GPU by torch, outputs stay tensors (mw_step_device)
import torch
from metaworld_amd.torch_env import MetaWorldTorchVectorEnv
env = MetaWorldTorchVectorEnv("MT50", num_envs=n, seed=42, use_one_hot=True, precision="fp32")
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
for t in range(20):
    env.step(torch.rand((n, 4), device="cuda", generator=g) * 2 - 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(300):
    obs, rew, term, trunc, info = env.step(torch.rand((n, 4), device="cuda", generator=g) * 2 - 1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"MetaWorldTorchVectorEnv.step (device in/out): {dt / 300 * 1e3:.2f} ms/step  {n * 300 / dt / 1e3:.1f} k env-steps/s")
env.close()
