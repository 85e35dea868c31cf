#!/usr/bin/env python
"""Which wave sets the time of a launch?  Inside the bench workload (MT50 @ 4096, staggered phases, timing build): K single steps,
the stage clocks of every env read before and after each -> per step the slowest env's stage split, and the distribution of the
per-step maximum vs the per-env means (the launch lasts as long as its slowest wave).  usage: tail_probe.py [K=24] [out.npz]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd import native  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lib = native.load("mw_", os.path.join(ROOT, "metaworld_amd", os.environ.get("MW_LIB", "libmwgpu_timing.so")))
N = 4096
env = MetaWorldGpuVectorEnv("MT50", num_envs=N, seed=42, use_one_hot=True, precision="fp64", lib=lib)
env.reset()
env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (64, N, 4)).astype(np.float32))
env.ctx.set_episode_phase((np.arange(N, dtype=np.int64) * 7919 % 500).astype(np.int32))
env.ctx.step_resident(520)
names = ["warm", "Hasm", "chol", "MvJv", "lsrch", "update", "n_ls", "n_newt", "kin", "crb", "coll", "cons", "smooth", "solve"]
tn = np.array(env.env_task_names)
prev = np.array([env.ctx.read_int(e, "icount") for e in range(N)])
rec, ms_all = [], []
for s in range(K):
    ms = env.ctx.step_resident(1)
    cur = np.array([env.ctx.read_int(e, "icount") for e in range(N)])
    d = (cur - prev)[:, 4:18].astype(np.float64) * 16e-3          # kcyc this step
    prev = cur
    tot = d[:, 8:14].sum(1)
    i = int(np.argmax(tot))
    rec.append(d); ms_all.append(ms)
    print(f"step {s:2d}: launch {ms:.2f} ms = {ms * 2.4e3:6.0f} kcyc | slowest env {i:4d} {tn[i]:26s} stages {tot[i]:6.0f} kcyc: " +
          " ".join(f"{k}:{v:.0f}" for k, v in zip(names[8:], d[i, 8:14])) + f" | newton its {d[i, 7] / 16e-3:.0f} ls evals {d[i, 6] / 16e-3:.0f} | p50 {np.median(tot):.0f} p99 {np.quantile(tot, 0.99):.0f}", flush=True)
if len(sys.argv) > 2:
    np.savez_compressed(sys.argv[2], d=np.array(rec), ms=np.array(ms_all), task=tn)
