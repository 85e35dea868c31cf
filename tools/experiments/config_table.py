#!/usr/bin/env python
"""env-steps/s of the BASELINE.json configurations and of larger batches on one GPU (200 timed steps after 20)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
rows = [("MT1", "reach-v3", 4096, "fp32"), ("MT10", None, 10240, "fp32"), ("MT50", None, 4096, "fp32"), ("MT50", None, 8192, "fp32"),
        ("MT50", None, 16384, "fp32"), ("MT50", None, 32768, "fp32"), ("MT50", None, 65536, "fp32"), ("ML45-train", None, 2048, "fp32"),
        ("MT50", None, 4096, "fp64")]
for bench, name, n, prec in rows:
    t0 = time.time()
    env = MetaWorldGpuVectorEnv(bench, name, num_envs=n, seed=42, use_one_hot=bench != "MT1", precision=prec)
    env.reset()
    setup = time.time() - t0
    env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (64, n, 4)).astype(np.float32))
    env.ctx.step_resident(20)
    ms = env.ctx.step_resident(200) / 200
    print(f"{bench:10s} {name or '':10s} {n:6d} envs {prec}: {ms:7.2f} ms/step  {n / ms * 1e3 / 1e3:8.1f} k env-steps/s   (setup {setup:.1f} s)", flush=True)
    env.close()
