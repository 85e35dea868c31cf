#!/usr/bin/env python
"""env-steps/s of the BASELINE.json configurations and of larger batches on one GPU, measured like bench.py measures: episode phases
staggered over the horizon, one untimed horizon of pre-roll, then 200 timed resident steps.  usage: config_table.py [precision ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
precs = sys.argv[1:] or ["fp64", "fp32"]
rows = [("MT1", "reach-v3", 4096), ("MT10", None, 10240), ("MT50", None, 4096), ("ML45-train", None, 2048), ("MT50", None, 8192),
        ("MT50", None, 16384), ("MT50", None, 32768), ("MT50", None, 65536)]
for prec in precs:
    for bench, name, n in rows:
        if n > int(os.environ.get("MW_MAX_ENVS", "65536")) or (prec == "fp32" and n > 16384 and "MW_MAX_ENVS" not in os.environ):
            continue
        t0 = time.time()
        env = MetaWorldGpuVectorEnv(bench, name, num_envs=n, seed=42, use_one_hot=bench != "MT1", precision=prec)
        env.reset()
        setup = time.time() - t0
        env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (64, n, 4)).astype(np.float32))
        env.ctx.set_episode_phase((np.arange(n, dtype=np.int64) * 7919 % 500).astype(np.int32))
        env.ctx.step_resident(500)
        ms = env.ctx.step_resident(200) / 200
        st = env.status()
        print(f"{bench:10s} {name or '':10s} {n:6d} envs {prec}: {ms:7.2f} ms/step  {n / ms:8.1f} k env-steps/s   flags {st['flags']} (setup {setup:.1f} s)", flush=True)
        env.close()
