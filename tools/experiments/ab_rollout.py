#!/usr/bin/env python
"""random-action rollout of one task with a library variant: status flags, step time, a checksum of the final state"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd import native  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

lib = native.load("mw_", os.path.join(ROOT, "metaworld_amd", os.environ.get("MW_LIB", "libmwgpu.so")))
n, steps = int(os.environ.get("MW_N", "82")), int(os.environ.get("MW_STEPS", "150"))
for task in sys.argv[1:]:
    for prec in os.environ.get("MW_PRECS", "fp64,fp32").split(","):
        env = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision=prec, lib=lib)
        env.reset()
        acts = np.random.default_rng(0).uniform(-1, 1, (64, n, 4)).astype(np.float32)
        if os.environ.get("MW_SAME"):          # every env the same action stream (and, MT1 with one seed, the same goals): no divergence inside a wave
            acts[:] = acts[:, :1]
        env.ctx.upload_actions(acts)
        first_bad = None
        ms = []
        for w in range(steps // 10):
            ms.append(env.ctx.step_resident(10) / 10)
            st = env.ctx.status()
            if st["flags"] & 4 and first_bad is None:
                first_bad = (w * 10, st)
        q = np.array([env.ctx.read(e, "qpos") for e in range(0, n, 9)])
        print(os.environ.get("MW_LIB", "libmwgpu.so"), "lpb", os.environ.get("MW_LANES_PER_BLOCK", "auto"), task, prec, f"ms/step {np.mean(ms):.2f} (last {ms[-1]:.2f})",
              "status", env.ctx.status(), "first unstable window", first_bad, "qpos checksum", float(np.nansum(np.abs(q))), flush=True)
        env.close()
