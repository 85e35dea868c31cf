#!/usr/bin/env python
"""GPU time of each stage of one dynamics evaluation (kinematics, CRB, collision, constraint rows, smooth forces, solver)
for a task, measured by running stage prefixes through the debug entry point."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
warm = int(os.environ.get("MW_WARM", "10"))
reps = 20
names = ["kin", "crb", "smooth", "coll", "cons", "solve"]          # the order of forward_dynamics (mw_phys.hpp)
for task in sys.argv[2:]:
    env = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision="fp32")
    env.reset()
    rng = np.random.default_rng(0)
    for _ in range(warm):
        env.step(rng.uniform(-1, 1, (n, 4)).astype(np.float32))
    prev, out = 0.0, []
    for k in range(6):
        env.ctx.debug(10 + k, 2)
        t0 = time.perf_counter()
        env.ctx.debug(10 + k, reps)
        dt = (time.perf_counter() - t0) / reps * 1e3
        out.append(dt - prev)
        prev = dt
    print(f"{task:26s} n={n} ms/eval {prev:6.3f} | " + " ".join(f"{a}:{b:6.3f}" for a, b in zip(names, out)), flush=True)
    env.close()
