#!/usr/bin/env python
"""Shader-clock breakdown of one VectorEnv.step on the GPU while really stepping (random actions), from the
-DMW_SOLVER_TIMING build of the library (metaworld_amd/libmwgpu_timing.so, see DESIGN.md "measuring").
Per window of steps: max over lanes of the cycles spent in each pipeline stage and solver phase, per step."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd import native  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

lib = native.load("mw_", os.path.join(ROOT, "metaworld_amd", os.environ.get("MW_LIB", "libmwgpu_timing.so")))
n = int(sys.argv[1])
win = int(os.environ.get("MW_WIN", "50"))
nwin = int(os.environ.get("MW_NWIN", "4"))
prec = os.environ.get("MW_PREC", "fp32")
names = ["warm", "Hasm", "chol", "MvJv", "lsrch", "update", "n_ls", "n_newt", "kin", "crb", "coll", "cons", "smooth", "solve"]
for task in sys.argv[2:]:
    env = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision=prec, lib=lib)
    env.reset()
    acts = np.random.default_rng(0).uniform(-1, 1, (64, n, 4)).astype(np.float32)
    if os.environ.get("MW_SAME_ACTIONS"):          # every env identical (same goal stream, same actions): no divergence between lanes
        acts[:] = acts[:, :1]
    env.ctx.upload_actions(acts)
    for w in range(nwin):
        ic0 = np.array([env.ctx.read_int(e, "icount") for e in range(n)])
        ms = env.ctx.step_resident(win) / win
        ic1 = np.array([env.ctx.read_int(e, "icount") for e in range(n)])
        d = (ic1 - ic0)[:, 4:18].astype(np.float64) / win
        scale = np.array([16e-3] * 6 + [1, 1] + [16e-3] * 6)      # kilo-cycles, counts
        d = d * scale
        print(f"{task:18s} steps {w*win:3d}-{(w+1)*win:3d} {ms:6.2f} ms/step nefc<= {ic1[:,1].max():3d} | kcyc/step (max lane) " +
              " ".join(f"{k}:{v:.0f}" for k, v in zip(names, d.max(0))), flush=True)
    print("   status", env.ctx.status(), flush=True)
    env.close()
