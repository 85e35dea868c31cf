#!/usr/bin/env python
"""Shader-clock breakdown of the Newton solver's phases on the GPU (needs the -DMW_SOLVER_TIMING build of the
library: tools/build_timing_lib.sh -> gpurun_out/libmwgpu_timing.so is NOT used; the library is built in-tree as
metaworld_amd/libmwgpu_timing.so).  Prints per-lane-max cycles per substep-equivalent evaluation."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd import native  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

lib = native.load("mw_", os.path.join(ROOT, "metaworld_amd", "libmwgpu_timing.so"))
n = int(sys.argv[1])
names = ["warm", "Hasm", "chol", "MvJv", "lsrch", "update", "n_ls", "n_newton"]
for task in sys.argv[2:]:
    env = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision="fp32", lib=lib)
    env.reset()
    rng = np.random.default_rng(0)
    for _ in range(int(os.environ.get("MW_WARM", "25"))):
        env.step(rng.uniform(-1, 1, (n, 4)).astype(np.float32))
    ic0 = np.array([env.ctx.read_int(e, "icount") for e in range(n)])
    env.ctx.debug(15, 10)          # 10 full dynamics evaluations on the current state
    ic1 = np.array([env.ctx.read_int(e, "icount") for e in range(n)])
    d = (ic1 - ic0)[:, 4:].astype(np.float64) / 10
    d[:, :6] *= 16 / 1e3           # kilo-cycles
    print(f"{task:22s} nefc max {ic1[:,1].max():3d} | max over lanes, kcycles/eval: " +
          " ".join(f"{k}:{v:7.1f}" for k, v in zip(names, d.max(0))) + f" | max over lanes n_ls {d[:,6].max():.1f} n_newton {d[:,7].max():.1f} mean {d[:,7].mean():.2f}", flush=True)
    env.close()
