#!/usr/bin/env python
"""Per-iteration solver trace (gn, cost, |s|, alpha, d1, d2, s[0], qfrc[1]) of the failing solve found by find_fail.py: device
matrix-core path vs the host build running the same FUSED control flow on its plain-loop restatement of the pass."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd import native  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

task, step, n = sys.argv[1], int(sys.argv[2]), 6
good = native.load("mw_", os.path.join(ROOT, "metaworld_amd", "libmwgpu_nofused.so"))
dev = native.load("mw_", os.path.join(ROOT, "metaworld_amd", "libmwgpu_trace.so"))
host = native.load("mwh_", os.path.join(ROOT, "tests", "_build", "libmw_hostsim_trace.so"))
acts = np.random.default_rng(0).uniform(-1, 1, (64, 82, 4)).astype(np.float32)[:, :1]
eg = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision="fp64", lib=good)
eg.reset()
for t in range(step):
    eg.ctx.step(np.repeat(acts[t % 64], n, axis=0))
state = [eg.ctx.read(e, "state") for e in range(n)]
a = np.repeat(acts[step % 64], n, axis=0)
os.environ["MW_NSUB"] = "16"
for name, lib in (("device", dev), ("host", host)):
    env = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision="fp64", lib=lib)
    env.ctx.reset(np.zeros(n, dtype=np.int32))
    for e in range(n):
        env.ctx.write(e, "state", state[e])
        mc = env.ctx.read(e, "mocap") + np.clip(a[e, :3], -1, 1).astype(np.float32) * np.float32(0.01)
        env.ctx.write(e, "mocap", mc); env.ctx.write(e, "ctrl", [a[e, 3], -a[e, 3]])
        env.ctx.write(e, "dbg", np.zeros(64))
    env.ctx.debug("forward")
    np.set_printoptions(linewidth=200, precision=6)
    print(name, "icount", list(env.ctx.read_int(0, "icount", 24)[:3]), "\n", env.ctx.read(0, "dbg").reshape(8, 8)[:4])
    env.close()
