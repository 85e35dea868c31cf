#!/usr/bin/env python
"""Find the first step at which the product library's rollout (MW_SAME action stream) turns unstable, then replay that step
from the healthy library's state of the step before, substep by substep, against the healthy library."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd import native  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

task, prec, n = sys.argv[1], "fp64", 6
good = native.load("mw_", os.path.join(ROOT, "metaworld_amd", "libmwgpu_nofused.so"))
bad = native.load("mw_", os.path.join(ROOT, "metaworld_amd", os.environ.get("MW_LIB", "libmwgpu.so")))
acts = np.random.default_rng(0).uniform(-1, 1, (64, 82, 4)).astype(np.float32)[:, :1]
eg = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision=prec, lib=good)
eb = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision=prec, lib=bad)
eg.reset(); eb.reset()
for t in range(150):
    a = np.repeat(acts[t % 64], n, axis=0)
    state = [eg.ctx.read(e, "state") for e in range(n)]
    eg.ctx.step(a); eb.ctx.step(a)
    err = max(np.abs(eg.ctx.read(e, "qpos") - eb.ctx.read(e, "qpos")).max() for e in range(n))
    st = eb.ctx.status(clear=True)
    if err > 1e-6 or st["flags"]:
        print(f"{task}: first divergence at step {t}: qpos err {err:.3e} status {st}")
        # replay the step from the healthy state, substep by substep, in both libraries
        for env in (eg, eb):
            env.ctx.reset(np.zeros(n, dtype=np.int32))
            for e in range(n):
                env.ctx.write(e, "state", state[e])
            # set_xyz_action by hand
            for e in range(n):
                mc = env.ctx.read(e, "mocap") + np.clip(a[e, :3], -1, 1).astype(np.float32) * np.float32(0.01)
                env.ctx.write(e, "mocap", mc); env.ctx.write(e, "ctrl", [a[e, 3], -a[e, 3]])
        for k in range(6):
            for env in (eg, eb):
                env.ctx.debug("substeps", 1) if k < 5 else env.ctx.debug("forward")
            err = max(np.abs(eg.ctx.read(e, "qacc") - eb.ctx.read(e, "qacc")).max() for e in range(n))
            ig, ib = eg.ctx.read_int(0, "icount", 24), eb.ctx.read_int(0, "icount", 24)
            print(f"   substep {k}: |qacc diff| {err:.3e}  healthy ncon/nefc/niter {list(ig[:3])}  fused {list(ib[:3])} why {ib[22]}  |qacc| {np.abs(eg.ctx.read(0,'qacc')).max():.3e}")
        break
else:
    print(task, "no divergence in 150 steps")
