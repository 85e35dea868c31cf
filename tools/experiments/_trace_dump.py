#!/usr/bin/env python
"""Dump (ncon, nefc, qpos) per step for a task under random actions: used to compare the HIP library with the host
harness build of the same lane code (tools/_trace_dump.py <gpu|host> task steps out.npz)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd import native
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
which, task, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
if which == "host":
    import __graft_entry__ as g
    lib = native.load("mwh_", g.build_host_harness())
else:
    lib = native.load()
n = 8
env = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision=sys.argv[5] if len(sys.argv) > 5 else "fp32", lib=lib)
env.reset()
acts = np.random.default_rng(0).uniform(-1, 1, (64, n, 4)).astype(np.float32)
ic, qp, ob = [], [], []
for t in range(steps):
    o, r, te, tr, info = env.step(acts[t % 64])
    ic.append([env.ctx.read_int(e, "icount")[:4] for e in range(n)])
    qp.append([env.ctx.read(e, "qpos") for e in range(n)])
    ob.append(o.copy())
np.savez(out, ic=np.array(ic), qpos=np.array(qp), obs=np.array(ob))
