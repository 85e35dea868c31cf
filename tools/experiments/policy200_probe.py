#!/usr/bin/env python
"""one-step-from-synchronised-state replay of a policy200 trace on the GPU library (or MW_LIB_OVERRIDE variant), printing the steps
over 1e-5 with the observation index of the worst deviation and the contact count: tools/experiments/policy200_probe.py <task>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from metaworld_amd import native
from tests.helpers import golden, make_env
lib = native.load()
task = sys.argv[1]
G = dict(golden(f"trace_policy200_{task}_seed42.npz"))
env = make_env(lib, task, n=1, precision="fp64")
ctx = env.ctx
ctx.reset(G["goal_idx"])
worst = []
for t in range(G["actions"].shape[1]):
    if t > 0:
        ctx.write(0, "qpos", G["qpos"][0, t - 1]); ctx.write(0, "qvel", G["qvel"][0, t - 1])
        ctx.write(0, "mocap", G["mocap"][0, t - 1]); ctx.write(0, "warm", G["warm"][0, t - 1])
        tk = ctx.read(0, "task"); tk[15:33] = G["obs"][0, t - 1][:18]; ctx.write(0, "task", tk)
    o, r, te, tr, su, info = ctx.step(G["actions"][:, t])
    d = np.abs(o - G["obs"][:, t])[0]
    dq = np.abs(ctx.read(0, "qpos") - G["qpos"][0, t])
    if d.max() > 1e-5:
        worst.append((t, f"{d.max():.1e}", int(d.argmax()), f"dq {dq.max():.1e}@{int(dq.argmax())}", int(ctx.read_int(0, "icount")[0]), int(ctx.read_int(0, "icount")[2])))
print(os.environ.get("MW_LANES_PER_BLOCK"), len(worst), worst[:10], ctx.status())
