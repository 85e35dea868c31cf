#!/usr/bin/env python
"""Which stage of which substep first produces a non-finite / wrong value in a given build of the library?
usage: MW_LIB=libmwgpu_timing.so python tools/experiments/fault_hunt.py <task> [precision=fp64] [n=8]
Replays `mj_resetData` + substeps through the debug entry point next to the oracle engine and reports the first substep whose
qpos differs, then re-runs that substep stage by stage (kinematics, crb, collision, constraints, smooth forces, solver) and lists
the columns that are non-finite after each prefix."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd import native, tasks as T  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402
from tests.helpers import oracle_for  # noqa: E402

task = sys.argv[1]
prec = sys.argv[2] if len(sys.argv) > 2 else "fp64"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lib = native.load("mw_", os.path.join(ROOT, "metaworld_amd", os.environ.get("MW_LIB", "libmwgpu.so")))
env = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision=prec, lib=lib)
obs = env.ctx.reset(np.zeros(n, dtype=np.int32))
print(task, prec, "reset obs finite:", bool(np.isfinite(obs).all()), "status", env.ctx.status(clear=True))
c = T.TASK_CONST[task]
om, d = oracle_for(c["model"])
d.mocap_pos[:] = c["hand_init_pos"]; d.mocap_quat[:] = [1, 0, 1, 0]; d.ctrl[:] = [-1, 1]
env.ctx.debug("reset_data")
for e in range(n):
    env.ctx.write(e, "mocap", c["hand_init_pos"]); env.ctx.write(e, "ctrl", [-1, 1])
cols = ["xpos", "xmat", "geom_xpos", "cdof", "qM", "qL", "bias", "smooth", "qacc_smooth", "con", "efcJ", "efcX", "qfrc_constraint", "qacc"]
bad_at = None
for k in range(300):
    state = [(env.ctx.read(e, "qpos"), env.ctx.read(e, "qvel"), env.ctx.read(e, "warm")) for e in range(n)]
    d.step(1)
    env.ctx.debug("substeps", 1)
    err = max(np.abs(env.ctx.read(e, "qpos") - d.qpos).max() if np.isfinite(env.ctx.read(e, "qpos")).all() else np.inf for e in range(n))
    if not err < 1e-6:
        bad_at = k
        print(f"substep {k}: qpos error {err} (ncon/nefc/niter/flags per env: {[list(env.ctx.read_int(e, 'icount', 20)[:4]) for e in range(n)]})")
        break
print("first bad substep:", bad_at)
if bad_at is not None:
    for e in range(n):          # rewind and replay stage by stage
        env.ctx.write(e, "qpos", state[e][0]); env.ctx.write(e, "qvel", state[e][1]); env.ctx.write(e, "warm", state[e][2])
    for stage, name in enumerate(["kin", "crb", "coll", "cons", "smooth", "solve"]):
        env.ctx.debug(10 + stage, 1)
        bad = []
        for col in cols:
            for e in range(n):
                v = env.ctx.read(e, col)
                nc, nf = env.ctx.read_int(e, "icount", 20)[:2]
                if col == "con": v = v[:26 * nc]
                if col == "efcJ": v = v[:len(state[e][1]) * nf]
                if col == "efcX": v = v[:15 * nf] if env.ctx._check(lib.column_size(env.ctx.ptr, 0, b"efcX")) % 15 == 0 else v[:11 * nf]
                if not np.isfinite(v).all():
                    bad.append((col, e, int(np.flatnonzero(~np.isfinite(v))[0])))
                    break
        print(f"after {name:7s}: non-finite columns (col, env, first index): {bad}   icount {list(env.ctx.read_int(0, 'icount', 20)[:4])}")
env.close()
