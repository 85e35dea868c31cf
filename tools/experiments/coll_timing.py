#!/usr/bin/env python
"""Collision-stage split inside the bench workload (MT50 @ 4096, staggered phases) from the -DMW_SOLVER_TIMING -DMW_COLL_TIMING
build (metaworld_amd/libmwgpu_colltiming.so): per task, for the environment with the most collision cycles: mid phase (candidate
list) vs narrow phase cycles per step, candidate pairs and narrow-phase rounds per step."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd import native
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
prec = sys.argv[2] if len(sys.argv) > 2 else "fp64"
lib = native.load("mw_", os.path.join(ROOT, "metaworld_amd", os.environ.get("MW_LIB", "libmwgpu_colltiming.so")))
N = 4096
env = MetaWorldGpuVectorEnv("MT50", num_envs=N, seed=42, use_one_hot=True, precision=prec, lib=lib)
env.reset()
env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (64, N, 4)).astype(np.float32))
env.ctx.set_episode_phase((np.arange(N, dtype=np.int64) * 7919 % 500).astype(np.int32))
env.ctx.step_resident(500)
ic0 = np.array([env.ctx.read_int(e, "icount") for e in range(N)])
ms = env.ctx.step_resident(steps) / steps
ic1 = np.array([env.ctx.read_int(e, "icount") for e in range(N)])
d = (ic1 - ic0)[:, 4:18].astype(np.float64) / steps
mid, narrow, ncand, rounds, coll = d[:, 0] * 16e-3, d[:, 1] * 16e-3, d[:, 2], d[:, 3], d[:, 10] * 16e-3
br = d[:, 4:8] * 16e-3          # per-branch kcyc / step inside collide_pair: box-box, portal refinement, other closed forms, face upgrade
tn = np.array(env.env_task_names)
if os.environ.get("MW_COLL_NPZ"):          # raw per-env numbers (wave-level analysis offline)
    np.savez_compressed(os.environ["MW_COLL_NPZ"], d=d, task=tn, ms=ms)
print(f"{ms:.2f} ms/launch")
rows = []
for t in env.task_list:
    m = np.flatnonzero(tn == t)
    i = m[np.argmax(coll[m])]
    rows.append((coll[i], t, mid[i], narrow[i], ncand[i], rounds[i], coll[m].mean(), mid[m].mean(), narrow[m].mean(), ncand[m].mean(), rounds[m].mean(), br[i], br[m].mean(0)))
for r in sorted(rows, reverse=True):
    print(f"{r[1]:30s} slowest env: coll {r[0]:6.0f} kcyc/step = mid {r[2]:5.0f} + narrow {r[3]:6.0f}; cand/step {r[4]:5.1f} rounds/step {r[5]:4.1f} -> {r[3] / max(r[5], 1e-9):6.0f} kcyc/round | mean env: coll {r[6]:6.0f} mid {r[7]:5.0f} narrow {r[8]:6.0f} cand {r[9]:5.1f} rounds {r[10]:4.1f}"
          f" | branches (box-box, mpr, closed forms, face upgrade) slowest env " + " ".join(f"{v:.0f}" for v in r[11]) + " mean " + " ".join(f"{v:.0f}" for v in r[12]))
