#!/usr/bin/env python
"""Shader-clock stage split PER TASK inside the bench workload (MT50 @ 4096 envs, episode phases staggered like bench.py): which
scene's waves set the launch time under the contention of the full batch.  Needs the -DMW_SOLVER_TIMING build
(metaworld_amd/libmwgpu_timing.so, tools/build_variants.sh).  Per task: max / mean over its environments of the cycles per step
in each pipeline stage.  usage: tools/mix_timing.py [steps=100] [precision=fp64]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd import native  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
prec = sys.argv[2] if len(sys.argv) > 2 else "fp64"
lib = native.load("mw_", os.path.join(ROOT, "metaworld_amd", os.environ.get("MW_LIB", "libmwgpu_timing.so")))
N = 4096
env = MetaWorldGpuVectorEnv("MT50", num_envs=N, seed=42, use_one_hot=True, precision=prec, lib=lib)
env.reset()
env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (64, N, 4)).astype(np.float32))
env.ctx.set_episode_phase((np.arange(N, dtype=np.int64) * 7919 % 500).astype(np.int32))
env.ctx.step_resident(500)
ic0 = np.array([env.ctx.read_int(e, "icount") for e in range(N)])
ms = env.ctx.step_resident(steps) / steps
ic1 = np.array([env.ctx.read_int(e, "icount") for e in range(N)])
d = (ic1 - ic0)[:, 4:18].astype(np.float64) / steps * 16e-3          # kilo-cycles per step
names = ["warm", "Hasm", "chol", "MvJv", "lsrch", "update", "n_ls", "n_newt", "kin", "crb", "coll", "cons", "smooth", "solve"]
stage = d[:, 8:14]
tot = stage.sum(1)
tn = np.array(env.env_task_names)
print(f"{ms:.2f} ms/launch; kcyc/step per env: total = kin + crb + coll + cons + smooth + solve")
rows = []
for t in env.task_list:
    m = tn == t
    i = np.argmax(tot[m])
    rows.append((tot[m].max(), t, tot[m].mean(), stage[m][i]))
for mx, t, mean, st in sorted(rows, reverse=True):
    print(f"{t:30s} max {mx:7.0f} mean {mean:7.0f} | slowest env: " + " ".join(f"{k}:{v:.0f}" for k, v in zip(names[8:], st)))
# solver phases (kcyc / step; n_ls, n_newt: counts / step) and problem sizes, mean over the envs of a task
cnt = (ic1 - ic0)[:, 10:12].astype(np.float64) / steps
print("solver phases, mean over the envs of a task: kcyc/step " + " ".join(names[:6]) + " | per step: line-search evals, Newton iterations | now: ncon nefc")
for t in env.task_list:
    m = tn == t
    ph = d[m][:, :6].mean(0)
    print(f"{t:30s} " + " ".join(f"{v:6.0f}" for v in ph) + f" | {cnt[m][:, 0].mean():5.1f} {cnt[m][:, 1].mean():5.1f} | {ic1[m][:, 0].mean():5.1f} {ic1[m][:, 1].mean():5.1f}")
if os.environ.get("MW_MIX_NPZ"):          # raw per-env numbers for offline analysis (wave-time distribution, load balance)
    np.savez_compressed(os.environ["MW_MIX_NPZ"], d=d, task=tn, ms=ms, ncon=ic1[:, 0], nefc=ic1[:, 1], names=np.array(names))
if os.environ.get("MW_MIX_JSON"):
    import json
    json.dump({t: dict(max_kcyc=float(tot[tn == t].max()), mean_kcyc=float(tot[tn == t].mean())) for t in env.task_list}, open(os.environ["MW_MIX_JSON"], "w"), indent=1)
