#!/usr/bin/env python
"""The check of tests/test_gpu_fullsize.py::test_bench_states_match_the_oracle for a library variant / MW_CHAIN_LDS setting, listing
every sampled env whose 5-substep deviation from the synchronised oracle exceeds 1e-7, and saving the synchronised states of the
worst ones (for a host-build replay).  usage: [MW_LIB=...] tools/experiments/bench_state_probe.py out.npz"""
import os
import sys
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from metaworld_amd import native  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402
from tests.test_gpu_fullsize import _oracle_synced_to  # noqa: E402
from tools.dump_bench_states import pick_envs  # noqa: E402

lib = native.load("mw_", os.path.join(ROOT, "metaworld_amd", os.environ.get("MW_LIB", "libmwgpu.so")))
n = 4096
env = MetaWorldGpuVectorEnv("MT50", num_envs=n, seed=42, use_one_hot=True, precision="fp64", lib=lib)
bench.prepare(env, SimpleNamespace(no_stagger=False, warmup=20, allow_status=False), 0)
elapsed = (np.arange(n, dtype=np.int64) * 7919 + bench.HORIZON + 20) % bench.HORIZON
chosen = pick_envs(env, elapsed)
cols = ("qpos", "qvel", "warm", "mocap", "ctrl", "reloc")
before = {e: {c: env.ctx.read(e, c) for c in cols} for e in chosen}
synced = [(e, env.env_task_names[e], _oracle_synced_to(env.ctx, e, env.env_task_names[e])) for e in chosen]
env.ctx.debug("substeps", 5)
rows = []
for e, name, (om, d) in synced:
    d.step(5)
    ic = env.ctx.read_int(e, "icount")
    eq, ev = np.abs(env.ctx.read(e, "qpos") - d.qpos).max(), np.abs(env.ctx.read(e, "qvel") - d.qvel).max()
    rows.append((eq, ev, name, e, int(ic[0]), d.ncon, int(ic[1]), d.nefc, int(ic[2])))
errs = np.array([r[0] for r in rows])
print(os.environ.get("MW_LIB", "libmwgpu.so"), "chain", os.environ.get("MW_CHAIN_LDS", "auto"), "quantiles 0.5 0.9 0.99 1.0:", np.quantile(errs, [0.5, 0.9, 0.99, 1.0]), "status", env.status())
for r in sorted(rows, reverse=True)[:12]:
    print("  qpos %.2e qvel %.2e %-28s env %4d ncon %d/%d nefc %d/%d niter %d" % r)
if len(sys.argv) > 1:
    worst = [r[3] for r in sorted(rows, reverse=True)[:8]]
    np.savez(sys.argv[1], envs=np.array(worst), tasks=np.array([env.env_task_names[e] for e in worst]),
             **{f"{c}_{e}": before[e][c] for e in worst for c in cols},
             **{f"after_qpos_{e}": env.ctx.read(e, "qpos") for e in worst}, **{f"after_qvel_{e}": env.ctx.read(e, "qvel") for e in worst})
