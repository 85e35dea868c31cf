#!/usr/bin/env python
"""Capacity planning on the GPU: the contacts / constraint rows every scene WANTS (icount demand slots, running maxima) over
whole episodes of random actions at the benchmark's own size, with the capacities opened wide so that nothing is dropped on
the way.  Writes metaworld_amd/data/model_caps.json = measured maximum x 2 (rounded up to multiples of 8; the file is written by hand from the printed demand).
usage: python tools/measure_caps_gpu.py [envs=4096] [steps=1500] [out=gpurun_out/model_caps_measured.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd import tasks as T  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "model_caps_measured.json")
demand = {}
# two generic seeds + the bench workload itself (bench.py: seed 42, action stream rng(0) of length 64, staggered episode phases), fp32 and fp64
RUNS = [(1, 1, 97, False, "fp32"), (42, 0, 64, True, "fp32"), (42, 0, 64, True, "fp64")]
for seed, aseed, alen, stagger, prec in RUNS:
    env = MetaWorldGpuVectorEnv("MT50", num_envs=n, seed=seed, use_one_hot=True, precision=prec, maxcon=192, maxefc=768)
    env.reset()
    env.ctx.upload_actions(np.random.default_rng(aseed).uniform(-1, 1, (alen, n, 4)).astype(np.float32))
    if stagger:
        env.ctx.set_episode_phase((np.arange(n, dtype=np.int64) * 7919 % 500).astype(np.int32))
    env.ctx.step_resident(steps)
    print("seed", seed, prec, "status", env.ctx.status(), flush=True)
    ic = env.ctx.read_int_all("icount", 24) if hasattr(env.ctx, "read_int_all") else np.array([env.ctx.read_int(e, "icount", 24) for e in range(n)])
    for e, name in enumerate(env.env_task_names):
        m = T.TASK_CONST[name]["model"]
        d = demand.setdefault(m, [0, 0])
        d[0] = max(d[0], int(ic[e][20])); d[1] = max(d[1], int(ic[e][21]))
    env.close()
cur = T.MODEL_CAPS
caps = {m: {"maxcon": int(-(-d[0] * 3 // 2) // 8 * 8 + 8), "maxefc": int(-(-d[1] * 3 // 2) // 8 * 8 + 8), "measured_ncon": d[0], "measured_nefc": d[1]}
        for m, d in sorted(demand.items())}
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as f:
    json.dump(caps, f, indent=1)
for m, c in caps.items():
    tight = "  <-- current capacity below 2x demand" if (cur[m]["maxcon"] < 2 * c["measured_ncon"] or cur[m]["maxefc"] < 2 * c["measured_nefc"]) else ""
    print(f"{m:36s} wanted ncon {c['measured_ncon']:3d} nefc {c['measured_nefc']:3d}  (shipped maxcon {cur[m]['maxcon']:3d} maxefc {cur[m]['maxefc']:3d}){tight}")
