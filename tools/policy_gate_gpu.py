#!/usr/bin/env python
"""The reference's scripted-policy gate (tests/metaworld/envs/mujoco/sawyer_xyz/test_scripted_policies.py: MT1(task, seed=42), all 50
goals, success within 500 steps, >= 0.8 per task) on the GPU: 50 envs per task, env j on goal j, scripted policies on the device
(mw_policy_rollout, bit-identical to the reference's numpy policies: tests/test_device_policies.py), goal visible (MT tasks
carry partially_observable = False, metaworld/__init__.py:80-103).
usage: tools/policy_gate_gpu.py [precision=fp64] [task ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd import tasks as T  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

lib = None
if os.environ.get("MW_HOST_HARNESS"):          # CPU dry run of this script on the host build of the lane programs
    import __graft_entry__ as g
    from metaworld_amd import native
    lib = native.load("mwh_", g.build_host_harness())
prec = sys.argv[1] if len(sys.argv) > 1 else "fp64"
names = sys.argv[2:] or T.ALL_V3
ok = 0
for name in names:
    env = MetaWorldGpuVectorEnv("MT1", name, num_envs=50, seed=42, precision=prec, partially_observable=False, max_episode_steps=500, lib=lib)
    pid = np.full(50, T.ALL_V3.index(name), dtype=np.int32)
    sched = np.stack([np.arange(50), np.arange(50)])          # one episode per env, on its own goal
    ep, su, ms = env.ctx.policy_rollout(pid, sched, 500)
    flags = env.status()["flags"]
    env.close()
    assert (ep == 1).all()
    ok += su.sum() >= 40
    print(f"{name:30s} {prec} success {int(su.sum()):2d}/50  failed goals {np.flatnonzero(su == 0).tolist()}  status flags {flags}", flush=True)
print(f"tasks passing the 80 % gate (50 goals, {prec}): {ok} of {len(names)}")
