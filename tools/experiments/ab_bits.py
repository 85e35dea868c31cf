#!/usr/bin/env python
"""Bitwise A/B of the host build of the lane programs: random-action rollouts of a few contact-rich tasks (both precisions, sub-lane
emulation on), every observation / reward / state hashed.  A restructuring that must not change any result (load batching, LDS
staging, reordered independent stages) prints the same digests before and after:
    python tools/experiments/ab_bits.py > /tmp/a.txt;  <edit, rebuild>;  python tools/experiments/ab_bits.py > /tmp/b.txt;  diff /tmp/a.txt /tmp/b.txt"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
from metaworld_amd import native  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

lib = native.load("mwh_", g.build_host_harness())
tasks = sys.argv[1:] or ["reach-v3", "box-close-v3", "hammer-v3", "door-unlock-v3", "plate-slide-back-v3", "stick-pull-v3", "assembly-v3", "soccer-v3"]
for nsub in ("1", "8"):
    os.environ["MW_NSUB"] = nsub
    for prec in ("fp64", "fp32"):
        for t in tasks:
            env = MetaWorldGpuVectorEnv("MT1", t, num_envs=4, seed=3, precision=prec, lib=lib, max_episode_steps=150)
            env.reset()
            rng = np.random.default_rng(1)
            h = hashlib.sha256()
            for s in range(170):
                o, r, te, tr, info = env.step(rng.uniform(-1, 1, (4, 4)).astype(np.float32))
                h.update(o.tobytes()); h.update(r.tobytes())
            for e in range(4):
                for c in ("qpos", "qvel", "warm"):
                    h.update(env.ctx.read(e, c).tobytes())
            print(f"nsub {nsub} {prec} {t:24s} {h.hexdigest()[:20]} status {env.status()['flags']}", flush=True)
            env.close()
