#!/usr/bin/env python
"""The reference's scripted policies (metaworld/policies, unmodified, needs /root/reference) driving the DEVICE lane
programs closed loop: host build (tests/_build/libmw_hostsim.so) by default, `gpu` as first argument for libmwgpu.so is
not possible on the GPU box (no reference there).  5 goals per task; the reference's gate is 80 %."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import numpy as np
from tests.helpers import make_env
from metaworld_amd import native, tasks as T
import __graft_entry__ as g
lib = native.load("mwh_", g.build_host_harness())
from oracle import refshim
refshim.install()
from metaworld.policies import ENV_POLICY_MAP
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
ok = 0
for task in T.ALL_V3:
    env = make_env(lib, task, n=5, precision=prec)
    obs = env.ctx.reset(np.arange(5)).copy()
    pol = [ENV_POLICY_MAP[task]() for _ in range(5)]
    done = np.zeros(5, dtype=bool); first = np.full(5, -1)
    for t in range(500):
        a = np.stack([np.clip(p.get_action(o[:39].copy()), -1, 1) for p, o in zip(pol, obs)]).astype(np.float32)
        obs, r, te, tr, su, info = env.ctx.step(a); obs = obs.copy()
        first[(first < 0) & su.astype(bool)] = t
        done |= su.astype(bool)
        if done.all():
            break
    env.close()
    ok += done.sum() >= 4
    print(f"{task:30s} {prec} succ {int(done.sum())}/5  first-success steps {first.tolist()}", flush=True)
print("tasks with >= 4/5:", ok, "of", len(T.ALL_V3))
