#!/usr/bin/env python
"""Host profile build (-DMW_PROFILE): narrow-phase statistics per geom-type pair over a window late in a random-action episode.
usage: g++ -O2 -std=c++17 -fPIC -shared -fopenmp -ffp-contract=off -DMW_PROFILE -o /tmp/libmw_prof.so tests/host_harness.cpp -lrt
       tools/experiments/pairstat.py task [first=250] [steps=100]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MW_NSUB", "1")
from metaworld_amd import native
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
lib = native.load("mwh_", "/tmp/libmw_prof.so")
dll = C.CDLL("/tmp/libmw_prof.so")
task = sys.argv[1]; first = int(sys.argv[2]) if len(sys.argv) > 2 else 250; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
n = 8
env = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision="fp64", lib=lib)
env.reset()
acts = np.random.default_rng(0).uniform(-1, 1, (64, 82, 4)).astype(np.float32)[:, :n]
out = (C.c_long * 256)(); cnt = (C.c_long * 8)(); prof = (C.c_double * 8)()
for t in range(first):
    env.step(acts[t % 64])
dll.mwh_pairstat(out, 1); dll.mwh_counters(cnt, 1); dll.mwh_profile(prof, 1)
for t in range(first, first + steps):
    env.step(acts[t % 64])
dll.mwh_pairstat(out, 1); dll.mwh_counters(cnt, 1); dll.mwh_profile(prof, 1)
names = ["plane", "?1", "sphere", "capsule", "?4", "cylinder", "box", "mesh"]
ps = np.array(out[:]).reshape(8, 8, 4)
print(task, "per env-step:  stage seconds (kin crb coll cons smooth solve)", np.round(np.array(prof[:6]) / (n * steps) * 1e3, 3), "ms")
print("  counters/env-step: support calls", cnt[3] / (n * steps), "hill steps", cnt[4] / (n * steps), "mpr calls", cnt[5] / (n * steps), "portal its", cnt[6] / (n * steps), "pairs", cnt[7] / (n * steps))
for a in range(8):
    for b in range(8):
        if ps[a, b, 0]:
            c = ps[a, b] / (n * steps)
            print(f"  {names[a]:8s}-{names[b]:8s} calls {c[0]:7.2f}  portal its {c[1]:8.2f} ({c[1] / max(c[0], 1e-9):5.1f}/call)  hill steps {c[2]:8.2f}  re-shot mpr runs {c[3]:6.2f} ({c[3] / max(c[0], 1e-9):4.1f}/call)")
hist = (C.c_long * 256)()
dll.mwh_hist(hist, 0)
H = np.array(hist[:]).reshape(4, 64)
print("  portal iterations per mpr() call (histogram, whole run):", {i: int(v) for i, v in enumerate(H[2]) if v})
print("  re-shoot rounds per mpr_refined():", {i: int(v) for i, v in enumerate(H[3]) if v})
