#!/usr/bin/env python
"""The launches of the bench workload that take several times the median (one ~24 ms launch per ~100-200 steps at MT50 @ 4096, round 5):
which environment is it, and where does its time go?  Timing build (libmwgpu_timing.so, tools/build_variants.sh): pass 1 finds the
slowest launch of a window of the resident loop (mw_launch_times), pass 2 replays the same deterministic sequence up to that launch and
reads every environment's stage clocks around it.  usage: outlier_probe.py [window=300]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd import native
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
W = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lib = native.load("mw_", os.path.join(ROOT, "metaworld_amd", os.environ.get("MW_LIB", "libmwgpu_timing.so")))
N = 4096


def start():
    env = MetaWorldGpuVectorEnv("MT50", num_envs=N, seed=42, use_one_hot=True, precision="fp64", lib=lib)
    env.reset()
    env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (64, N, 4)).astype(np.float32))
    env.ctx.set_episode_phase((np.arange(N, dtype=np.int64) * 7919 % 500).astype(np.int32))
    env.step_resident(500)
    return env


env = start()
env.step_resident(W)
t = np.asarray(env.ctx.launch_times())
order = np.argsort(-t)
print(f"window of {W} launches: median {np.median(t):.2f} ms, slowest {[(int(i), round(float(t[i]), 2)) for i in order[:5]]}", flush=True)
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # how many of the slowest launches to look into (in launch order)
ks = sorted(int(i) for i in order[:TOP])
env.close()
env = start()
done = 0
names = ["warm", "Hasm", "chol", "MvJv", "lsrch", "update", "n_ls", "n_newt", "kin", "crb", "coll", "cons", "smooth", "solve"]
tn = np.array(env.env_task_names)
for k in ks:
    if k > done:
        env.step_resident(k - done)
    prev = np.array([env.ctx.read_int(e, "icount") for e in range(N)])
    ms = env.step_resident(1)
    done = k + 1
    cur = np.array([env.ctx.read_int(e, "icount") for e in range(N)])
    d = (cur - prev)[:, 4:18].astype(np.float64) * 16e-3          # kcyc in this step
    tot = d[:, 8:14].sum(1)
    print(f"launch {k}: {ms:.2f} ms = {ms * 2.4e3:.0f} kcyc; per-env stage totals: p50 {np.median(tot):.0f} p99 {np.quantile(tot, 0.99):.0f} max {tot.max():.0f} kcyc")
    seen = set()
    for i in np.argsort(-tot):
        key = (tn[i], round(float(tot[i])))          # (the environments of one wave share its clocks: one line per wave)
        if key in seen:
            continue
        seen.add(key)
        if len(seen) > 5:
            break
        same = np.flatnonzero((tn == tn[i]) & (np.abs(tot - tot[i]) < 1))
        print(f"  wave of env {i:4d} {tn[i]:24s} total {tot[i]:7.0f} kcyc: " + " ".join(f"{n}:{v:.0f}" for n, v in zip(names[8:], d[i, 8:14])) +
              f" | solver phases " + " ".join(f"{n}:{v:.0f}" for n, v in zip(names[:6], d[i, :6])) +
              " | per env (newton its, ls evals, ncon, nefc): " + " ".join(f"({d[j, 7] / 16e-3:.0f},{d[j, 6] / 16e-3:.0f},{cur[j, 0]},{cur[j, 1]})" for j in same))
print(env.status())
