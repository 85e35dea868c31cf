#!/usr/bin/env python
"""Measure tau(model, lanes per workgroup): the slowest wave of every model group, ms per step, with all groups forced to the
same lanes-per-workgroup value (MW_LANES_PER_BLOCK) on MT50 @ 4096 envs with random actions, from the per-workgroup wall-clock
ticks of the resident step launches (mw_wave_profile).  Writes metaworld_amd/data/lpb_costs.json, the cost table of
metaworld_amd/lpb_policy.py.  Run on the GPU box: python tools/calibrate_lpb.py [precision] [out.json]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "metaworld_amd", "data", "lpb_costs.json")
n, warm, K = 4096, 120, 120
acts = np.random.default_rng(0).uniform(-1, 1, (64, n, 4)).astype(np.float32)
costs, launch = {}, {}
for l in (1, 2, 4, 8, 16, 32):
    os.environ["MW_LANES_PER_BLOCK"] = str(l)
    t0 = time.perf_counter()
    env = MetaWorldGpuVectorEnv("MT50", num_envs=n, seed=42, use_one_hot=True, precision=prec, lanes_per_block=None)
    env.reset()
    env.ctx.upload_actions(acts)
    env.ctx.step_resident(warm)
    nb = env.ctx.wave_profile_start()
    ms = env.ctx.step_resident(K)
    ticks, model = env.ctx.wave_profile_read(nb)
    names = {i: m for m, i in env.model_index.items()}
    for i in np.unique(model):
        costs.setdefault(names[int(i)], {})[str(l)] = round(float(ticks[model == i].max()) / K / 1e5, 4)          # 100 MHz ticks -> ms
    launch[str(l)] = round(ms / K, 3)
    print(f"l={l:2d}: {nb} workgroups, {ms / K:.2f} ms/launch, slowest wave {ticks.max() / K / 1e5:.2f} ms, mean wave {ticks.mean() / K / 1e5:.2f} ms "
          f"({time.perf_counter() - t0:.1f} s)", flush=True)
    env.close()
del os.environ["MW_LANES_PER_BLOCK"]
tab = json.load(open(out)) if os.path.exists(out) else {}
tab[prec] = costs
tab.setdefault("_meta", {})[prec] = dict(workload=f"MT50 @ {n} envs, random actions, steps {warm}..{warm + K} after reset", launch_ms=launch,
                                         unit="ms per step of the model's slowest wave")
json.dump(tab, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out)
