#!/usr/bin/env python
"""Development loop of the v1 reward port: replay the v1 golden traces (reference v1 Python on the oracle) one step from the
synchronised state on the host build of the v1 library and print the reward / info / success deviations per task."""
import os, sys, warnings
warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
from metaworld_amd import native, tasks as T
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
lib = native.load("mwh_", g.build_host_harness(v1=True))
gdir = os.environ["V1_GOLDEN"]          # a directory of full traces: tools/gen_golden.py <tasks> --reward-version v1 --out DIR
for task in (sys.argv[1:] or T.ALL_V3):
    G = dict(np.load(os.path.join(gdir, f"trace_v1_{task}_seed42.npz")))
    E, TT = G["actions"].shape[:2]
    if task == "basketball-v3": E = 1
    env = MetaWorldGpuVectorEnv("MT1", task, num_envs=len(G["goal_idx"]), seed=0, precision="fp64", lib=lib, reward_function_version="v1")
    ctx = env.ctx
    ctx.reset(G["goal_idx"])
    wr = wi = 0.0; ws = 0; first = None
    for t in range(TT):
        if t > 0:
            for e in range(len(G["goal_idx"])):
                ctx.write(e, "qpos", G["qpos"][e, t - 1]); ctx.write(e, "qvel", G["qvel"][e, t - 1])
                ctx.write(e, "mocap", G["mocap"][e, t - 1]); ctx.write(e, "warm", G["warm"][e, t - 1])
                tk = ctx.read(e, "task"); tk[15:33] = G["obs"][e, t - 1][:18]; ctx.write(e, "task", tk)
        o, r, te, tr, su, info = ctx.step(G["actions"][:, t])
        for e in range(E):
            dr = abs(r[e] - G["reward"][e, t]) / max(1.0, abs(G["reward"][e, t]))
            di = (np.abs(info[e] - G["info"][e, t]) / np.maximum(1.0, np.abs(G["info"][e, t]))).max()
            if (dr > 1e-5 or di > 1e-5 or su[e] != G["success"][e, t]) and first is None:
                first = (e, t, r[e], G["reward"][e, t], info[e].tolist(), G["info"][e, t].tolist(), su[e], G["success"][e, t])
            wr = max(wr, dr); wi = max(wi, di); ws += int(su[e] != G["success"][e, t])
    env.close()
    ok = wr < 1e-5 and wi < 1e-5 and ws == 0
    print(f"{task:30s} {'OK ' if ok else 'BAD'} rel reward dev {wr:.2e} info dev {wi:.2e} success mismatches {ws}" + ("" if ok else f"\n      first: env {first[0]} step {first[1]} reward {first[2]:.6f} vs {first[3]:.6f}\n      info {np.round(first[4], 5).tolist()}\n      gold {np.round(first[5], 5).tolist()} success {first[6]} vs {first[7]}"), flush=True)
