#!/usr/bin/env python
"""For every task: (env, step) of the largest one-step-from-sync deviation of the device code (host build, fp64) from the
golden trace, and how much the REFERENCE's own next observation moves when the synchronised qpos is perturbed by 1e-12 at that
state (the numbers behind tests/test_tasks_parity.py::TOL and tests/test_ill_conditioning.py::CASES)."""
import os, sys, warnings
warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
from metaworld_amd import native, tasks as T
from tests.helpers import golden, make_env
from tests.test_ill_conditioning import _step_reference_from
from oracle import refshim
refshim.install()
lib = native.load("mwh_", g.build_host_harness())
for task in (sys.argv[1:] or T.ALL_V3):
    G = dict(golden(f"trace_{task}_seed42.npz"))
    E, TT = G["actions"].shape[:2]
    if task == "basketball-v3": E = 1
    env = make_env(lib, task, n=len(G["goal_idx"]), precision="fp64")
    ctx = env.ctx
    ctx.reset(G["goal_idx"])
    worst = (0, 0, 0); wr = 0; wre = (0, 0)
    for t in range(TT):
        if t > 0:
            for e in range(len(G["goal_idx"])):
                ctx.write(e, "qpos", G["qpos"][e, t - 1]); ctx.write(e, "qvel", G["qvel"][e, t - 1])
                ctx.write(e, "mocap", G["mocap"][e, t - 1]); ctx.write(e, "warm", G["warm"][e, t - 1])
                tk = ctx.read(e, "task"); tk[15:33] = G["obs"][e, t - 1][:18]; ctx.write(e, "task", tk)
        o, r, te, tr, su, info = ctx.step(G["actions"][:, t])
        for e in range(E):
            d = np.abs(o[e] - G["obs"][e, t]).max()
            if d > worst[0]: worst = (d, e, t)
            if abs(r[e] - G["reward"][e, t]) > wr: wr = abs(r[e] - G["reward"][e, t]); wre = (e, t)
    env.close()
    d, e, t = worst
    if d > 3e-6:
        base = _step_reference_from(task, G, e, t, 0.0)
        moved = max(np.abs(_step_reference_from(task, G, e, t, eps) - base).max() for eps in (1e-12, -1e-12))
        print(f"{task:28s} worst obs dev {d:.2e} at env {e} step {t}; reward dev {wr:.2e}; reference moves {moved:.2e} under a 1e-12 perturbation", flush=True)
    else:
        extra = ""
        if wr > 3e-6:
            import metaworld
            def rew(eps):
                mt1 = metaworld.MT1(task, seed=42); env2 = mt1.train_classes[task](); env2.seed(42)
                env2.set_task(mt1.train_tasks[int(G["goal_idx"][wre[0]])]); env2.reset()
                e2, t2 = wre
                dd = env2.data
                dd.qpos[:] = G["qpos"][e2, t2 - 1] + eps; dd.qvel[:] = G["qvel"][e2, t2 - 1]; dd.mocap_pos[0][:] = G["mocap"][e2, t2 - 1]; dd.qacc_warmstart[:] = G["warm"][e2, t2 - 1]
                env2.curr_path_length = t2; env2._prev_obs = G["obs"][e2, t2 - 1][:18].copy()
                return env2.step(G["actions"][e2, t2])[1]
            extra = f" (worst reward at env {wre[0]} step {wre[1]}: reference reward moves {max(abs(rew(1e-12) - rew(0.0)), abs(rew(-1e-12) - rew(0.0))):.2e} under 1e-12)"
        print(f"{task:28s} worst obs dev {d:.2e} reward dev {wr:.2e}" + extra, flush=True)
