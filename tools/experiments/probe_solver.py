#!/usr/bin/env python
"""The solver's converged point from the same states with the matrix-core path (MW_LANES_PER_BLOCK=4) and the sub-lane-free path
(64) of the SAME library: qacc, qfrc_constraint, iteration counts."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] in ("gen", "probe"):
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    mode, task, prec, out = sys.argv[1:5]
    n = 6
    env = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision=prec)
    if mode == "gen":
        env.reset()
        rng = np.random.default_rng(1)
        for t in range(int(os.environ.get("MW_STEPS", "40"))):
            env.step(rng.uniform(-1, 1, (n, 4)).astype(np.float32))
        np.save(out, np.array([env.ctx.read(e, "state") for e in range(n)]))
    else:
        state = np.load(sys.argv[5])
        env.ctx.reset(np.zeros(n, dtype=np.int32))
        for e in range(n):
            env.ctx.write(e, "state", state[e])
        env.ctx.debug("forward")
        np.savez(out, qacc=np.array([env.ctx.read(e, "qacc") for e in range(n)]), qf=np.array([env.ctx.read(e, "qfrc_constraint") for e in range(n)]),
                 ic=np.array([env.ctx.read_int(e, "icount", 24) for e in range(n)]), smooth=np.array([env.ctx.read(e, "qacc_smooth") for e in range(n)]))
    env.close()
    sys.exit(0)
for task in sys.argv[1:] or ["reach-v3", "box-close-v3"]:
    for prec in ("fp64", "fp32"):
        st = f"/tmp/ps_{task}_{prec}_state.npy"
        subprocess.check_call([sys.executable, __file__, "gen", task, prec, st], env=dict(os.environ, MW_LANES_PER_BLOCK="64"))
        o = {}
        for lpb in ("4", "64"):
            out = f"/tmp/ps_{task}_{prec}_{lpb}.npz"
            subprocess.check_call([sys.executable, __file__, "probe", task, prec, out, st], env=dict(os.environ, MW_LANES_PER_BLOCK=lpb))
            o[lpb] = np.load(out)
        a, b = o["4"], o["64"]
        print(f"{task} {prec}: niter fused {a['ic'][:,2]} sub {b['ic'][:,2]} nefc {a['ic'][:,1]}/{b['ic'][:,1]} | max|qacc_f - qacc_s| {np.abs(a['qacc']-b['qacc']).max():.3e} "
              f"(|qacc| {np.abs(b['qacc']).max():.3e}, |qacc_smooth diff| {np.abs(a['smooth']-b['smooth']).max():.1e}) max|qf_f - qf_s| {np.abs(a['qf']-b['qf']).max():.3e} (|qf| {np.abs(b['qf']).max():.3e})")
        if np.abs(a['qacc']-b['qacc']).max() > 1e-4 * np.abs(b['qacc']).max():
            e = int(np.abs(a['qacc']-b['qacc']).max(1).argmax())
            np.set_printoptions(linewidth=220, precision=5)
            print("  env", e, "\n  qacc fused", a['qacc'][e], "\n  qacc sub  ", b['qacc'][e], "\n  qf fused", a['qf'][e], "\n  qf sub  ", b['qf'][e])
