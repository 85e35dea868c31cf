#!/usr/bin/env python
"""Compare the matrix-core Hessian / J'force pass with the sub-lane assembly on the same states (probe build of the library:
-DMW_SOLVER_PROBE stops the solver after its first Hessian evaluation).  MW_LANES_PER_BLOCK=4 -> fused pass, 64 -> sub-lane path."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] in ("gen", "probe"):
    from metaworld_amd import native
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    mode, task, prec, out = sys.argv[1:5]
    n = 6
    if mode == "gen" and os.environ.get("MW_SCENARIO"):          # the raw-physics scenario of test_gpu_physics_matches_oracle, k substeps in
        env = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision=prec)
        env.reset()
        env.ctx.debug("reset_data")
        for e in range(n):
            env.ctx.write(e, "mocap", [0, 0.6, 0.2]); env.ctx.write(e, "ctrl", [-1, 1])
        env.ctx.debug("substeps", int(os.environ["MW_SCENARIO"]))
        np.save(out, np.array([env.ctx.read(e, "state") for e in range(n)]))
        env.close()
        sys.exit(0)
    if mode == "gen" and os.environ.get("MW_ROLLOUT"):          # tools/experiments/ab_rollout.py's MW_SAME action stream, k steps in
        lib = native.load("mw_", os.path.join(ROOT, "metaworld_amd", "libmwgpu_nofused.so"))
        env = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision=prec, lib=lib)
        env.reset()
        acts = np.random.default_rng(0).uniform(-1, 1, (64, 82, 4)).astype(np.float32)[:, :1]
        for t in range(int(os.environ["MW_ROLLOUT"])):
            env.step(np.repeat(acts[t % 64], n, axis=0))
        # one more action applied to the state by hand (mocap / ctrl), so that the probe's forward sees a mid-step state
        np.save(out, np.array([env.ctx.read(e, "state") for e in range(n)]))
        env.close()
        sys.exit(0)
    if mode == "gen":          # sub-lane-free configuration (lanes per block 64) of the full library: reach an interesting state
        env = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision=prec)
        env.reset()
        rng = np.random.default_rng(1)
        for t in range(int(os.environ.get("MW_STEPS", "40"))):
            env.step(rng.uniform(-1, 1, (n, 4)).astype(np.float32))
        np.save(out, np.array([env.ctx.read(e, "state") for e in range(n)]))
        env.close()
        sys.exit(0)
    if os.environ.get("MW_HOST_PROBE"):
        probe = native.load("mwh_", os.path.join(ROOT, "tests", "_build", "libmw_hostsim_probe" + os.environ.get("MW_PROBE_K", "1") + ".so"))
    else:
        probe = native.load("mw_", os.path.join(ROOT, "metaworld_amd", "libmwgpu_probe" + os.environ.get("MW_PROBE_K", "1") + ".so"))
    state = np.load(sys.argv[5])
    env = MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision=prec, lib=probe)
    env.ctx.reset(np.zeros(n, dtype=np.int32))
    for e in range(n):
        env.ctx.write(e, "state", state[e])
    env.ctx.debug("forward")
    nv = len(env.ctx.read(0, "qvel"))
    np.savez(out, qH=np.array([env.ctx.read(e, "qH") for e in range(n)]), qf=np.array([env.ctx.read(e, "qfrc_constraint") for e in range(n)]),
             ic=np.array([env.ctx.read_int(e, "icount", 24) for e in range(n)]), nv=nv,
             J=np.array([env.ctx.read(e, "efcJ")[:200 * nv] for e in range(n)]), X=np.array([env.ctx.read(e, "efcX")[:200 * 15] for e in range(n)]),
             M=np.array([env.ctx.read(e, "qM") for e in range(n)]))
    env.close()
    sys.exit(0)
for task in sys.argv[1:] or ["reach-v3", "box-close-v3"]:
    for prec in ("fp64",):
        outs = {}
        st = f"/tmp/probe_{task}_{prec}_state.npy"
        subprocess.check_call([sys.executable, __file__, "gen", task, prec, st], env=dict(os.environ, MW_LANES_PER_BLOCK="64"))
        for lpb in ("4", "64"):
            out = f"/tmp/probe_{task}_{prec}_{lpb}.npz"
            extra = {"MW_HOST_PROBE": "1", "MW_NSUB": "1"} if lpb == "64" else {}          # reference: the host build's sub-lane assembly
            subprocess.check_call([sys.executable, __file__, "probe", task, prec, out, st], env=dict(os.environ, MW_LANES_PER_BLOCK=lpb, **extra))
            outs[lpb] = np.load(out)
        a, b = outs["4"], outs["64"]
        nv = int(a["nv"])
        Ha, Hb = a["qH"].reshape(-1, nv, nv), b["qH"].reshape(-1, nv, nv)
        tril = np.tril(np.ones((nv, nv), bool))
        relH = np.abs(Ha - Hb)[:, tril].max() / np.abs(Hb).max()
        print(f"{task} {prec}: nefc {a['ic'][:,1]} vs {b['ic'][:,1]}  |H_fused - H_sub|/max|H| = {relH:.3e}   |qf_fused - qf_sub| = {np.abs(a['qf']-b['qf']).max():.3e} (max |qf| {np.abs(b['qf']).max():.3e})")
        if relH > 1e-3:
            e = int(np.abs(Ha - Hb).reshape(len(Ha), -1).max(1).argmax())
            np.set_printoptions(linewidth=250, precision=3, suppress=False)
            D = np.tril(Ha[e] - Hb[e]); print("largest |H_fused - H_host| entries (i, j, fused, host):", [(int(i), int(j), float(Ha[e][i, j]), float(Hb[e][i, j])) for i, j in zip(*np.unravel_index(np.argsort(-np.abs(D), axis=None)[:6], D.shape))])
            for tag, o in (("fused", a), ("sub", b)):
                nefc = int(o["ic"][e, 1]); J = o["J"][e][:nefc * nv].reshape(nefc, nv); X = o["X"][e][:nefc * 15].reshape(nefc, 15)
                print(tag, "env", e, "M00", o["M"][e][0], "H00", o["qH"][e][0], "rows with J[r,0] != 0: (row, info, state(col-store), D, J[r,0], D*J0^2, nnz)")
                for r in range(nefc):
                    if J[r, 0] != 0: print("   ", r, X[r, 9], X[r, 10], f"{X[r,3]:.4g} {J[r,0]:.4g} {X[r,3]*J[r,0]**2:.4g}", int((J[r] != 0).sum()))
            np.set_printoptions(linewidth=250, precision=3, suppress=False)
            print("env", e, "H_fused - H_sub (lower triangle):\n", np.tril(Ha[e] - Hb[e]), "\nH_sub:\n", np.tril(Hb[e]))
