"""200-step CLEAN scripted-policy episodes of the reference (its policies + env classes on the oracle engine, goal 7 of MT1(task, 42);
tools/gen_golden.py --mode policy --steps 200 --episodes 1 --first-goal 7 --tag policy200), replayed ONE STEP FROM A SYNCHRONISED STATE
at every step: observation / reward within the tolerances of the 60-step traces, success flags exact.  Unlike the mixed 60-step
traces these stay in the manipulation regime long after the first success -- a button held against its stop, a peg seated in its
hole, a lock at its limit (the v1 goldens found a real parity bug in exactly such states).

Sustained stiff contacts are where the REFERENCE computation itself is ill-conditioned (tests/test_ill_conditioning.py): at 8 of the 50
tasks some steps exceed 1e-5.  The test does not widen a tolerance by hand: a step over the limit is re-run on the oracle engine from
the same synchronised state with qpos perturbed by 1e-12 (ten random directions), and the limit of THAT step follows the reference's
own response (10 x it, at most 1e-3) -- the rule of tests/test_gpu_fullsize.py::test_bench_states_match_the_oracle.  Every relaxed step
is reported (stdout with -rA, gpurun_out/policy200_relaxed_<backend>.txt; committed under profiles/), and a step whose deviation the
reference's conditioning does not explain fails."""
import os

import numpy as np
import pytest

from metaworld_amd import tasks as T
from tests.helpers import golden, make_env
from tests.test_tasks_parity import TOL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _replay(lib, task, backend):
    from tests.test_gpu_fullsize import _oracle_response
    G = dict(golden(f"trace_policy200_{task}_seed42.npz"))
    env = make_env(lib, task, n=1, precision="fp64")
    ctx = env.ctx
    obs = ctx.reset(G["goal_idx"]).copy()
    assert np.abs(obs - G["reset_obs"]).max() < 1e-7
    tol_obs, tol_rew = TOL.get(task, (1e-5, 1e-5))
    relaxed, bad, mism = [], [], 0
    nsteps = G["actions"].shape[1]
    for t in range(nsteps):
        if t > 0:
            ctx.write(0, "qpos", G["qpos"][0, t - 1]); ctx.write(0, "qvel", G["qvel"][0, t - 1])
            ctx.write(0, "mocap", G["mocap"][0, t - 1]); ctx.write(0, "warm", G["warm"][0, t - 1])
            tk = ctx.read(0, "task"); tk[15:33] = G["obs"][0, t - 1][:18]; ctx.write(0, "task", tk)
        state = {c: ctx.read(0, c) for c in ("qpos", "qvel", "warm", "reloc")}
        o, r, te, tr, su, info = ctx.step(G["actions"][:, t])
        eo, er = np.abs(o - G["obs"][:, t]).max(), abs(r[0] - G["reward"][0, t])
        mism += int(su[0] != G["success"][0, t])
        if eo < tol_obs and er < tol_rew:
            continue
        state["mocap"], state["ctrl"] = ctx.read(0, "mocap"), ctx.read(0, "ctrl")          # (what the five substeps of this step used)
        # (such states are two-branched: most 1e-12 perturbations change nothing, some flip a contact and move the answer by 1e-4 --
        #  ten probes, so that the branch the device took is seen)
        rq, _ = _oracle_response(ctx, 0, task, state, trials=10)
        lim_o = min(max(tol_obs, 10 * rq), 1e-3)
        lim_r = min(max(tol_rew, 10 * rq * 100), 2e-2)          # (rewards have slopes of up to ~1e2 per metre)
        (relaxed if (eo < lim_o and er < lim_r) else bad).append((t, float(eo), float(er), float(rq)))
    st = env.status()
    env.close()
    lines = [f"{task:28s} {len(relaxed):3d} of {nsteps} steps relaxed, {len(bad)} unexplained"] + \
            [f"    step {t:3d}  obs err {eo:.2e}  reward err {er:.2e}  oracle response to 1e-12: {rq:.2e}" for t, eo, er, rq in relaxed + bad]
    print("\n".join(lines))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"policy200_relaxed_{backend}.txt"), "a") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass
    assert not bad, (task, bad)
    assert mism == 0, (task, mism)
    assert len(relaxed) <= nsteps // 4, (task, len(relaxed))
    assert st["flags"] == 0, st


@pytest.mark.parametrize("task", T.ALL_V3)
def test_policy200_one_step_parity_on_the_host_build(hostsim, task):
    _replay(hostsim, task, "hostbuild")


@pytest.mark.gpu
@pytest.mark.parametrize("task", T.ALL_V3)
def test_policy200_one_step_parity_on_the_gpu(gpulib, task):
    _replay(gpulib, task, "gpu")
