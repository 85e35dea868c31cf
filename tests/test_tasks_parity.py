"""All 50 v3 tasks: the device task layer + physics (host build of the lane programs, fp64) replayed against
golden traces produced by the REFERENCE's own Python (metaworld/envs/*.py, sawyer_xyz_env.py) running on the oracle
engine (tools/gen_golden.py).  One step from a synchronised state: obs / reward within 1e-5, success flags bit-exact."""
import numpy as np
import pytest

from metaworld_amd import tasks as T
from tests.helpers import golden, make_env, replay_trace

# Documented exceptions.  At the states where these three tasks exceed 1e-5 the reference itself amplifies a 1e-12 perturbation of
# the synchronised state to 2e-5 ... 8e-4 in ONE step (tests/test_ill_conditioning.py proves it on the reference's own Python;
# tools/experiments/waiver_scan.py finds the states): a contact sitting at its activation margin, or a face-on-face contact whose
# single contact point is not a continuous function of the poses (the lock's flat mesh faces, the plug seated in its socket).
# The tolerances are ~3x the deviation measured on the host build (door-unlock 5.9e-5 / 7.8e-4, peg-unplug 8.8e-5 / 2.8e-5); 16 sub-lanes, FMA contraction and the single-precision Hessian factor change the rounding on the GPU.
# (peg-unplug on the GPU, round 3: obs 9.9e-5, reward 3.3e-4 -- the reward limit is 3x that)
TOL = {"door-unlock-v3": (2e-4, 3e-3), "peg-unplug-side-v3": (3e-4, 1e-3)}          # (door-close meets 1e-5 since box faces are decided on their axes)

@pytest.mark.parametrize("task", T.ALL_V3)
def test_task_matches_reference_trace(hostsim, task):
    G = dict(golden(f"trace_{task}_seed42.npz"))
    if task == "basketball-v3":     # only the first episode of a fresh env is history-free
        G = {k: (v[:1] if getattr(v, "ndim", 0) >= 1 and len(v) == len(G["goal_idx"]) and k != "rand_vecs" else v) for k, v in G.items()}
    env = make_env(hostsim, task, n=len(G["goal_idx"]), precision="fp64")
    r = replay_trace(env, G, sync=True)
    env.close()
    tol_obs, tol_rew = TOL.get(task, (1e-5, 1e-5))
    assert r["reset"] < 1e-7, r
    assert r["obs"] < tol_obs and r["reward"] < tol_rew, r
    assert r["info"] < max(2e-5, tol_rew), r          # near_object, grasp_success, ... (float32 at the ABI)
    assert r["success_mismatch"] == 0, r


def test_every_mt50_task_has_device_code():
    assert T.supported_tasks() == T.ALL_V3 and len(T.ALL_V3) == 50


@pytest.mark.parametrize("task", T.ALL_V3)
def test_task_fp32_close_to_reference_trace(hostsim, task):
    """The throughput precision (fp32 state and arithmetic): success flags exact; observations / rewards one step from a
    synchronised state within the single-precision floor of the contact geometry (MPR depth ~2e-6, poses 6e-8): 28/50
    tasks <= 1e-5, 46/50 <= 1e-3, all <= 1e-2."""
    G = dict(golden(f"trace_{task}_seed42.npz"))
    if task == "basketball-v3":
        G = {k: (v[:1] if getattr(v, "ndim", 0) >= 1 and len(v) == len(G["goal_idx"]) and k != "rand_vecs" else v) for k, v in G.items()}
    env = make_env(hostsim, task, n=len(G["goal_idx"]), precision="fp32")
    r = replay_trace(env, G, sync=True, steps=50)
    env.close()
    # box-close: the reference's reward adds a bonus once the lid is above z = 0.02 -- exactly its resting height; in the trace
    # the lid sits at 0.02000011 (fp64) / 0.01999891 (fp32) at one step, so the single-precision reward is on the other branch
    tol_rew = 2.0 if task == "box-close-v3" else 5e-2
    # (success flags are exact in single precision too: the one-mismatch allowance door-unlock had in round 3 is gone)
    assert r["reset"] < 1e-2 and r["obs"] < 1e-2 and r["reward"] < tol_rew and r["success_mismatch"] == 0, r
