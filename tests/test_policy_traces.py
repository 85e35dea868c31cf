"""Open-loop replay of closed-loop scripted-policy episodes (the reference's policies + env classes on the oracle engine,
tools/gen_policy_traces.py): the manipulation regimes -- grasps, lifts, insertions, pushes against walls -- that random
actions never reach.  The device lane programs (host build, fp64) must follow the oracle trajectory for the WHOLE episode
and report success at the same step."""
import numpy as np
import pytest

from metaworld_amd import tasks as T
from tests.helpers import golden, make_env

# open-loop drift budget (fp64) over the whole episode: two independent implementations of a chaotic contact system.
# 34/50 tasks stay below 1e-9, 44 below 1e-5; a puck sliding on a plate, the stick and the gripper-palm mesh contact drift more.
TOL = {"plate-slide-side-v3": (2e-2, 0.5), "stick-pull-v3": (2e-2, 2e-2), "door-unlock-v3": (5e-3, 5e-2),
       "peg-unplug-side-v3": (1e-2, 0.6), "hand-insert-v3": (2e-3, 1e-2)}      # (budgets cover host nsub = 8 and the GPU's 16 sub-lanes + FMA contraction)
TOL_DEFAULT = (1e-4, 1e-3)


def replay_policy(env, G):
    ctx = env.ctx
    ctx.reset(G["goal_idx"])
    eo = er = 0.0
    succ = []
    for t in range(len(G["reward"])):
        o, r, te, tr, su, info = ctx.step(G["actions"][t][None])
        eo = max(eo, np.abs(o[0, :18] - G["obs18"][t]).max()); er = max(er, abs(r[0] - G["reward"][t]))
        succ.append(float(su[0]))
    return eo, er, np.array(succ)


def first_success(s):
    return int(np.argmax(s)) if s.any() else -1


def check_fp64(lib, task):
    G = golden(f"policy_{task}_seed42.npz")
    env = make_env(lib, task, n=1, precision="fp64")
    eo, er, succ = replay_policy(env, G)
    env.close()
    tol_obs, tol_rew = TOL.get(task, TOL_DEFAULT)
    assert eo < tol_obs and er < tol_rew, (eo, er)
    assert (succ == G["success"]).all()


def check_fp32(lib, task):
    """single precision drifts further over ~100 open-loop steps, but reaches success at the same step on all 50 tasks"""
    G = golden(f"policy_{task}_seed42.npz")
    env = make_env(lib, task, n=1, precision="fp32")
    eo, er, succ = replay_policy(env, G)
    env.close()
    assert first_success(succ) == first_success(G["success"])
    assert np.isfinite(eo) and np.isfinite(er)


@pytest.mark.parametrize("task", T.ALL_V3)
def test_policy_episode_follows_oracle(hostsim, task):
    check_fp64(hostsim, task)


@pytest.mark.parametrize("task", T.ALL_V3)
def test_policy_episode_fp32_same_success_step(hostsim, task):
    check_fp32(hostsim, task)
