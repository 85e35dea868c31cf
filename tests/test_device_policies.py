"""Scripted policies on the device: the generated lane code (metaworld_amd/csrc/mw_policies_gen.hpp) must
reproduce the batched numpy policies -- themselves pinned bit-exactly against the reference's -- action for action, and the
all-device closed loop (mw_policy_rollout) must book the same episodes as the same loop driven from the host."""
import subprocess
import sys

import numpy as np
import pytest

from metaworld_amd import policies as P, tasks as T
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
from tests.helpers import golden


def test_generated_header_is_current():
    assert subprocess.call([sys.executable, "tools/gen_device_policies.py", "--check"]) == 0, \
        "metaworld_amd/csrc/mw_policies_gen.hpp is stale: run tools/gen_device_policies.py"


def _obs_pool(task, rng):
    """observations along the task's golden scripted episode, plus perturbed copies that reach the other branches"""
    G = golden(f"policy_{task}_seed42.npz")
    goal = golden(f"trace_{task}_seed42.npz")["reset_obs"][0][36:39]
    n = len(G["obs18"])
    obs = np.zeros((n, 39)); obs[:, :18] = G["obs18"]; obs[1:, 18:36] = G["obs18"][:-1]; obs[0, 18:36] = G["obs18"][0]; obs[:, 36:39] = goal
    return np.concatenate([obs, obs + rng.normal(0, 0.01, obs.shape), obs + rng.normal(0, 0.05, obs.shape), obs + rng.normal(0, 0.3, obs.shape)])


def _check_actions(lib, rounds):
    rng = np.random.default_rng(0)
    per = 4
    env = MetaWorldGpuVectorEnv("MT50", num_envs=50 * per, seed=42, precision="fp32", lib=lib, partially_observable=False)
    names = env.env_task_names
    pid = np.array([T.ALL_V3.index(n) for n in names], dtype=np.int32)
    pools = {t: _obs_pool(t, rng) for t in T.ALL_V3}
    bad = {}
    for r in range(rounds):
        obs = np.stack([pools[n][rng.integers(len(pools[n]))] for n in names])
        got = env.ctx.policy_actions(pid, obs)
        want = P.batched_actions(names, obs)
        for e in np.flatnonzero((got != want).any(axis=1)):
            bad[names[e]] = (got[e], want[e])
    env.close()
    assert not bad, bad


def test_device_policies_bit_exact_on_host_harness(hostsim):
    _check_actions(hostsim, rounds=150)


@pytest.mark.gpu
def test_device_policies_bit_exact_on_gpu(gpulib):
    _check_actions(gpulib, rounds=150)


def _check_rollout(lib, steps):
    kw = dict(num_envs=20, seed=42, precision="fp32", lib=lib, partially_observable=False, terminate_on_success=True, max_episode_steps=120)
    env = MetaWorldGpuVectorEnv("MT10", **kw)
    names = env.env_task_names
    pid = np.array([T.ALL_V3.index(n) for n in names], dtype=np.int32)
    K = 4
    sched = (np.arange(K)[:, None] * 7 + np.arange(20)[None, :]) % 50
    ep, su, ms = env.ctx.policy_rollout(pid, sched, steps)
    # the same loop with the host in it: numpy policies, mw_step, goals from the same schedule
    obs = env.ctx.reset(sched[0]).copy()
    ep2, su2, ever = np.zeros(20, dtype=int), np.zeros(20, dtype=int), np.zeros(20, dtype=bool)
    for t in range(steps):
        nxt = sched[np.minimum(ep2 + 1, K - 1), np.arange(20)]
        obs, r, te, tr, s, info = env.ctx.step(P.batched_actions(names, obs), nxt)
        obs = obs.copy()
        ever |= s.astype(bool)
        done = (te | tr).astype(bool)
        su2 += done & ever; ep2 += done; ever &= ~done
    env.close()
    assert np.array_equal(ep, ep2) and np.array_equal(su, su2), (ep, ep2, su, su2)
    assert ep.sum() >= 20 and su.sum() >= 0.8 * ep.sum()          # the reference's own gate: 80 % scripted success


def test_device_rollout_equals_host_driven_loop(hostsim):
    _check_rollout(hostsim, 300)


@pytest.mark.gpu
def test_device_rollout_equals_host_driven_loop_gpu(gpulib):
    _check_rollout(gpulib, 300)
