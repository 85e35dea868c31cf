"""The resident loop draws a new task at every auto-reset like RandomTaskSelectWrapper.reset (metaworld/wrappers.py:116-119):
`MetaWorldGpuVectorEnv.step_resident` (goal schedule handed to the kernel, consumed counts read back) must leave the batch and the
task-selection streams exactly where the same actions through `step()` -- whose per-step host look-ahead is pinned against the
reference's wrapper stack in tests/test_ml_wrappers.py -- leave them."""
import numpy as np
import pytest

from metaworld_amd.vector_env import MetaWorldGpuVectorEnv


def _pair(lib, **kw):
    return [MetaWorldGpuVectorEnv("MT10", num_envs=20, seed=3, use_one_hot=True, precision="fp64", lib=lib, max_episode_steps=7, **kw)
            for _ in range(2)]


def _same_state(a, b):
    assert np.array_equal(a._cur_goal, b._cur_goal) and np.array_equal(a._next_goal, b._next_goal)
    assert np.array_equal(a._reset_count, b._reset_count)
    for e in range(a.num_envs):
        for col in ("qpos", "qvel"):
            assert np.array_equal(a.ctx.read(e, col), b.ctx.read(e, col)), (e, col)


def test_resident_loop_resamples_tasks_like_the_step_loop(hostsim):
    a, b = _pair(hostsim)
    oa, _ = a.reset(); ob, _ = b.reset()
    assert np.array_equal(oa, ob)
    acts = np.random.default_rng(0).uniform(-1, 1, (8, 20, 4)).astype(np.float32)
    a.ctx.upload_actions(acts)
    phase = (np.arange(20) * 3 % 7).astype(np.int32)          # staggered: every step some env resets
    a.ctx.set_episode_phase(phase); b.ctx.set_episode_phase(phase)
    n = 23                                                     # > 3 auto-resets per env; K rows default = 2 would clamp, so ask for enough
    a.step_resident(n, schedule_rows=6)
    goals_seen = set()
    for t in range(n):
        b.step(acts[t % 8])
        goals_seen.update(b._cur_goal.tolist())
    assert len(goals_seen) >= 4                               # (identically seeded sub-envs share ONE stream: draw k is the same goal index everywhere)
    _same_state(a, b)
    # and the two continue identically through the ordinary boundary
    x = a.step(acts[0]); y = b.step(acts[0])
    assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    _same_state(a, b)
    a.close(); b.close()


def test_resident_loop_without_resampling_keeps_the_goal(hostsim):
    """toggle_sample_tasks_on_reset off (wrappers.py:103-104): the auto-resets inside the loop re-use the current task"""
    a, b = _pair(hostsim)
    for env in (a, b):
        env.reset()
        env.call("toggle_sample_tasks_on_reset", False)
    acts = np.random.default_rng(1).uniform(-1, 1, (4, 20, 4)).astype(np.float32)
    a.ctx.upload_actions(acts)
    a.step_resident(16)
    for t in range(16):
        b.step(acts[t % 4])
    _same_state(a, b)
    a.close(); b.close()


def test_schedule_validation(hostsim):
    env = MetaWorldGpuVectorEnv("MT1", "reach-v3", num_envs=2, seed=0, precision="fp32", lib=hostsim)
    env.reset()
    with pytest.raises(RuntimeError):
        env.ctx.set_goal_schedule(np.full((2, 2), 50, dtype=np.int32))          # outside the goal table
    env.ctx.set_goal_schedule(np.zeros((2, 2), dtype=np.int32))
    assert (env.ctx.goal_schedule_pos() == 0).all()
    env.ctx.set_goal_schedule(None)
    env.close()


def test_vectorised_selection_equals_the_per_env_rule(hostsim):
    """`_random_goals` (one gather per stream) against `_select` (the per-env restatement of wrappers.py:98-100) over several
    resets of staggered subsets"""
    env = MetaWorldGpuVectorEnv("MT10", num_envs=30, seed=11, use_one_hot=True, precision="fp32", lib=hostsim)
    rng = np.random.default_rng(5)
    for _ in range(6):
        mask = rng.random(30) < 0.5
        idx = np.flatnonzero(mask)
        want = [env._select(e, commit=False) for e in idx]
        assert env._random_goals(idx).tolist() == want
        want2 = []
        for e in idx:          # the draw after next, by committing one on a copy of the counter
            env._reset_count[e] += 1
            want2.append(env._select(e, commit=False))
            env._reset_count[e] -= 1
        assert env._random_goals(idx, ahead=1).tolist() == want2
        env._begin_episodes(mask)
        assert env._cur_goal[idx].tolist() == want
    env.close()
