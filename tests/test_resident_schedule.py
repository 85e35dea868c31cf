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


def test_default_schedule_rows_follow_the_reset_bound(hostsim):
    """with short episodes the DEFAULT number of schedule rows must cover every auto-reset an env can make (round 5:
    K = nsteps // 50 + 2 = 2 against 3-4 resets here -> RuntimeError after the kernel had run)"""
    a, b = _pair(hostsim)
    a.reset(); b.reset()
    acts = np.random.default_rng(2).uniform(-1, 1, (8, 20, 4)).astype(np.float32)
    a.ctx.upload_actions(acts)
    a.step_resident(23)                                         # max_episode_steps = 7: up to 4 resets per env
    for t in range(23):
        b.step(acts[t % 8])
    _same_state(a, b)
    a.close(); b.close()


def test_schedule_overflow_leaves_the_bookkeeping_consistent(hostsim):
    """an explicit table that is too short: the error is raised AFTER the host bookkeeping has followed the device for the rows that
    were consumed (goal of every env = the last row the kernel used), and the env object stays usable"""
    a, _b = _pair(hostsim)
    _b.close()
    a.reset()
    acts = np.random.default_rng(3).uniform(-1, 1, (8, 20, 4)).astype(np.float32)
    a.ctx.upload_actions(acts)
    count0 = a._reset_count.copy()
    with pytest.raises(RuntimeError, match="schedule rows"):
        a.step_resident(23, schedule_rows=2)
    assert (a._reset_count - count0 == 2).all()                # every env made >= 3 resets, two rows were real draws
    with pytest.raises(ValueError):
        a.step_resident(3, schedule_rows=0)
    o, r, *_ = a.step(acts[0])                                  # still steps
    assert np.isfinite(o).all() and np.isfinite(r).all()
    a.close()


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


def test_fused_launches_equal_the_per_step_loop_bit_for_bit(hostsim):
    """mw_step_resident_fused: several consecutive steps of every environment per launch -- same state, same outputs, same
    task-selection streams as one launch per step (auto-resets with per-reset task draws included)"""
    a, b = _pair(hostsim)
    a.reset(); b.reset()
    acts = np.random.default_rng(2).uniform(-1, 1, (8, 20, 4)).astype(np.float32)
    for env in (a, b):
        env.ctx.upload_actions(acts)
        env.ctx.set_episode_phase((np.arange(20) * 3 % 7).astype(np.int32))
    a.step_resident(23, schedule_rows=6, steps_per_launch=5)          # launches of 5, 5, 5, 5, 3 steps
    b.step_resident(23, schedule_rows=6)
    _same_state(a, b)
    x = a.step(acts[0]); y = b.step(acts[0])          # (the resident loops leave their outputs on the device: compare through one ordinary step)
    assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    _same_state(a, b)
    assert a.ctx.status()["flags"] == 0
    with pytest.raises(ValueError):
        a.step_resident(2, gather=True, steps_per_launch=2)
    a.close(); b.close()


@pytest.mark.gpu
def test_gpu_fused_launches_equal_the_per_step_loop_bit_for_bit(gpulib):
    """the same on the GPU, MT50 @ 400 (every scene, partial last workgroups -> ghost lanes), fp64: 120 steps as 3 launches of 40 against 120
    launches; also the schedule-driven resampling against the per-step host loop"""
    envs = [MetaWorldGpuVectorEnv("MT50", num_envs=400, seed=3, use_one_hot=True, precision="fp64", lib=gpulib, max_episode_steps=45) for _ in range(3)]
    acts = np.random.default_rng(4).uniform(-1, 1, (16, 400, 4)).astype(np.float32)
    for env in envs:
        env.reset()
        env.ctx.upload_actions(acts)
        env.ctx.set_episode_phase((np.arange(400) * 7 % 45).astype(np.int32))
    a, b, c = envs
    a.step_resident(120, schedule_rows=8, steps_per_launch=40)
    b.step_resident(120, schedule_rows=8)
    for t in range(120):
        c.step(acts[t % 16])
    for other in (b, c):
        assert np.array_equal(a._cur_goal, other._cur_goal) and np.array_equal(a._reset_count, other._reset_count)
        for e in range(0, 400, 7):
            assert np.array_equal(a.ctx.read(e, "qpos"), other.ctx.read(e, "qpos")), e
    x = a.step(acts[0]); y = b.step(acts[0])
    assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    for env in envs:
        assert env.ctx.status()["flags"] == 0
        env.close()
