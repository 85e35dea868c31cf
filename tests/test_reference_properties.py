"""Mirrors of the reference's own property tests at the SawyerXYZEnv level (tests/metaworld/envs/mujoco/sawyer_xyz/):
test_sawyer_xyz_env.py (`test_reset_returns_same_obj_and_goal`), test_obs_space_hand.py (`test_reaching_limit`) and
test_seeded_rand_vec.py (`test_observations_match`), on the VectorEnv boundary."""
import numpy as np
import pytest

from metaworld_amd.vector_env import MetaWorldGpuVectorEnv

HAND_LOW, HAND_HIGH = np.array([-0.525, 0.348, -0.0525]), np.array([0.525, 1.025, 0.7])          # sawyer_xyz_env.py:146-150


@pytest.fixture(scope="module")
def mt50(hostsim):
    env = MetaWorldGpuVectorEnv("MT50", seed=42, precision="fp64", lib=hostsim)          # one build of the 2500 reset snapshots
    yield env
    env.close()


def test_reset_returns_same_obj_and_goal(mt50):
    """test_sawyer_xyz_env.py:8-47: resetting an env twice on the same task gives the same object pose and goal"""
    env = mt50
    env.call("toggle_sample_tasks_on_reset", False)
    with pytest.raises(AssertionError):
        env.reset()                               # no task yet (sawyer_xyz_env.py:699-701)
    env.call("sample_tasks")
    (o1, _), (o2, _) = env.reset(), env.reset()
    env.step(np.ones((50, 4), dtype=np.float32))
    o3, _ = env.reset()
    assert np.array_equal(o1[:, 3:9], o2[:, 3:9])
    # basketball is the one env whose goal is NOT stable in the reference: `_target_pos` is a live view of the goal site's
    # world position and reset_model writes it back as the site's LOCAL position (envs/sawyer_basketball_v3.py:118-121), so
    # the goal observation grows by one basket offset per reset.  Reproduced faithfully (and pinned by the golden traces).
    bb = np.array([n == "basketball-v3" for n in env.env_task_names])
    assert np.array_equal(o1[~bb, -3:], o2[~bb, -3:]) and np.array_equal(o1[~bb], o3[~bb])          # also after the env has moved
    assert np.allclose(o3[bb, -3:] - o2[bb, -3:], o2[bb, -3:] - o1[bb, -3:], atol=1e-12) and not np.allclose(o1[bb, -3:], o2[bb, -3:])
    env.call("toggle_sample_tasks_on_reset", True)


def _reach_limit(lib, n=100):
    rng = np.random.default_rng(0)
    targets = rng.standard_normal((3, n)); targets = (targets / np.linalg.norm(targets, axis=0)).T * 10.0          # sample_spherical(100, 10.0)
    env = MetaWorldGpuVectorEnv("MT1", "reach-v3", num_envs=n, seed=1, precision="fp32", lib=lib, partially_observable=False)
    o_prev, _ = env.reset()
    live = np.ones(n, dtype=bool)
    final = o_prev[:, :3].copy()
    for _ in range(499):
        a = np.clip(25.0 * (targets - o_prev[:, :3]), -1, 1).astype(np.float32)          # move(hand, to_xyz=target, p=25), grab 0
        o, *_ = env.step(np.concatenate([a, np.zeros((n, 1), dtype=np.float32)], axis=1))
        final[live] = o[live, :3]
        live &= np.linalg.norm(o[:, :3] - o_prev[:, :3], axis=1) >= 0.001          # the reference stops an env once the hand stalls
        o_prev = o
        if not live.any():
            break
    env.close()
    assert (final >= HAND_LOW).all() and (final <= HAND_HIGH).all(), (final.min(axis=0), final.max(axis=0))
    assert np.abs(final).max() > 0.3               # the hand really went to the limits


def test_reaching_limit(hostsim):
    """test_obs_space_hand.py:44-64: driven towards 100 far-away targets the hand stays inside `_HAND_SPACE`"""
    _reach_limit(hostsim)


@pytest.mark.gpu
def test_reaching_limit_gpu(gpulib):
    _reach_limit(gpulib)


def _observations_match(lib, steps, tasks=None):
    """two identically built envs fed the same actions (the reference draws uniform(-1, -1) = -1) stay bit-identical"""
    env = MetaWorldGpuVectorEnv("MT50", num_envs=100, seed=7, precision="fp32", lib=lib, task_names=tasks)
    n = env.num_envs
    o, _ = env.reset()
    pair = lambda x: (x[0::2], x[1::2])           # envs 2k and 2k+1 run the same task with the same goal (one task-selection stream)
    assert n % 2 == 0 and env.env_task_names[0::2] == env.env_task_names[1::2]
    a, b = pair(o); assert np.array_equal(a, b)
    act = -np.ones((n, 4), dtype=np.float32)
    for t in range(steps):
        o, r, te, tr, info = env.step(act)
        a, b = pair(o); assert np.array_equal(a, b), t
        a, b = pair(r); assert np.array_equal(a, b), t
        assert not te.any() and not tr.any()
    env.close()


def test_observations_match(hostsim):
    """test_seeded_rand_vec.py:10-28 on a few tasks (the host harness is slow)"""
    _observations_match(hostsim, 60, tasks=["reach-v3", "box-close-v3", "door-unlock-v3", "stick-pull-v3", "sweep-into-v3"])


@pytest.mark.gpu
def test_observations_match_gpu(gpulib):
    """all 50 tasks, two lanes each: identical inputs in different lanes / waves give bit-identical trajectories"""
    _observations_match(gpulib, 150)


def test_target_positions_unique_and_benchmarks_identical(hostsim, mt50):
    """tests/integration/test_new_api.py: `check_target_poss_unique` (:249-275: the 50 rand_vecs of a task give 50 distinct
    goals, except the four envs that only randomise the object), `test_identical_environments` (:278-330: equal seeds give
    equal tasks, the ML1 test split differs from its train split) and the goal visibility asserted in `test_all_mt50` /
    `test_all_ml45` (:146, :212)."""
    from metaworld_amd import tasks as T
    env = mt50
    goals = np.stack([env.ctx.reset(np.full(50, g, dtype=np.int32))[:, 36:39].copy() for g in range(50)], axis=1)          # [task, goal, 3]
    assert np.any(goals != 0, axis=(1, 2)).all()                                              # MT: goal visible
    fixed_goal = {"hammer-v3", "sweep-into-v3", "bin-picking-v3", "basketball-v3"}
    for k, name in enumerate(T.ALL_V3):
        if name not in fixed_goal:
            assert len(np.unique(goals[k], axis=0)) == 50, name
    for bench, name in (("MT1", "sweep-into-v3"), ("ML1-train", "sweep-into-v3"), ("MT10", "reach-v3"), ("ML10-train", "reach-v3")):
        T._goal_cache.clear()
        a = T.goal_table(bench, name, 10).copy()
        T._goal_cache.clear()
        assert np.array_equal(a, T.goal_table(bench, name, 10))
        assert not np.array_equal(a, T.goal_table(bench, name, 11))
    assert not np.array_equal(T.goal_table("ML1-train", "sweep-into-v3", 10), T.goal_table("ML1-test", "sweep-into-v3", 10))
    ml = MetaWorldGpuVectorEnv("ML10-test", seed=42, precision="fp64", lib=hostsim)
    o, _ = ml.reset()
    assert np.all(o[:, -3:] == 0)                                                             # ML: goal hidden
    ml.close()
