"""the ML construction path (`make_ml_envs`: meta_batch_size split, PseudoRandomTaskSelectWrapper,
RNNBasedMetaRLWrapper, NormalizeRewardsExponential) on the GPU VectorEnv against the reference's own wrapper stack
(unmodified Python from /root/reference running on the oracle engine through oracle/refshim.py).  Task selection and the
recurrent tail are exact; obs / reward follow the engine tolerance."""
import os
import warnings

import numpy as np
import pytest

from metaworld_amd import make as mk

needs_ref = pytest.mark.skipif(not os.path.isdir("/root/reference/metaworld"), reason="reference checkout not present")


def _pair(hostsim, name, **kw):
    warnings.filterwarnings("ignore")
    from oracle import refshim
    refshim.install()
    import metaworld
    ref = metaworld.make_ml_envs(name, **kw)
    mine = mk.make_ml_envs(name, precision="fp64", lib=hostsim, **kw)
    return ref, mine


def _cmp_obs(a, b, tol=2e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= tol, np.abs(a - b).max()


@needs_ref
@pytest.mark.parametrize("select", ["pseudorandom", "random"])
def test_ml1_stack_matches_reference(hostsim, select):
    kw = dict(seed=42, meta_batch_size=5, split="train", task_select=select, recurrent_info_in_obs=True,
              reward_normalization_method="exponential", terminate_on_success=False, max_episode_steps=6)
    ref, mine = _pair(hostsim, "reach-v3", **kw)
    assert mine.num_envs == ref.num_envs == 5
    assert mine.single_observation_space.shape == ref.single_observation_space.shape == (45,)
    assert mine.single_observation_space.dtype == ref.single_observation_space.dtype == np.float32
    for a, b in zip(ref.get_attr("tasks"), mine.get_attr("tasks")):          # tasks[i::k] split of the class's 50 goals
        import pickle
        assert np.array_equal(np.stack([pickle.loads(t.data)["rand_vec"] for t in a]), b)
    rng = np.random.default_rng(0)
    for rnd in range(13):          # 10 goals per sub-env: wraps and reshuffles once
        ro, mo = ref.call("sample_tasks"), mine.call("sample_tasks")
        assert np.array_equal(np.stack(ref.get_attr("_last_rand_vec")), np.stack(mine.get_attr("_last_rand_vec")))
        if select == "pseudorandom":
            assert ref.get_attr("current_task_idx") == mine.get_attr("current_task_idx")
        _cmp_obs(np.stack([o for o, _ in ro]), np.stack([o for o, _ in mo]))
        if rnd % 4:
            continue
        for t in range(8):          # crosses the 6-step horizon: SAME_STEP auto-reset
            a = rng.uniform(-1, 1, (5, 4)).astype(np.float32)
            o1, r1, te1, tr1, i1 = ref.step(a)
            o2, r2, te2, tr2, i2 = mine.step(a)
            assert o2.dtype == o1.dtype == np.float32
            assert np.array_equal(te1, te2) and np.array_equal(tr1, tr2)
            _cmp_obs(o1, o2)
            assert np.array_equal(o1[:, 39:43], o2[:, 39:43]) and np.array_equal(o1[:, 44], o2[:, 44])     # action, done: exact
            assert np.abs(r1 - r2).max() <= 1e-5 * max(1.0, np.abs(r1).max())
            # a pseudorandom env keeps its task over an auto-reset, a random one redraws
            assert np.array_equal(np.stack(ref.get_attr("_last_rand_vec")), np.stack(mine.get_attr("_last_rand_vec")))
            if (te1 | tr1).any():
                for e in np.flatnonzero(te1 | tr1):
                    _cmp_obs(i1["final_obs"][e], i2["final_obs"][e])
                    assert i2["final_obs"][e][44] == 1.0
                f1, f2 = i1["final_info"]["episode"], i2["final_info"]["episode"]
                assert np.array_equal(f1["l"], f2["l"])
                assert np.abs(f1["r"] - f2["r"]).max() <= 1e-5 * max(1.0, np.abs(f1["r"]).max())
    ref.close(); mine.close()


@needs_ref
def test_ml10_train_meta_batch_split_and_toggle(hostsim):
    """two sub-envs per class (tasks[0::2], tasks[1::2]); sampling toggled on mid-run takes effect at the next auto-reset"""
    kw = dict(seed=7, meta_batch_size=20, split="train", task_select="pseudorandom", terminate_on_success=False, max_episode_steps=3)
    ref, mine = _pair(hostsim, "ML10", **kw)
    assert mine.num_envs == ref.num_envs == 20
    assert [type(e.unwrapped).__name__ for e in ref.envs] == list(mine.get_attr("task_name"))
    with pytest.raises(AssertionError):          # no task set before sample_tasks (sawyer_xyz_env.py:699-701)
        mine.reset()
    ref.call("sample_tasks"); mine.call("sample_tasks")
    rng = np.random.default_rng(1)
    for t in range(10):
        if t == 4:
            ref.call("toggle_sample_tasks_on_reset", True); mine.call("toggle_sample_tasks_on_reset", True)
        if t == 8:
            ref.call("toggle_sample_tasks_on_reset", False); mine.call("toggle_sample_tasks_on_reset", False)
        a = rng.uniform(-1, 1, (20, 4)).astype(np.float32)
        o1, r1, te1, tr1, i1 = ref.step(a)
        o2, r2, te2, tr2, i2 = mine.step(a)
        assert np.array_equal(tr1, tr2)
        for e, (v1, v2) in enumerate(zip(ref.get_attr("_last_rand_vec"), mine.get_attr("_last_rand_vec"))):
            assert np.array_equal(v1, v2[:len(v1)]), (t, e)          # 3- or 6-long in the reference, one 6-wide table here
        assert ref.get_attr("current_task_idx") == mine.get_attr("current_task_idx")
    ref.close(); mine.close()


def test_factories_and_errors(hostsim):
    with pytest.raises(ValueError):
        mk.make_mt_envs("MT11")
    with pytest.raises(ValueError):
        mk.make_ml_envs("ML11")
    with pytest.raises(AssertionError):          # 5 test classes do not divide 12 (metaworld/__init__.py:527-529)
        mk.make_ml_envs("ML10", split="test", meta_batch_size=12, lib=hostsim)
    with pytest.raises(AssertionError):          # 50 goals over 4 sub-envs per class: uneven (metaworld/__init__.py:540-542)
        mk.make_ml_envs("ML10", split="test", meta_batch_size=20, lib=hostsim)
    v1 = mk.make_mt_envs("reach-v3", reward_function_version="v1", lib=hostsim)          # one library since round 5: v1 is a run-time flag (tests/test_v1_rewards.py)
    assert v1.reward_function_version == "v1"
    v1.close()
    with pytest.raises(ValueError):
        mk.make_mt_envs("reach-v3", reward_function_version="v3", lib=hostsim)
    with pytest.raises(NotImplementedError):
        mk.make_mt_envs("MT10", autoreset_mode="NextStep", lib=hostsim)
    env = mk.make_ml_envs_test("ML10", seed=3, meta_batch_size=20, total_tasks_per_cls=40, lib=hostsim)
    assert env.num_envs == 20 and env.terminate_on_success and env.task_select == "pseudorandom" and not env.sample_tasks_on_reset
    assert all(len(t) == 10 for t in env.get_attr("tasks")) and env.partially_observable
    env.close()
    env = mk.make_mt_envs("MT10", seed=5, use_one_hot=True, num_envs=20, lib=hostsim)
    assert env.num_envs == 20 and env.single_observation_space.shape == (49,)
    from metaworld_amd import tasks as T
    assert np.array_equal(env.goal_tables["reach-v3"], T.goal_table("MT10", "reach-v3", 5))          # one seed feeds the goals too
    env.close()
    assert mk.register_mw_envs() in (True, False)


def test_normalizers_and_checkpoint(hostsim):
    """gymnasium-style normalisers (restated, unpinned): running statistics per sub-env; the checkpoint carries them"""
    kw = dict(seed=1, num_envs=3, max_episode_steps=4, reward_normalization_method="gymnasium", normalize_observations=True, lib=hostsim)
    env = mk.make_mt_envs("reach-v3", **kw)
    obs, _ = env.reset()
    assert obs.dtype == np.float32 and np.abs(obs).max() < 2e-2          # first sample: (x - mean) / std ~ 1e-2 x
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (12, 3, 4)).astype(np.float32)
    for t in range(6):
        obs, rew, te, tr, info = env.step(acts[t])
    ck = env.call("get_checkpoint")          # one (env_id, ckpt) pair per sub-env, like the reference's CheckpointWrapper
    tail = [env.step(acts[t]) for t in range(6, 12)]
    env2 = mk.make_mt_envs("reach-v3", **kw)
    env2.reset()
    env2.call("load_checkpoint", ck)
    for t in range(6, 12):
        o, r, te, tr, info = env2.step(acts[t])
        assert np.array_equal(o, tail[t - 6][0]) and np.array_equal(r, tail[t - 6][1])
    env.close(); env2.close()
    # raw reward r with return variance v -> r / sqrt(v + 1e-8)
    env = mk.make_mt_envs("reach-v3", seed=1, num_envs=2, reward_normalization_method="gymnasium", lib=hostsim)
    raw = mk.make_mt_envs("reach-v3", seed=1, num_envs=2, lib=hostsim)
    env.reset(); raw.reset()
    a = np.zeros((2, 4), dtype=np.float32)
    r_n, r_r = env.step(a)[1], raw.step(a)[1]
    cnt = 1e-4
    var = (1.0 * cnt + r_r ** 2 * cnt / (cnt + 1)) / (cnt + 1)
    assert np.allclose(r_n, r_r / np.sqrt(var + 1e-8), rtol=1e-12)
    env.close(); raw.close()


@needs_ref
def test_custom_benchmarks_match_reference(hostsim):
    """"Meta-World/custom-mt-envs" (env idx = MT1(name, seed + idx), its own stream) and CustomML (one stream over the list)"""
    warnings.filterwarnings("ignore")
    from oracle import refshim
    refshim.install()
    import gymnasium as gym
    import metaworld
    import pickle
    from functools import partial
    lst, seed = ["drawer-close-v3", "reach-v3", "window-open-v3"], 11
    ref = gym.vector.SyncVectorEnv([partial(metaworld.make_mt_envs, n, num_tasks=3, env_id=i, seed=seed + i, use_one_hot=True, max_episode_steps=3)
                                    for i, n in enumerate(lst)], autoreset_mode="SameStep")          # metaworld/__init__.py:752-767
    mine = mk.make_custom_mt_envs(lst, seed=seed, use_one_hot=True, max_episode_steps=3, precision="fp64", lib=hostsim)
    assert mine.single_observation_space.shape == ref.single_observation_space.shape == (42,)
    for a, b in zip(ref.get_attr("tasks"), mine.get_attr("tasks")):
        rv = [pickle.loads(t.data)["rand_vec"] for t in a]
        assert all(np.array_equal(v, w[:len(v)]) for v, w in zip(rv, b))
    o1, _ = ref.reset(); o2, _ = mine.reset()
    _cmp_obs(o1, o2)
    rng = np.random.default_rng(2)
    for t in range(10):
        a = rng.uniform(-1, 1, (3, 4)).astype(np.float32)
        o1, r1, te1, tr1, i1 = ref.step(a); o2, r2, te2, tr2, i2 = mine.step(a)
        assert np.array_equal(tr1, tr2)
        for v1, v2 in zip(ref.get_attr("_last_rand_vec"), mine.get_attr("_last_rand_vec")):
            assert np.array_equal(v1, v2[:len(v1)])          # per-env streams (seed + idx) drive the redraws
        _cmp_obs(o1, o2)
    ref.close(); mine.close()

    bench = metaworld.CustomML(["reach-v3", "drawer-close-v3"], ["window-open-v3"], seed=seed)
    mine = mk.make_custom_ml_envs(["reach-v3", "drawer-close-v3"], ["window-open-v3"], seed=seed, meta_batch_size=4, lib=hostsim)
    got = mine.get_attr("tasks")
    for k, name in enumerate(["reach-v3", "reach-v3", "drawer-close-v3", "drawer-close-v3"]):
        rv = [pickle.loads(t.data)["rand_vec"] for t in bench.train_tasks if t.env_name == name][k % 2::2]
        assert all(np.array_equal(v, w[:len(v)]) for v, w in zip(rv, got[k])) and len(rv) == len(got[k]) == 25
    assert mine.partially_observable
    mine.close()
    with pytest.raises(ValueError):
        mk.make_custom_ml_envs(["reach-v3"], ["reach-v3"], lib=hostsim)


def _task_names(envs):
    from metaworld_amd import tasks as T
    cls_to_name = {T.TASK_CONST[n]["cls"]: n for n in T.ALL_V3}
    return [cls_to_name[c] for c in envs.get_attr("task_name")]          # metaworld/evaluation.py:38-45


@pytest.mark.parametrize("env_name", ["reach-v3", "button-press-v3", "stick-pull-v3"])
@pytest.mark.parametrize("split", ("train", "test"))
def test_ml1(hostsim, env_name, split):
    """mirror of the reference's tests/metaworld/test_gym_make.py:126-155 (`test_ml1`)"""
    gen = mk.make_ml_envs_train if split == "train" else mk.make_ml_envs_test
    envs = gen(env_name, meta_batch_size=10, max_episode_steps=10, lib=hostsim)
    assert envs.num_envs == 10
    assert all(t == env_name for t in _task_names(envs))
    assert sum(len(t) for t in envs.get_attr("tasks")) == 50
    assert all(envs.get_attr("_partially_observable"))
    envs.close()


@pytest.mark.parametrize("benchmark", ("ML10", "ML45"))
@pytest.mark.parametrize("split", ("train", "test"))
def test_ml_benchmarks(hostsim, benchmark, split):
    """mirror of the reference's tests/metaworld/test_gym_make.py:158-211 (`test_ml_benchmarks`)"""
    from metaworld_amd import tasks as T
    meta_batch_size = 20 if benchmark != "ML45" else 45
    total_tasks_per_cls = 45 if benchmark == "ML45" else (40 if split == "test" else 50)
    gen = mk.make_ml_envs_train if split == "train" else mk.make_ml_envs_test
    envs = gen(benchmark, meta_batch_size=meta_batch_size, max_episode_steps=10, total_tasks_per_cls=total_tasks_per_cls, lib=hostsim)
    assert envs.num_envs == meta_batch_size
    names = _task_names(envs)
    expected = T.benchmark_task_names(f"{benchmark}-{split}")
    assert set(names) == set(expected)
    per = {t: 0 for t in expected}
    for tasks, n in zip(envs.get_attr("tasks"), names):
        per[n] += len(tasks)
    assert all(v == total_tasks_per_cls for v in per.values())
    assert all(envs.get_attr("_partially_observable"))
    envs.close()


@needs_ref
@pytest.mark.parametrize("name", ["door-open-v3", "pick-place-v3", "hammer-v3"])
def test_single_goal_envs_match_reference(hostsim, name):
    """tests/integration/test_single_goal_envs.py: `<name>-goal-observable` / `-goal-hidden` (seed) hold one frozen goal --
    the same draws as goal 0 of MT1(name, seed) -- shown or hidden in the observation; equal seeds give equal envs."""
    warnings.filterwarnings("ignore")
    from oracle import refshim
    refshim.install()
    from metaworld.env_dict import ALL_V3_ENVIRONMENTS_GOAL_HIDDEN, ALL_V3_ENVIRONMENTS_GOAL_OBSERVABLE
    for seed in (5, 10):
        ref = ALL_V3_ENVIRONMENTS_GOAL_OBSERVABLE[name + "-goal-observable"](seed=seed)
        mine = mk.make_goal_observable(name + "-goal-observable", seed=seed, num_envs=2, precision="fp64", lib=hostsim)
        o_ref, _ = ref.reset()
        o, _ = mine.reset()
        rv = mine.get_attr("_last_rand_vec")[0]
        assert np.array_equal(ref._last_rand_vec, rv[:len(ref._last_rand_vec)])
        _cmp_obs(o_ref, o[0]); assert np.array_equal(o[0], o[1])
        a = np.random.default_rng(seed).uniform(-1, 1, (3, 4)).astype(np.float32)
        for t in range(3):
            o_ref, r_ref, *_ = ref.step(a[t])
            o, r, *_ = mine.step(np.stack([a[t], a[t]]))
            _cmp_obs(o_ref, o[0]); assert abs(r_ref - r[0]) < 1e-6
        o2, _ = mine.reset()
        assert np.array_equal(mine.get_attr("_last_rand_vec")[0], rv)          # one goal, frozen
        hid_ref = ALL_V3_ENVIRONMENTS_GOAL_HIDDEN[name + "-goal-hidden"](seed=seed)
        hid = mk.make_goal_hidden(name, seed=seed, precision="fp64", lib=hostsim)
        oh_ref, _ = hid_ref.reset(); oh, _ = hid.reset()
        assert np.all(oh[0, -3:] == 0) and np.all(oh_ref[-3:] == 0)
        _cmp_obs(oh_ref, oh[0])
        assert np.array_equal(hid.get_attr("_last_rand_vec")[0], rv)
        mine.close(); hid.close()
    with pytest.raises(KeyError):
        mk.make_goal_observable("no-such-v3", lib=hostsim)
