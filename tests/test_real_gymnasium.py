"""Drop-in readiness against a REAL gymnasium (the reference's own `tests/metaworld/test_gym_make.py:37-93`).

gymnasium cannot be installed in this container or on the GPU box, so the first group of tests checks, without it, that the env
class carries every attribute `gymnasium.vector.VectorEnv` (gymnasium 1.1) defines and `gym.make_vec` / vector wrappers touch;
the second group runs wherever a real `gymnasium` imports (auto-skipped otherwise): the class IS a `gymnasium.vector.VectorEnv`,
`register_mw_envs("Meta-World")` + `gym.make_vec("Meta-World/MT10", ...)` resolves to it, and the checks of the reference's
`test_mt_benchmarks` hold.  The lane programs run on the host harness here and through libmwgpu.so under `-m gpu`."""
import importlib.util
import sys

import numpy as np
import pytest


def _real_gymnasium():
    mod = sys.modules.get("gymnasium")
    if mod is not None and getattr(mod, "__file__", None) is None:          # oracle/refshim.py's stand-in is installed in this process
        return None
    if importlib.util.find_spec("gymnasium") is None:
        return None
    import gymnasium
    return gymnasium if hasattr(gymnasium, "__version__") else None


def test_vector_env_surface_without_gymnasium(hostsim):
    """every attribute of gymnasium.vector.VectorEnv (1.1): spec / render_mode / closed / metadata / np_random / np_random_seed /
    unwrapped / render / close_extras / close(**kwargs), on the plain-object spelling of the class"""
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    env = MetaWorldGpuVectorEnv("MT1", "reach-v3", num_envs=2, seed=3, lib=hostsim)
    assert env.unwrapped is env and env.spec is None and env.render_mode is None and env.closed is False
    assert str(getattr(env.metadata["autoreset_mode"], "value", env.metadata["autoreset_mode"])) == "SameStep" and env.metadata["render_modes"] == []
    env.unwrapped.spec = "written by make_vec"          # gymnasium/envs/registration.py: `env.unwrapped.spec = copied_id_spec`
    assert env.spec == "written by make_vec"
    a, b = env.np_random.integers(1 << 30), np.random.Generator(np.random.PCG64(np.random.SeedSequence(3))).integers(1 << 30)
    assert a == b and env.np_random_seed == 3
    env.np_random = np.random.default_rng(9)
    assert env.np_random_seed == -1
    assert env.render() is None
    assert env.single_action_space.shape == (4,) and env.action_space.shape == (2, 4) and env.observation_space.shape == (2, 39)
    obs, info = env.reset(seed=1, options=None)
    assert obs.shape == (2, 39) and info == {}
    env.close(terminate=True)                            # VectorEnv.close(**kwargs) -> close_extras(**kwargs)
    assert env.closed is True
    env.close()


@pytest.mark.parametrize("benchmark,ntask", (("MT10", 10),))
def test_make_vec_with_a_real_gymnasium(hostsim, benchmark, ntask):
    gym = _real_gymnasium()
    if gym is None:
        pytest.skip("a real gymnasium is not importable here (pip install 'gymnasium>=1.1'); tools/pin/run_pin.sh runs this test where it is")
    from metaworld_amd import make as mk
    from metaworld_amd import tasks as T
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    assert issubclass(MetaWorldGpuVectorEnv, gym.vector.VectorEnv)
    assert mk.register_mw_envs("Meta-World") is True
    max_episode_steps = 10
    envs = gym.make_vec(f"Meta-World/{benchmark}", seed=42, use_one_hot=True, max_episode_steps=max_episode_steps, lib=hostsim)
    assert isinstance(envs.unwrapped, MetaWorldGpuVectorEnv) and envs.unwrapped.spec is not None and envs.unwrapped.spec.id == f"Meta-World/{benchmark}"
    assert isinstance(envs.single_observation_space, gym.spaces.Box) and isinstance(envs.action_space, gym.spaces.Box)
    # the reference's test_mt_benchmarks, check by check (tests/metaworld/test_gym_make.py:54-93)
    cls_to_name = {T.TASK_CONST[n]["cls"]: n for n in T.ALL_V3}
    names = [cls_to_name[c] for c in envs.get_attr("task_name")]
    assert envs.num_envs == ntask and set(names) == set(T.benchmark_task_names(benchmark, None))
    for env_tasks in envs.get_attr("tasks"):
        assert len(env_tasks) == 50                      # _N_GOALS
    obs, _ = envs.reset()
    original_vecs = envs.get_attr("_last_rand_vec")
    has_truncated = False
    for _ in range(max_episode_steps + 1):
        obs, _, _, truncated, _ = envs.step(envs.action_space.sample())
        assert set(np.argmax(obs[:, -envs.num_envs:], axis=1)) == set(range(envs.num_envs))
        has_truncated |= bool(truncated.any())
    assert has_truncated
    assert any(np.any(a != b) for a, b in zip(original_vecs, envs.get_attr("_last_rand_vec")))
    assert not all(envs.get_attr("_partially_observable"))
    envs.close()


@pytest.mark.gpu
def test_make_vec_with_a_real_gymnasium_gpu(gpulib):
    gym = _real_gymnasium()
    if gym is None:
        pytest.skip("a real gymnasium is not importable here (pip install 'gymnasium>=1.1'); tools/pin/run_pin.sh runs this test where it is")
    from metaworld_amd import make as mk
    mk.register_mw_envs("Meta-World")
    envs = gym.make_vec("Meta-World/MT50", num_envs=100, seed=42, use_one_hot=True, lib=gpulib)
    obs, _ = envs.reset()
    assert obs.shape == (100, 89) and isinstance(envs.unwrapped, gym.vector.VectorEnv)
    obs, rew, term, trunc, info = envs.step(envs.action_space.sample())
    assert np.isfinite(obs).all() and rew.shape == (100,)
    envs.close()
