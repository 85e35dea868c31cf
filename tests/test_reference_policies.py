"""The reference's behavioural gate on the oracle engine: its scripted policies, driving its own env classes (unmodified
Python from /root/reference, via oracle/refshim.py), must reach success.  Only runs where the reference checkout exists."""
import os

import numpy as np
import pytest

from tests.helpers import make_env

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/metaworld"), reason="reference checkout not present")

TASKS = ["reach-v3", "push-v3", "pick-place-v3", "door-open-v3", "drawer-open-v3", "button-press-topdown-v3", "window-open-v3",
         "hammer-v3", "stick-pull-v3", "assembly-v3"]


@pytest.mark.parametrize("task", TASKS)
def test_scripted_policy_succeeds_on_oracle(task):
    import warnings
    warnings.filterwarnings("ignore")
    from oracle import refshim
    refshim.install()
    import metaworld
    from metaworld.policies import ENV_POLICY_MAP
    mt1 = metaworld.MT1(task, seed=42)
    env = mt1.train_classes[task]()
    policy = ENV_POLICY_MAP[task]()
    wins = 0
    for t in mt1.train_tasks[:3]:
        env.set_task(t)
        obs, _ = env.reset()
        for _ in range(500):
            obs, r, te, tr, info = env.step(policy.get_action(obs))
            if int(info["success"]) == 1:
                wins += 1
                break
    assert wins >= 2, f"{task}: {wins}/3"


def _all_tasks():
    from metaworld_amd import tasks as T
    return T.ALL_V3


@pytest.mark.parametrize("task", _all_tasks())
def test_scripted_policy_succeeds_on_device_code(hostsim, task):
    """Closed loop: the reference's scripted policies (metaworld/policies, unmodified) drive the DEVICE lane programs
    (host build, fp32 = the throughput precision) through the VectorEnv boundary, 5 goals per task; the reference's own
    gate is 80 % success (tests/metaworld/test_scripted_policies.py).  basketball is the one task whose policy also
    fails on the oracle engine (profiles/r02_policy_gate_oracle_50goals.txt) and is only required to run."""
    import warnings
    warnings.filterwarnings("ignore")
    from oracle import refshim
    refshim.install()
    from metaworld.policies import ENV_POLICY_MAP
    env = make_env(hostsim, task, n=5, precision="fp32")
    obs = env.ctx.reset(np.arange(5)).copy()
    policies = [ENV_POLICY_MAP[task]() for _ in range(5)]
    done = np.zeros(5, dtype=bool)
    for t in range(500):
        a = np.stack([np.clip(p.get_action(o[:39].copy()), -1, 1) for p, o in zip(policies, obs)]).astype(np.float32)
        obs, r, te, tr, su, info = env.ctx.step(a)
        obs = obs.copy()
        done |= su.astype(bool)
        if done.all():
            break
    env.close()
    need = 0 if task == "basketball-v3" else 4
    assert done.sum() >= need, f"{task}: {int(done.sum())}/5"


def test_reference_evaluation_loop_runs_on_the_vector_env(hostsim):
    """Drop-in at the evaluation surface: the reference's own `metaworld.evaluation.evaluation` (unmodified; it toggles
    terminate_on_success through VectorEnv.call, reads task names through get_attr and episode statistics from the
    SAME_STEP `final_info`) drives MetaWorldGpuVectorEnv("MT10") with the reference's scripted policies as the agent."""
    import warnings
    warnings.filterwarnings("ignore")
    from metaworld_amd import tasks as T
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    from oracle import refshim
    refshim.install()
    from metaworld.evaluation import evaluation
    from metaworld.policies import ENV_POLICY_MAP

    env = MetaWorldGpuVectorEnv("MT10", num_envs=10, seed=42, use_one_hot=True, precision="fp32", lib=hostsim)

    class ScriptedAgent:
        def __init__(self):
            self.policies = [ENV_POLICY_MAP[n]() for n in env.env_task_names]

        def eval_action(self, observations):
            return np.stack([np.clip(p.get_action(np.asarray(o[:39], dtype=np.float64)), -1, 1)
                             for p, o in zip(self.policies, observations)]).astype(np.float32)

        def reset(self, env_mask):
            for i in np.flatnonzero(env_mask):
                self.policies[i] = ENV_POLICY_MAP[env.env_task_names[i]]()

    mean_success, mean_return, per_task, returns = evaluation(ScriptedAgent(), env, num_episodes=2)
    env.close()
    assert set(per_task) == set(T.MT10) and all(len(r) == 2 for r in returns.values())
    assert mean_success >= 0.8, per_task
    assert env.terminate_on_success is False          # evaluation() restores the flag it found


def test_reference_metalearning_evaluation_runs_on_ml_split(hostsim):
    """The reference's `metalearning_evaluation` (unmodified) on MetaWorldGpuVectorEnv("ML10-test"): it relies on
    call("toggle_sample_tasks_on_reset" / "toggle_terminate_on_success" / "sample_tasks"), reset and step only."""
    import warnings
    warnings.filterwarnings("ignore")
    from metaworld_amd import tasks as T
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    from oracle import refshim
    refshim.install()
    from metaworld.evaluation import metalearning_evaluation
    from metaworld.policies import ENV_POLICY_MAP

    env = MetaWorldGpuVectorEnv("ML10-test", num_envs=5, seed=42, precision="fp32", lib=hostsim, max_episode_steps=150,
                                partially_observable=False)      # goal visible, as tests/metaworld/test_evaluation.py:70-82

    class Agent:
        def init(self):
            self.policies = [ENV_POLICY_MAP[n]() for n in env.env_task_names]

        def _act(self, observations):
            return np.stack([np.clip(p.get_action(np.asarray(o[:39], dtype=np.float64)), -1, 1)
                             for p, o in zip(self.policies, observations)]).astype(np.float32)

        def adapt_action(self, observations):
            return self._act(observations), {}

        def eval_action(self, observations):
            return self._act(observations)

        def step(self, timestep):
            pass

        def adapt(self):
            pass

        def reset(self, env_mask):
            for i in np.flatnonzero(env_mask):
                self.policies[i] = ENV_POLICY_MAP[env.env_task_names[i]]()

    mean_success, mean_return, per_task = metalearning_evaluation(Agent(), env, num_evals=1, adaptation_steps=1,
                                                                   adaptation_episodes=1, evaluation_episodes=1)
    env.close()
    assert set(per_task) == set(T.benchmark_task_names("ML10-test")) and 0.0 <= mean_success <= 1.0 and np.isfinite(mean_return)
    assert mean_success >= 0.6, per_task
