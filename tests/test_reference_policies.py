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
    fails on the oracle engine (profiles/r01_policy_gate_oracle.txt) and is only required to run."""
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
