"""The reference's behavioural gate on the oracle engine: its scripted policies, driving its own env classes (unmodified
Python from /root/reference, via oracle/refshim.py), must reach success.  Only runs where the reference checkout exists."""
import os

import numpy as np
import pytest

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
