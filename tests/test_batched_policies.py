"""metaworld_amd/policies.py (batched numpy restatement of all 50 scripted policies) against the reference policies,
action for action, on observations of closed-loop episodes (needs /root/reference), and closed loop on the device code."""
import os

import numpy as np
import pytest

from metaworld_amd import policies as P
from tests.helpers import golden, make_env

HAVE_REF = os.path.isdir("/root/reference/metaworld")


@pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present")
@pytest.mark.parametrize("task", sorted(P.POLICIES))
def test_batched_policy_equals_reference_policy(task):
    import warnings
    warnings.filterwarnings("ignore")
    from oracle import refshim
    refshim.install()
    from metaworld.policies import ENV_POLICY_MAP
    G = golden(f"policy_{task}_seed42.npz")
    rng = np.random.default_rng(0)
    # full observations along a successful episode (goal = last 3 of the golden's reset obs ... rebuilt from obs18 + goal)
    T = len(G["obs18"])
    goal = golden(f"trace_{task}_seed42.npz")["reset_obs"][0][36:39]
    obs = np.zeros((T, 39)); obs[:, :18] = G["obs18"]; obs[1:, 18:36] = G["obs18"][:-1]; obs[0, 18:36] = G["obs18"][0]; obs[:, 36:39] = goal
    obs = np.concatenate([obs, obs + rng.normal(0, 0.01, obs.shape), obs + rng.normal(0, 0.05, obs.shape)])
    ref = np.stack([ENV_POLICY_MAP[task]().get_action(o.copy()) for o in obs])
    got = P.POLICIES[task](obs)
    assert got.dtype == np.float32 and np.array_equal(got, ref.astype(np.float32))


@pytest.mark.parametrize("task", sorted(P.POLICIES))
def test_batched_policy_succeeds_closed_loop_on_device_code(hostsim, task):
    env = make_env(hostsim, task, n=5, precision="fp32")
    obs = env.ctx.reset(np.arange(5)).copy()
    done = np.zeros(5, dtype=bool)
    for t in range(500):
        obs, r, te, tr, su, info = env.ctx.step(P.batched_actions([task] * 5, obs))
        obs = obs.copy()
        done |= su.astype(bool)
        if done.all():
            break
    env.close()
    assert done.sum() >= (0 if task == "basketball-v3" else 4), int(done.sum())     # (basketball's policy fails on the oracle too)
