"""Lanes-per-workgroup choice from the measured wave-time table (metaworld_amd/lpb_policy.py)."""
import numpy as np
import pytest

from metaworld_amd import lpb_policy as LP


def test_choose_minimises_the_slowest_wave_within_the_slot_budget():
    costs = {"heavy": {"1": 4.0, "2": 5.0, "4": 6.5, "8": 9.0}, "light": {"1": 1.5, "2": 1.6, "4": 1.8, "8": 2.4, "16": 3.9},
             "mid": {"2": 3.0, "4": 3.5, "8": 5.5}}
    n = {"heavy": 80, "light": 800, "mid": 400}
    # plenty of slots: everybody at the fastest setting that is not beaten by the critical path (heavy at 1 -> theta = 4.0)
    pick = LP.choose(n, costs=costs, wave_slots=10_000)
    assert pick == {"heavy": 1, "light": 16, "mid": 4}
    # 300 slots: heavy at 1 (80) + mid at 4 (100) + light at 16 (50) = 230 fits theta = 4.0 too
    assert LP.choose(n, costs=costs, wave_slots=300) == pick
    # 200 slots: theta has to rise until the waves fit
    p2 = LP.choose(n, costs=costs, wave_slots=200)
    waves = sum(-(-n[m] // l) for m, l in p2.items())
    assert waves <= 200 and max(costs[m][str(l)] for m, l in p2.items()) == 5.0
    # an unmeasured model, or a batch that cannot fit: leave it to the runtime
    assert LP.choose({"heavy": 10, "unknown": 5}, costs=costs) == {}
    assert LP.choose(n, costs=costs, wave_slots=50) == {}


def test_explicit_lanes_per_block_is_honoured_and_does_not_change_results(hostsim):
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    acts = np.random.default_rng(0).uniform(-1, 1, (12, 6, 4)).astype(np.float32)
    runs = []
    for lpb in (None, {"sawyer_reach_v3": 2}, {"sawyer_reach_v3": 16}):
        env = MetaWorldGpuVectorEnv("MT1", "reach-v3", num_envs=6, seed=3, precision="fp64", lib=hostsim, lanes_per_block=lpb)
        assert env.lanes_per_block == (lpb or {})
        env.reset()
        runs.append(np.stack([env.step(a)[0] for a in acts]))
        env.close()
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])          # the host harness' sub-lanes do not depend on it
    with pytest.raises(RuntimeError):
        MetaWorldGpuVectorEnv("MT1", "reach-v3", num_envs=6, seed=3, lib=hostsim, lanes_per_block={"sawyer_reach_v3": 3})
