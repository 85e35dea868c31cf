"""The boundary's checkpoints have the reference's shape (CheckpointWrapper, metaworld/wrappers.py:275-301: `envs.call("get_checkpoint")`
-> one `(env_id, ckpt)` pair per sub-env; `load_checkpoint(list of pairs)` looks its own id up and raises ValueError when it is
missing; ckpt keys of RandomTaskSelectWrapper / PseudoRandomTaskSelectWrapper, wrappers.py:125-131, :187-193), and the one-hot may be
wider than the benchmark (`num_tasks`, metaworld/__init__.py:434-436, :501)."""
import base64
import pickle

import numpy as np
import pytest

from metaworld_amd import make as mk


def test_checkpoint_pairs_like_the_reference(hostsim):
    env = mk.make_mt_envs("MT10", seed=3, num_envs=10, max_episode_steps=5, precision="fp32", lib=hostsim)
    env.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (14, 10, 4)).astype(np.float32)
    for t in range(7):
        env.step(acts[t])
    ck = env.call("get_checkpoint")
    assert len(ck) == 10 and all(isinstance(c, tuple) and len(c) == 2 and isinstance(c[0], str) and isinstance(c[1], dict) for c in ck)
    assert len({c[0] for c in ck}) == 10 and ck[0][0].startswith("<class 'metaworld.envs.sawyer_") and ck[3][0].endswith("_3")
    for _, c in ck:
        assert {"tasks", "rng_state", "sample_tasks_on_reset", "env_rng_state"} <= set(c)
        assert c["rng_state"]["bit_generator"] == "PCG64" and len(c["tasks"]) == 50
        d = pickle.loads(base64.b64decode(c["tasks"][0]["data"]))
        assert d["rand_vec"].shape[0] in (3, 6) and d["partially_observable"] is False
    # the stream state is the one a RandomTaskSelectWrapper seeded with 3 has after the same number of draws
    gen = np.random.Generator(np.random.PCG64(np.random.SeedSequence(3)))
    for _ in range(int(ck[0][1]["mwgpu"]["reset_count"])):
        gen.choice(50)
    assert ck[0][1]["rng_state"] == gen.bit_generator.state
    tail = [env.step(acts[t]) for t in range(7, 14)]
    env2 = mk.make_mt_envs("MT10", seed=3, num_envs=10, max_episode_steps=5, precision="fp32", lib=hostsim)
    env2.reset()
    env2.call("load_checkpoint", list(reversed(ck)))          # order does not matter: every sub-env finds its own id
    for t in range(7, 14):
        o, r, te, tr, info = env2.step(acts[t])
        assert np.array_equal(o, tail[t - 7][0]) and np.array_equal(r, tail[t - 7][1]) and np.array_equal(tr, tail[t - 7][3])
    with pytest.raises(ValueError, match="no checkpoint found with id"):
        env2.call("load_checkpoint", list(ck[:4]))
    env.close(); env2.close()


def test_pseudorandom_checkpoint_keys(hostsim):
    env = mk.make_mt_envs("reach-v3", seed=1, num_envs=2, task_select="pseudorandom", precision="fp32", lib=hostsim)
    env.call("sample_tasks")          # (PseudoRandomTaskSelectWrapper does not sample on reset: wrappers.py:154)
    ck = env.call("get_checkpoint")
    assert "current_task_idx" in ck[0][1] and "rng_state" not in ck[0][1]
    env.close()


def test_one_hot_wider_than_the_benchmark(hostsim):
    env = mk.make_mt_envs("MT10", seed=1, num_tasks=16, use_one_hot=True, precision="fp32", lib=hostsim)
    obs, _ = env.reset()
    assert obs.shape == (10, 39 + 16) and env.single_observation_space.shape == (55,)
    assert np.array_equal(obs[:, 39:], np.eye(10, 16, dtype=obs.dtype))
    o, *_ = env.step(np.zeros((10, 4), dtype=np.float32))
    assert np.array_equal(o[:, 39:], np.eye(10, 16, dtype=o.dtype))
    env.close()
    with pytest.raises(IndexError):          # the reference's OneHotWrapper: one_hot[task_idx] = 1.0 with task_idx >= num_tasks
        mk.make_mt_envs("MT10", seed=1, num_tasks=4, use_one_hot=True, precision="fp32", lib=hostsim)
