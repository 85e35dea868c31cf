"""BASELINE config 5 -- "ML45-train with scripted metaworld.policies actions, 2048 envs/GPU, success-rate parity vs CPU" -- at full
size: every env runs one closed-loop 500-step episode of its task under the device-side scripted policy (mw_policy_rollout), and
the per-task success counts must EQUAL those of the reference's own env classes + policies on the oracle engine for the same
(task, goal) assignment (tests/golden/cfg5_ml45_train_2048_seed42.npz, tools/gen_cfg5_fixture.py).  The GPU library under `-m gpu`;
the host build of the same lane programs here (the assignment, the rollout plumbing and the policies are the same code)."""
import os

import numpy as np
import pytest

from metaworld_amd import tasks as T
from tests.helpers import ROOT

FIX = os.path.join(ROOT, "tests", "golden", "cfg5_ml45_train_2048_seed42.npz")


def _run(lib, precision, steps_per_launch=None):
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    G = np.load(FIX)
    env = MetaWorldGpuVectorEnv("ML45-train", num_envs=2048, seed=42, goal_seed=42, use_one_hot=False, precision=precision,
                                partially_observable=False, max_episode_steps=500, lib=lib)
    assert list(env.env_task_names) == [str(t) for t in G["task"]]
    pid = np.array([T.ALL_V3.index(n) for n in env.env_task_names], dtype=np.int32)
    sched = np.stack([G["goal"], G["goal"]]).astype(np.int32)          # one episode per env on its assigned goal
    ep, su, ms = env.ctx.policy_rollout(pid, sched, 500, steps_per_launch=steps_per_launch)
    flags = env.status()["flags"]
    env.close()
    assert flags == 0 and (ep == 1).all()
    names = np.array(env.env_task_names)
    rows = []
    for t in env.task_list:
        m = names == t
        rows.append((t, int(su[m].sum()), int(G["success"][m].sum()), int(m.sum()), np.flatnonzero(m)[su[m] != G["success"][m]].tolist()))
    return rows, float(su.mean()), float(G["success"].mean()), ms


def _report(rows, dev, ref, ms, what):
    lines = [f"config 5 (ML45-train @ 2048, scripted policies, one 500-step episode per env), {what}: mean success device {dev:.4f} reference {ref:.4f}"
             + (f", 500 policy + step launches in {ms:.0f} ms = {2048 * 500 / ms / 1e3:.2f} M env-steps/s" if ms else "")]
    lines += [f"  {t:30s} device {a:3d} reference {b:3d} of {n:3d}  envs that differ {d}" for t, a, b, n, d in rows]
    print("\n".join(lines))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"cfg5_{what.split()[0]}.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass


@pytest.mark.gpu
def test_cfg5_success_counts_equal_the_reference_on_the_gpu(gpulib):
    rows, dev, ref, ms = _run(gpulib, "fp64")
    _report(rows, dev, ref, ms, "gpu fp64")
    assert all(a == b for _, a, b, _, _ in rows), [r for r in rows if r[1] != r[2]]
    # the same rollout with 50 (policy, step) pairs per launch (mw_policy_rollout_fused): same episodes, no per-step batch synchronisation
    rows_f, dev_f, _, ms_f = _run(gpulib, "fp64", steps_per_launch=50)
    _report(rows_f, dev_f, ref, ms_f, "gpu_fused fp64, 50 steps per launch")
    assert [(t, a) for t, a, _, _, _ in rows_f] == [(t, a) for t, a, _, _, _ in rows]


def test_cfg5_success_counts_equal_the_reference_on_the_host_build(hostsim):
    rows, dev, ref, ms = _run(hostsim, "fp64")
    _report(rows, dev, ref, None, "hostbuild fp64")
    assert all(a == b for _, a, b, _, _ in rows), [r for r in rows if r[1] != r[2]]


def test_fused_policy_rollout_equals_the_two_kernel_loop(hostsim):
    """mw_policy_rollout_fused on a small batch: identical episodes and successes per env as one (policy, step) pair per launch"""
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    out = []
    for spl in (None, 7):
        env = MetaWorldGpuVectorEnv("MT10", num_envs=20, seed=5, use_one_hot=True, precision="fp64", lib=hostsim, max_episode_steps=60)
        pid = np.array([T.ALL_V3.index(n) for n in env.env_task_names], dtype=np.int32)
        sched = np.stack([np.arange(20) % 50, (np.arange(20) + 7) % 50, (np.arange(20) + 13) % 50]).astype(np.int32)
        ep, su, _ = env.ctx.policy_rollout(pid, sched, 150, steps_per_launch=spl)
        out.append((ep.copy(), su.copy(), [env.ctx.read(e, "qpos").copy() for e in range(20)]))
        assert env.status()["flags"] == 0
        env.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][0].sum() > 20
    assert all(np.array_equal(x, y) for x, y in zip(out[0][2], out[1][2]))
