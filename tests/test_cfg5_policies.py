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


def _run(lib, precision):
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    G = np.load(FIX)
    env = MetaWorldGpuVectorEnv("ML45-train", num_envs=2048, seed=42, goal_seed=42, use_one_hot=False, precision=precision,
                                partially_observable=False, max_episode_steps=500, lib=lib)
    assert list(env.env_task_names) == [str(t) for t in G["task"]]
    pid = np.array([T.ALL_V3.index(n) for n in env.env_task_names], dtype=np.int32)
    sched = np.stack([G["goal"], G["goal"]]).astype(np.int32)          # one episode per env on its assigned goal
    ep, su, ms = env.ctx.policy_rollout(pid, sched, 500)
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


def test_cfg5_success_counts_equal_the_reference_on_the_host_build(hostsim):
    rows, dev, ref, ms = _run(hostsim, "fp64")
    _report(rows, dev, ref, None, "hostbuild fp64")
    assert all(a == b for _, a, b, _, _ in rows), [r for r in rows if r[1] != r[2]]
