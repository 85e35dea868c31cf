#!/usr/bin/env python
"""Record, ON THE GPU, the states `bench.py` times and what the device computes from them -- the input of the full-size parity
check against the reference's own Python (tests/test_bench_state_parity.py, which runs where /root/reference exists).

The batch is built and pre-rolled exactly as bench.py::prepare does (MT50 @ 4096 / MT10 @ 10 240, fp64, one-hot, episode phases
staggered over the 500-step horizon, one untimed horizon of random actions + warm-up).  For PER_TASK environments of every task,
spread over the episode phases, it stores the persistent state before a step (qpos, qvel, qacc_warmstart, ctrl, mocap, relocated
body positions, time, the task block), the action, and the device's outputs of that step (obs, reward, success, info, flags),
plus the state after it.  Small (a few hundred KB): committed under tests/golden/.

usage (GPU box):  python tools/dump_bench_states.py [MT50 4096] [MT10 10240] -> gpurun_out/benchstate_<bench>_<n>.npz"""
import os
import sys
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv  # noqa: E402

PER_TASK = 4
COLS = ("qpos", "qvel", "warm", "ctrl", "mocap", "reloc", "time", "task")


def pick_envs(env, elapsed):
    """PER_TASK envs of every task at evenly spread episode phases (quantiles of the elapsed-step distribution of the task)"""
    names = np.array(env.env_task_names)
    out = []
    for t in env.task_list:
        idx = np.flatnonzero(names == t)
        order = idx[np.argsort(elapsed[idx])]
        qs = [(2 * k + 1) * len(order) // (2 * PER_TASK) for k in range(PER_TASK)]
        out += [int(order[q]) for q in qs]
    return out


def dump(bench_name, n, out_dir, lib=None):
    env = MetaWorldGpuVectorEnv(bench_name, num_envs=n, seed=42, use_one_hot=True, precision="fp64", lib=lib)
    args = SimpleNamespace(no_stagger=False, warmup=20, allow_status=False, fixed_goals=False)
    bench.prepare(env, args, 0)
    ctx = env.ctx
    # TimeLimit phase of every env now: prepare() gave env i the phase 7919 i mod 500 and ran 500 + warm-up steps
    elapsed = (np.arange(n, dtype=np.int64) * 7919 + bench.HORIZON + args.warmup) % bench.HORIZON
    chosen = pick_envs(env, elapsed)
    assert all(int(ctx.read(e, "task")[3]) == elapsed[e] for e in chosen[::13])
    rec = {c: [ctx.read(e, c) for e in chosen] for c in COLS}
    acts = np.random.default_rng(123).uniform(-1, 1, (n, 4)).astype(np.float32)
    o, r, te, tr, su, info = ctx.step(acts, env._next_goal)
    st = ctx.status()
    assert st["flags"] == 0, st
    post = {c: [ctx.read(e, c) for e in chosen] for c in ("qpos", "qvel", "mocap")}
    names = [env.env_task_names[e] for e in chosen]
    from metaworld_amd import native
    res = dict(bench=bench_name, n=n, env=np.array(chosen), task=np.array(names), action=acts[chosen],
               source_hash=np.array(native.source_hash()),          # the device sources this recording belongs to (checked by the test)
               obs=o[chosen].copy(), reward=r[chosen].copy(), terminated=te[chosen].copy(), truncated=tr[chosen].copy(),
               success=su[chosen].copy(), info=info[chosen].copy(), final_obs=ctx.final_obs[chosen].copy())
    for c in COLS:          # ragged over scenes: padded with NaN, true lengths beside
        m = max(len(v) for v in rec[c])
        res["pre_" + c] = np.array([np.pad(v, (0, m - len(v)), constant_values=np.nan) for v in rec[c]])
        res["len_" + c] = np.array([len(v) for v in rec[c]])
    for c in post:
        m = max(len(v) for v in post[c])
        res["post_" + c] = np.array([np.pad(v, (0, m - len(v)), constant_values=np.nan) for v in post[c]])
    path = os.path.join(out_dir, f"benchstate_{bench_name}_{n}.npz")
    np.savez_compressed(path, **res)
    print(f"{bench_name} @ {n}: {len(chosen)} envs recorded -> {path} ({os.path.getsize(path) / 1024:.0f} KiB); elapsed steps of the "
          f"chosen envs {int(elapsed[chosen].min())}..{int(elapsed[chosen].max())}; status {st}", flush=True)
    env.close()


if __name__ == "__main__":
    a = sys.argv[1:] or ["MT50", "4096"]
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    lib = None
    if os.environ.get("MW_HOST_HARNESS"):          # CPU dry run on the host build of the lane programs
        import __graft_entry__ as g
        from metaworld_amd import native
        lib = native.load("mwh_", g.build_host_harness())
    for k in range(0, len(a), 2):
        dump(a[k], int(a[k + 1]), out_dir, lib)
