#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched Meta-World step kernel on MI355X.

One "step" = one VectorEnv.step over the rank's env batch (mocap update + 5 physics substeps + forward +
obs/reward + wrappers + SAME_STEP auto-reset), actions resident in HBM, outputs left in HBM.
`python bench.py --gpus N --steps K --warmup W`; for N > 1 launched under torch.distributed.run
(one process per GPU, weak scaling: --envs per GPU).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = {"fp32": 900.0, "fp64": 1600.0}   # action + state in, state + obs out
HBM_PEAK_GBPS = 8000.0                                        # MI355X_MICROARCH.md


def cpu_baseline(task_names, seconds=15.0):
    """The CPU oracle (fp64 C restatement, 1 thread) stepping the same workload: every task of the benchmark gets an equal
    share of the env-steps (as in the vector env), random actions, 5 substeps + forward per env-step.  Physics only (the
    reference's Python obs/reward layer is not part of the port): an upper bound on the port's speed."""
    from metaworld_amd import tasks as T
    from oracle.mjlite import OracleData, OracleModel
    rng = np.random.default_rng(0)
    sims = []
    for task in task_names:
        c = T.TASK_CONST[task]
        om = OracleModel(T.compiled_model(c["model"]))
        om.view("eq_data")[:] = [0, 0, 0, 0, 0, 0, -1, 0, 0, 0, 5.0]
        d = OracleData(om)
        d.mocap_pos[:] = np.array(c["hand_init_pos"]); d.mocap_quat[:] = [1, 0, 1, 0]; d.ctrl[:] = [-1, 1]
        d.step(100)
        sims.append((om, d, np.array(c["mocap_low"]), np.array(c["mocap_high"])))
    chunk = 10
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for om, d, lo_, hi_ in sims:          # one round = `chunk` env-steps of every task
            for _ in range(chunk):
                a = rng.uniform(-1, 1, 4)
                d.mocap_pos[:] = np.clip(d.mocap_pos + 0.01 * a[:3], lo_, hi_)
                d.ctrl[:] = [a[3], -a[3]]
                d.step(5)
                d.forward()
                n += 1
    dt = time.perf_counter() - t0
    what = task_names[0] if len(task_names) == 1 else f"{len(task_names)} tasks in equal shares"
    return {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} env-steps of {what} (random actions, 5 substeps + forward each, physics only) in {dt:.1f}s on 1 host thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps (default covers two full 500-step episodes incl. the auto-reset waves)")
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--benchmark", default="auto")
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the extra fp64 (parity precision) measurement at N=1")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from metaworld_amd import tasks as T
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    bench = args.benchmark
    if bench == "auto":   # the metric's config (MT50) once every task has device code; until then the largest supported set
        bench = "MT50" if len(T.supported_tasks()) == 50 else "MT1"
    if bench == "MT1":
        env = MetaWorldGpuVectorEnv("MT1", "reach-v3", num_envs=args.envs, seed=42 + rank, precision=args.precision,
                                    device_id=local_rank, rank=rank, world_size=world)
        workload, wl_task = f"MT1 reach-v3, {args.envs} batched envs/GPU, {args.precision}, random actions", ["reach-v3"]
    else:
        env = MetaWorldGpuVectorEnv(bench, num_envs=args.envs, seed=42 + rank, use_one_hot=True, precision=args.precision,
                                    device_id=local_rank, rank=rank, world_size=world)
        workload, wl_task = f"{bench} sync-vector, {args.envs} envs/GPU, {args.precision}, random actions", list(env.task_list)
    N = env.num_envs
    env.reset()
    T_act = 64
    acts = np.random.default_rng(rank).uniform(-1, 1, (T_act, N, 4)).astype(np.float32)
    env.ctx.upload_actions(acts)
    env.ctx.step_resident(args.warmup)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    kernel_ms = env.ctx.step_resident(args.steps)     # K launches on the library's stream, bracketed by HIP events
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        tw = torch.tensor([wall], device="cuda")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    if world > 1:   # the one real exchange of this path: per-step bookkeeping all-gather over RCCL (not in the timed loop)
        from metaworld_amd.vector_env import gather_bookkeeping
        env.ctx.step(acts[0], env._next_goal)
        gather_bookkeeping(env.bookkeeping(), device=torch.device("cuda", local_rank))
    if rank == 0:
        total_steps = N * world * args.steps
        value = total_steps / wall
        per_launch_s = kernel_ms / 1e3 / args.steps
        bytes_launch = ALGO_BYTES_PER_ENV_STEP[args.precision] * N
        ach = bytes_launch / per_launch_s / 1e9
        traffic = None                 # HBM bytes per launch from the committed PMC profile of this exact workload
        prof = os.path.join(ROOT, "profiles", "r01_mt50_pmc.json")
        if os.path.exists(prof):
            with open(prof) as f:
                pj = json.load(f)
            if pj.get("workload") == workload and world == 1:
                traffic = pj["fetch_bytes_per_launch"] + pj["write_bytes_per_launch"]
        out = {"metric": "env-steps/sec (whole node) MT50 @4096 envs/GPU; achieved HBM GB/s vs peak", "value": value,
               "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.precision == "fp32" else "f64", "data": "synthetic",
               "config": {"workload": workload, "envs_per_gpu": N, "tasks_with_device_code": len(T.supported_tasks()),
                          "parallelism": f"dp{world} (independent env shards, no data-path collective)"},
               "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                            "traffic": traffic, "kernel_ms_per_launch": kernel_ms / args.steps,
                            "note": "achieved = algorithmic bytes/env-step x envs / HIP-event kernel time; traffic = FETCH_SIZE + "
                                    "WRITE_SIZE bytes per launch of the committed rocprofv3 profile (profiles/r01_mt50_pmc.json); the "
                                    "step kernel is bound by the latency of its per-environment dependency chain, not by HBM (DESIGN.md 5)"}}
        if world == 1 and args.precision == "fp32" and not args.no_parity_mode:
            # the same workload in the parity precision (fp64 state/arithmetic: obs/reward <= 1e-5 vs the reference traces)
            env.close()
            env64 = MetaWorldGpuVectorEnv(bench, "reach-v3" if bench == "MT1" else None, num_envs=args.envs, seed=42 + rank,
                                          use_one_hot=bench != "MT1", precision="fp64", device_id=local_rank)
            env64.reset()
            env64.ctx.upload_actions(acts)
            env64.ctx.step_resident(args.warmup)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            k64 = env64.ctx.step_resident(args.steps)
            torch.cuda.synchronize()
            w64 = time.perf_counter() - t1
            out["parity_mode"] = {"dtype": "f64", "value": N * args.steps / w64, "unit": "env-steps/s", "ms_per_step": w64 / args.steps * 1e3,
                                  "kernel_ms_per_launch": k64 / args.steps}
            env64.close()
        if world == 1 and not args.no_cpu_baseline:      # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(wl_task)
        print(json.dumps(out))
    env.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
